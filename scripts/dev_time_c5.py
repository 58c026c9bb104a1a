"""C5: 4096 data sets x n = 1e4, p = 8 (Gauss1 family), bisquare IRLS, one workgroup per data set"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from test_gpu_batch import c5_data, GAUSS1_START
import gslnls_amd as A

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
t0 = time.time()
X, Y, TH = c5_data(B, n)
print("generated in %.1f s" % (time.time() - t0), flush=True)
prob = A.BatchProblem(4, 8, X, Y)
for rep in range(3):
    t0 = time.time()
    out = prob.irls(GAUSS1_START, loss="bisquare", jac=True, control=dict(solver="cholesky"))
    el = time.time() - t0
    it = int(out["irls_niter"].sum())
    print("rep %d: wall %.3f s kernel %.1f ms -> %.0f datasets/s, %.0f IRLS iterations/s; conv ok %d/%d irls ok %d; mean irls iters %.2f, mean last niter %.2f, max rel par err %.3g" % (
        rep, el, out["kernel_ms"], B / el, it / el, int((out["conv"] == 0).sum()), B, int((out["irls_status"] == 0).sum()),
        out["irls_niter"].mean(), out["niter"].mean(), np.max(np.abs(out["par"] / TH - 1))))
prob.close()
