"""C5 cost model by experiment: vary irls_maxiter / LM maxiter and the number of resident data sets"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from test_gpu_batch import c5_data, GAUSS1_START
import gslnls_amd as A
n = 10000
X, Y, TH = c5_data(4096, n)
prob = A.BatchProblem(4, 8, X, Y)
def run(label, B, **ctrl):
    c = dict(solver="cholesky"); c.update(ctrl)
    best = 1e9
    for _ in range(2):
        out = prob.irls(GAUSS1_START, loss="bisquare", jac=True, control=c, lo=0, hi=B)
        best = min(best, out["kernel_ms"])
    print("%-46s B=%4d kernel %.2f ms  (LM iters of last solve mean %.2f, irls iters mean %.2f)" % (
        label, B, best, out["niter"][:B].mean(), out["irls_niter"][:B].mean()), flush=True)
for B in (256, 512, 1024, 4096):
    run("full", B)
run("irls_maxiter=1, maxiter=1 (2 passes + reweight)", 256, irls_maxiter=1, maxiter=1)
run("irls_maxiter=1, maxiter=5", 256, irls_maxiter=1, maxiter=5)
run("irls_maxiter=1, maxiter=50 (one full solve)", 256, irls_maxiter=1, maxiter=50)
run("irls_maxiter=2, maxiter=50", 256, irls_maxiter=2, maxiter=50)
prob.close()
