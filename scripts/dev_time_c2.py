"""developer timing of the C2 dense path (run on the GPU box)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from conftest import c2_data
from gslnls_amd import _lib
if os.environ.get('GSLNLS_LIB'):
    _lib.LIB_PATH = os.environ['GSLNLS_LIB']  # developer variants of the library
import gslnls_amd as A

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
x, y = c2_data(n)
ctrl = A.gsl_nls_control(solver="cholesky", xtol=1.49e-8, gtol=1.49e-8)
prob = A.DenseProblem(1, 3, x, y)
for jac in (True, False):
    for chunk in (4, 8, 16, 32):
        fit = prob.solve([1.0, 1.0, 0.0], jac=jac, control=ctrl, want_vectors=False, chunk=chunk)
        t0 = time.perf_counter()
        reps = 20
        ms = 0.0
        for _ in range(reps):
            fit = prob.solve([1.0, 1.0, 0.0], jac=jac, control=ctrl, want_vectors=False, chunk=chunk)
            ms += fit["loop_ms"]
        wall = (time.perf_counter() - t0) / reps * 1e3
        print("jac=%d chunk=%2d niter=%d launches=%d neval=%s loop_ms=%.4f wall_ms=%.4f it/s(loop)=%.0f it/s(wall)=%.0f par=%s" % (
            jac, chunk, fit["niter"], fit["n_launches"], fit["neval"], ms / reps, wall,
            fit["niter"] / (ms / reps) * 1e3, fit["niter"] / wall * 1e3, fit["par"]))
    tp = prob.time_pass([5.0, 1.5, 1.0], jac=jac, reps=2000)
    print("jac=%d time_pass avg ms/launch=%.5f -> %.1f GB/s (16n B per launch)" % (jac, tp, 16.0 * n / (tp * 1e-3) / 1e9))
prob.close()
