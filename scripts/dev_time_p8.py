"""developer timing: dense grid-per-fit path with p = 8 (NIST Gauss1 family) at n = 1e6"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import gslnls_amd as A
from test_gpu_batch import GAUSS1_START
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rng = np.random.Generator(np.random.PCG64(5))
x = 250.0 * (np.arange(n) + 1) / n
th = np.array([98.778210871, 0.010497276517, 100.48990633, 67.481111276, 23.129773360, 71.994503004, 178.99805021, 18.389389025])
f = th[0] * np.exp(-th[1] * x) + th[2] * np.exp(-(x - th[3]) ** 2 / th[4] ** 2) + th[5] * np.exp(-(x - th[6]) ** 2 / th[7] ** 2)
y = f + 2.5 * rng.standard_normal(n)
prob = A.DenseProblem(4, 8, x, y)
ctrl = A.gsl_nls_control(solver="cholesky")
for jac in (True, False):
    fit = prob.solve(GAUSS1_START, jac=jac, control=ctrl, want_vectors=False)
    ms = 0.0
    for _ in range(10):
        fit = prob.solve(GAUSS1_START, jac=jac, control=ctrl, want_vectors=False)
        ms += fit["loop_ms"]
    tp = prob.time_pass(th, jac=jac, reps=500)
    print("jac=%d niter=%d launches=%d loop_ms=%.3f us/launch=%.2f (16n B: %.0f GB/s) relerr=%.2e" % (
        jac, fit["niter"], fit["n_launches"], ms / 10, tp * 1e3, 16.0 * n / (tp * 1e-3) / 1e9, np.max(np.abs(fit["par"] / th - 1))))
prob.close()
