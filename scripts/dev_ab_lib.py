"""developer A/B of two builds of the library on the same box: python scripts/dev_ab_lib.py libA.so libB.so [n]
(the C2 fit, adaptive chunks, 3 rounds x 200 fits per build, interleaved; each build in its own child process)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
ROOT = %r
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gslnls_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "gslnls_amd", sys.argv[1])
import numpy as np
from conftest import c2_data
import gslnls_amd as A
n = int(sys.argv[2])
x, y = c2_data(n)
prob = A.DenseProblem(1, 3, x, y)
ctrl = A.gsl_nls_control(solver="cholesky")
for jac in (True, False):
    for _ in range(30):
        fit = prob.solve([1.0, 1.0, 0.0], jac=jac, control=ctrl, want_vectors=False)
    t0 = time.perf_counter(); dev = 0.0
    for _ in range(300):
        fit = prob.solve([1.0, 1.0, 0.0], jac=jac, control=ctrl, want_vectors=False)
        dev += fit["loop_ms"]
    el = (time.perf_counter() - t0) / 300
    tp = prob.time_pass(np.array([5.0, 1.5, 1.0]) * 1.01, jac=jac, reps=500)
    print("%%-28s jac=%%d: wall %%.4f ms/fit, device loop %%.4f ms/fit, %%d launches, %%.0f it/s, time_pass %%.2f us/launch" %% (
        sys.argv[1], jac, el * 1e3, dev / 300, fit["n_launches"], fit["niter"] / el, tp * 1e3))
prob.close()
''' % ROOT
libs = sys.argv[1:3]
n = sys.argv[3] if len(sys.argv) > 3 else "1000000"
for rnd in range(3):
    for lib in libs:
        subprocess.run([sys.executable, "-c", CHILD, lib, n], check=False)
