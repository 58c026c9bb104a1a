"""developer A/B: the C2 fit with and without ladder probes (GSLNLS_PROBE is read at the first launch of every fit),
adaptive chunks as bench.py uses them, interleaved on the same box"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import c2_data
import gslnls_amd as A
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
x, y = c2_data(n)
prob = A.DenseProblem(1, 3, x, y)
ctrl = A.gsl_nls_control(solver="cholesky")
for jac in (True, False):
    for rnd in range(3):
        for mode in ("probe", "plain"):
            if mode == "plain":
                os.environ["GSLNLS_PROBE"] = "0"
            else:
                os.environ.pop("GSLNLS_PROBE", None)
            for _ in range(20):
                fit = prob.solve([1.0, 1.0, 0.0], jac=jac, control=ctrl, want_vectors=False, chunk=-1)
            t0 = time.perf_counter(); dev = 0.0
            for _ in range(200):
                fit = prob.solve([1.0, 1.0, 0.0], jac=jac, control=ctrl, want_vectors=False, chunk=-1)
                dev += fit["loop_ms"]
            el = (time.perf_counter() - t0) / 200
            print("jac=%d %-5s: wall %.4f ms/fit  device loop %.4f ms/fit  steps %d launches %d niter %d -> %.0f it/s (wall)" % (
                jac, mode, el * 1e3, dev / 200, fit["n_steps"], fit["n_launches"], fit["niter"], fit["niter"] / el))
prob.close()
