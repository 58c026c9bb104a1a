"""developer aid (round 5): gsl_nls() on formulas with more than 64 parameters (the matrix path, csrc/bd_host.hpp) -- wall
time per trial step with the fused trial step (one host synchronisation per trial) against the stepwise form
(GSLNLS_BD_STEPWISE=1), same fit bit for bit.  Sums of Gaussians: p = 3 ng.
Usage: python scripts/dev_time_matrix_path.py [ng n ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gslnls_amd as amd

args = [int(a) for a in sys.argv[1:]] or [33, 3000, 66, 5000, 167, 20000]
for ng, n in zip(args[0::2], args[1::2]):
    rng = np.random.Generator(np.random.PCG64(ng))
    x = np.linspace(0.0, 10.0 * ng, n)
    amp, mid, wid = rng.uniform(2.0, 6.0, ng), 10.0 * np.arange(ng) + rng.uniform(3.0, 7.0, ng), rng.uniform(1.2, 2.4, ng)
    truth = np.stack([amp, mid, wid], axis=1).reshape(-1)
    y = np.sum(amp * np.exp(-((x[:, None] - mid) / wid) ** 2), axis=1) + 0.01 * rng.standard_normal(n)
    rhs = " + ".join("a%d * exp(-((x - m%d) / w%d)^2)" % (g, g, g) for g in range(ng))
    start = {}
    for g in range(ng):
        start["a%d" % g], start["m%d" % g], start["w%d" % g] = 0.9 * amp[g], mid[g] + 0.15, 1.1 * wid[g]
    fits = {}
    for mode in ("fused", "stepwise"):
        if mode == "stepwise":
            os.environ["GSLNLS_BD_STEPWISE"] = "1"
        else:
            os.environ.pop("GSLNLS_BD_STEPWISE", None)
        amd.gsl_nls("y ~ " + rhs, data=dict(x=x, y=y), start=start, jac=True, control=dict(solver="cholesky"))  # (compiles the kernels)
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            fit = amd.gsl_nls("y ~ " + rhs, data=dict(x=x, y=y), start=start, jac=True, control=dict(solver="cholesky"))
            el = time.perf_counter() - t0
            best = el if best is None else min(best, el)
        fits[mode] = fit
        print("p = %3d n = %6d  %-8s niter %3d  trial steps %3d  loop %.2f ms = %.1f us per trial step  (whole call %.2f ms)  conv %d" % (
            3 * ng, n, mode, fit["niter"], fit["n_steps"], fit["loop_ms"], 1e3 * fit["loop_ms"] / max(1, fit["n_steps"]), best * 1e3, fit["conv"]), flush=True)
    print("          same fit bit for bit: %s" % bool(np.array_equal(fits["fused"]["par"], fits["stepwise"]["par"]) and fits["fused"]["ssr"] == fits["stepwise"]["ssr"]))
os.environ.pop("GSLNLS_BD_STEPWISE", None)
