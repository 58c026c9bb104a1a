"""developer check of the one-launch-per-fit kernel against the launch-per-step kernel (run on the GPU box)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import c2_data
import gslnls_amd as A

sizes = [int(a) for a in sys.argv[1:]] or [4096, 100_000, 1_000_000, 4_000_000]
ctrl = A.gsl_nls_control(solver="cholesky", xtol=1.49e-8, gtol=1.49e-8)
for n in sizes:
    x, y = c2_data(n)
    prob = A.DenseProblem(1, 3, x, y)
    for jac in (True, False):
        res = {}
        for label, chunk in (("persist", 0), ("launch", -1)):
            fit = prob.solve([1.0, 1.0, 0.0], jac=jac, control=ctrl, want_vectors=False, chunk=chunk)
            reps, ms = 50, 0.0
            t0 = time.perf_counter()
            for _ in range(reps):
                fit = prob.solve([1.0, 1.0, 0.0], jac=jac, control=ctrl, want_vectors=False, chunk=chunk)
                ms += fit["loop_ms"]
            wall = (time.perf_counter() - t0) / reps * 1e3
            res[label] = fit
            print("n=%d jac=%d %-7s niter=%d conv=%d launches=%d steps=%d loop_ms=%.4f wall_ms=%.4f it/s=%.0f par=%s ssr=%.10g" % (
                n, jac, label, fit["niter"], fit["conv"], fit["n_launches"], fit.get("n_steps", -1), ms / reps, wall,
                fit["niter"] / wall * 1e3, fit["par"], fit["ssr"]), flush=True)
        a, b = res["persist"], res["launch"]
        print("   niter equal: %s   max rel par diff: %.3e   neval %s vs %s" % (
            a["niter"] == b["niter"], np.max(np.abs(a["par"] - b["par"]) / np.abs(b["par"])), a["neval"], b["neval"]), flush=True)
        tp = prob.time_pass([5.0, 1.5, 1.0], jac=jac, reps=-2000)
        tl = prob.time_pass([5.0, 1.5, 1.0], jac=jac, reps=2000)
        print("   per step: persist %.3f us, launch-per-step %.3f us" % (tp * 1e3, tl * 1e3), flush=True)
    prob.close()
