"""developer timing: the C2 fit through the hand-written model vs the same formula through the expression VM"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import c2_data
import gslnls_amd as A
from gslnls_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
x, y = c2_data(n)
ctrl = A.gsl_nls_control(solver="cholesky", xtol=1.49e-8, gtol=1.49e-8)
X3 = np.zeros((n, 3)); X3[:, 0] = x
for label, prob in (("native", A.DenseProblem(1, 3, x, y)),
                    ("vm", A.DenseProblem(_lib.MODEL_EXPR, 3, x.reshape(-1, 1), y, expr="A*exp(-lam*x)+b",
                                          parnames=["A", "lam", "b"], xnames=["x"], lowering="vm")),
                    ("jit", A.DenseProblem(_lib.MODEL_EXPR, 3, x.reshape(-1, 1), y, expr="A*exp(-lam*x)+b",
                                           parnames=["A", "lam", "b"], xnames=["x"], lowering="jit"))):
    for jac in (True, False):
        fit = prob.solve([1.0, 1.0, 0.0], jac=jac, control=ctrl, want_vectors=False)
        ms = 0.0
        reps = 20
        for _ in range(reps):
            fit = prob.solve([1.0, 1.0, 0.0], jac=jac, control=ctrl, want_vectors=False)
            ms += fit["loop_ms"]
        tp = prob.time_pass([5.0, 1.5, 1.0], jac=jac, reps=1000)
        print("%-6s jac=%d niter=%d launches=%d loop_ms=%.4f it/s=%.0f  us/launch=%.2f par=%s" % (
            label, jac, fit["niter"], fit["n_launches"], ms / reps, fit["niter"] / (ms / reps) * 1e3, tp * 1e3, fit["par"]))
    prob.close()
