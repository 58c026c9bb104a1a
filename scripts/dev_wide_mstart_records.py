"""developer script: per-point records of a wide (p = 12) concentration batch against the oracle's single-start runs"""
import sys, os, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import gslref
import gslnls_amd as amd
from gslnls_amd import _lib
from gslnls_amd.control import gsl_nls_control, pack_control
from test_gpu_wide import gaussians_problem
q = gaussians_problem(4, 0, 1500, seed=13, noise=0.02)
p, n, N = 12, 1500, 12
lo = q["truth"] * np.where(np.arange(p) % 3 == 1, 0.97, 0.8)
hi = q["truth"] * np.where(np.arange(p) % 3 == 1, 1.03, 1.2)
ranges = np.ascontiguousarray(np.stack([lo, hi], axis=1).reshape(-1))
kd = np.full(p, 0.75)
ci, cd = pack_control(gsl_nls_control(solver="cholesky"), "lm")
prob = amd.DenseProblem(_lib.MODEL_EXPR, p, q["x"], q["y"], expr=q["formula"].split("~")[1].strip(), parnames=q["names"], xnames=["x"], lowering="jit")
K = 3 * p + 8
for first in (0, 12, 24):
    rec = np.zeros((N, K)); ms = C.c_float(0)
    rc = _lib.lib().gslnls_mstart_batch(prob._h, 1, ranges.ctypes.data_as(_lib.DP), kd.ctypes.data_as(_lib.DP), first, N, 0, N, 5, 1e-6,
                                        ci.ctypes.data_as(_lib.IP), cd.ctypes.data_as(_lib.DP), None, rec.ctypes.data_as(C.c_void_p), 0, C.byref(ms))
    assert rc == 0, rc
    octrl = gslref.control(solver="cholesky", maxiter=5, gtol=1e-3)
    for i in range(N):
        x0 = rec[i, 2 * p:3 * p]; sc = rec[i, 3 * p:]
        o = gslref.nls(n, p, x0, fn=lambda th: q["model"](th) - q["y"], jac=q["jac"], ctrl=octrl)
        J = q["jac"](o["par"]); det1 = np.linalg.det(J.T @ J); J0 = q["jac"](x0); det0 = np.linalg.det(J0.T @ J0)
        print(first + i, "niter", int(sc[5]), o["niter"], "status", int(sc[6]), o["conv"], "ssr %.10g %.10g" % (sc[1], o["ssr"]),
              "det0 %.4g %.4g det1 %.4g %.4g" % (sc[2], det0, sc[3], det1), "dpar %.2e" % np.max(np.abs(rec[i, :p] - o["par"]) / np.abs(o["par"])))
