"""developer timing + residual check of the device modified Cholesky (gsl_nls_large lm step), p = 100 .. 2000"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gslnls_amd import _lib
L = _lib.lib()
rng = np.random.default_rng(1)
for p in (64, 100, 250, 500, 777, 1000, 1500, 2000):
    J = rng.standard_normal((2 * p, p))
    A = np.ascontiguousarray(J.T @ J)
    diag = np.sqrt(np.diag(A)).copy(); rhs = rng.standard_normal(p); sol = np.zeros(p)
    args = (p, A.ctypes.data_as(_lib.DP), diag.ctypes.data_as(_lib.DP), 1e-3, rhs.ctypes.data_as(_lib.DP), sol.ctypes.data_as(_lib.DP))
    rc = L.gslnls_debug_mchol_solve(*args)
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        rc = L.gslnls_debug_mchol_solve(*args)
    dt = (time.perf_counter() - t0) / reps
    M = A + 1e-3 * np.diag(diag ** 2)
    print(("pivoted" if os.environ.get("GSLNLS_LARGE_CHOL_PIVOTED") == "1" else "natural") + " p = %4d: rc %d, %.3f ms per solve (incl. the %d KB upload), residual %.2e" % (
        p, rc, 1e3 * dt, p * p * 8 // 1024, np.max(np.abs(M @ sol - rhs)) / np.max(np.abs(rhs))), flush=True)
