"""developer timing of the wide path's damped solve alone (one wavefront), p = 16, 32, 48, 64"""
import os, sys
os.environ["GSLNLS_WIDE_SOLVE_REPS"] = "200"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gslnls_amd import _lib
L = _lib.lib()
rng = np.random.default_rng(1)
for p in (16, 32, 48, 64):
    J = rng.standard_normal((4 * p, p))
    A = J.T @ J
    Ap = np.ascontiguousarray(np.concatenate([A[i, :i + 1] for i in range(p)]))
    diag = np.sqrt(np.diag(A)); rhs = rng.standard_normal(p); sol = np.zeros(p)
    L.gslnls_debug_wide_solve(p, Ap.ctypes.data_as(_lib.DP), diag.ctypes.data_as(_lib.DP), 1e-3, rhs.ctypes.data_as(_lib.DP), sol.ctypes.data_as(_lib.DP))
    print(p, np.max(np.abs((A + 1e-3 * np.diag(diag ** 2)) @ sol - rhs)))
