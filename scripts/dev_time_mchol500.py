"""developer profile target: the device modified Cholesky at p = 500 only (README Example 4's size), 20 solves"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gslnls_amd import _lib
L = _lib.lib()
rng = np.random.default_rng(1)
p = int(sys.argv[1]) if len(sys.argv) > 1 else 500
J = rng.standard_normal((2 * p, p))
A = np.ascontiguousarray(J.T @ J)
diag = np.sqrt(np.diag(A)).copy(); rhs = rng.standard_normal(p); sol = np.zeros(p)
args = (p, A.ctypes.data_as(_lib.DP), diag.ctypes.data_as(_lib.DP), 1e-3, rhs.ctypes.data_as(_lib.DP), sol.ctypes.data_as(_lib.DP))
L.gslnls_debug_mchol_solve(*args)
t0 = time.perf_counter()
for _ in range(20):
    L.gslnls_debug_mchol_solve(*args)
print("p = %d: %.3f ms per solve" % (p, 1e3 * (time.perf_counter() - t0) / 20))
