"""developer timing: the host routine of the large lm step against the device factorisation, p = 150 .. 700 -- where the
threshold GSLNLS_LARGE_CHOL_DEVICE_MIN (default 400) belongs on this host"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gslnls_amd import _lib
L = _lib.lib()
rng = np.random.default_rng(3)
for p in (150, 200, 250, 300, 350, 400, 450, 500, 600, 700):
    J = rng.standard_normal((2 * p, p))
    A = np.ascontiguousarray(J.T @ J)
    diag = np.sqrt(np.diag(A)).copy(); rhs = rng.standard_normal(p); sol = np.zeros(p)
    args = (p, A.ctypes.data_as(_lib.DP), diag.ctypes.data_as(_lib.DP), 1e-3, rhs.ctypes.data_as(_lib.DP), sol.ctypes.data_as(_lib.DP))
    out = []
    for fn in (L.gslnls_debug_host_mchol_solve, L.gslnls_debug_mchol_solve):
        fn(*args)
        t0 = time.perf_counter()
        for _ in range(10):
            fn(*args)
        out.append(1e3 * (time.perf_counter() - t0) / 10)
    print("p = %3d: host %.3f ms, device %.3f ms" % (p, out[0], out[1]), flush=True)
