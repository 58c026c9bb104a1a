"""developer aid: wall time of every one of N damped solves (J^T J resident) -- looking for the host-side stalls the bench sees at
p = 2000 (median 1.06 ms, one call in twenty ~70 ms).  Usage: python scripts/dev_mchol_stalls.py [p [N]]"""
import sys, os, time, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gslnls_amd import _lib
L = _lib.lib()
DP = C.POINTER(C.c_double)
p = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rng = np.random.default_rng(7)
J = rng.standard_normal((p + 50, p))
A = np.ascontiguousarray(J.T @ J)
d = np.sqrt(np.diag(A)).copy()
rhs = rng.standard_normal(p)
dA = C.c_void_p()
L.gslnls_debug_device_alloc(C.byref(dA), A.nbytes)
L.gslnls_debug_device_copy(dA, A.ctypes.data_as(C.c_void_p), A.nbytes, 1)
sol = np.zeros(p)
rargs = (p, dA, d.ctypes.data_as(DP), 1e-3, rhs.ctypes.data_as(DP), sol.ctypes.data_as(DP))
L.gslnls_debug_mchol_solve_resident(*rargs)
ts = []
for _ in range(N):
    t0 = time.perf_counter()
    L.gslnls_debug_mchol_solve_resident(*rargs)
    ts.append((time.perf_counter() - t0) * 1e3)
ts = np.array(ts)
out = np.flatnonzero(ts > 3 * np.median(ts))
print("p = %d: %d solves, median %.3f ms, mean %.3f ms, max %.1f ms; outliers (> 3 x median) at calls %s: %s ms" % (
    p, N, np.median(ts), ts.mean(), ts.max(), out.tolist(), np.round(ts[out], 1).tolist()))
