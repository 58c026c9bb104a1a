"""are there periodic stalls in a long sequence of C2 fits (runtime housekeeping)?"""
import sys, os, ctypes as C, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import c2_data
from gslnls_amd import _lib
from gslnls_amd.control import gsl_nls_control, pack_control
L = _lib.lib()
x, y = c2_data(1_000_000)
X = np.asfortranarray(x.reshape(-1, 1))
model = _lib.Model(1, 3, 1, X.ctypes.data_as(C.c_void_p), 0)
err = C.c_int(0)
h = L.gslnls_dense_create(C.byref(model), y.ctypes.data_as(C.c_void_p), len(y), None, C.byref(err))
ci, cd = pack_control(gsl_nls_control(solver="cholesky"), "lm")
par = np.zeros(3); res = _lib.Result(); res.par = par.ctypes.data_as(_lib.DP)
st = np.array([1.0, 1.0, 0.0])
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
ts = np.zeros(N)
for i in range(N):
    t0 = time.perf_counter()
    L.gslnls_dense_solve(h, 1, 0, st.ctypes.data_as(_lib.DP), None, ci.ctypes.data_as(_lib.IP), cd.ctypes.data_as(_lib.DP), 16, C.byref(res))
    ts[i] = time.perf_counter() - t0
print("median %.4f ms, mean %.4f ms, max %.3f ms; fits slower than 1 ms: %s" % (np.median(ts) * 1e3, ts.mean() * 1e3, ts.max() * 1e3,
      [(int(i), round(float(ts[i]) * 1e3, 2)) for i in np.nonzero(ts > 1e-3)[0]]))
