import sys, os, json, ctypes as C, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench
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
def fits(n):
    for _ in range(n):
        L.gslnls_dense_solve(h, 1, 0, st.ctypes.data_as(_lib.DP), None, ci.ctypes.data_as(_lib.IP), cd.ctypes.data_as(_lib.DP), 16, C.byref(res))
for nfit, ms_steps, tp in ((0, 50, 0), (200, 25, 0), (100, 50, 0), (200, 50, 0), (0, 50, 2000), (0, 50, 0)):
    fits(nfit)
    if tp:
        th = np.array([4.0, 1.2, 0.8]); L.gslnls_dense_time_pass(h, 1, th.ctypes.data_as(_lib.DP), tp)
    r = bench.multistart_bench(L, _lib, torch, None, 0, 1, ms_steps, 3)
    print(nfit, ms_steps, tp, r["strong_8192_total"]["ms_per_batch"], r["weak_65536_per_gpu"]["ms_per_batch"], flush=True)
