"""multi-start kernel cost model: kernel time vs LM iterations per point"""
import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gslnls_amd import _lib
from gslnls_amd.control import gsl_nls_control, pack_control
L = _lib.lib()
X = np.asfortranarray(np.array([1., 2., 3., 5., 7., 10.]).reshape(6, 1)); y = np.array([109., 149., 149., 191., 213., 224.])
model = _lib.Model(2, 2, 1, X.ctypes.data_as(C.c_void_p), 0)
err = C.c_int(0)
h = L.gslnls_dense_create(C.byref(model), y.ctypes.data_as(C.c_void_p), 6, None, C.byref(err))
ci, cd = pack_control(gsl_nls_control(solver="cholesky"), "lm")
ranges = np.array([1.0, 500.0, 0.01, 5.0]); kd = np.array([0.75, 0.75])
K = L.gslnls_mstart_record_size(2)
total = 8192
shard = torch.zeros(total * K, dtype=torch.float64, device="cuda")
ms = C.c_float(0)
for jac in (1, 0):
    for it in (0, 1, 2, 5, 10, 20):
        ks = []
        for _ in range(10):
            rc = L.gslnls_mstart_batch(h, jac, ranges.ctypes.data_as(_lib.DP), kd.ctypes.data_as(_lib.DP), 0, total, 0, total, it, 1e-6,
                                       ci.ctypes.data_as(_lib.IP), cd.ctypes.data_as(_lib.DP), None, C.c_void_p(shard.data_ptr()), 1, C.byref(ms))
            ks.append(ms.value)
        print("jac=%d maxiter=%2d kernel %.1f us" % (jac, it, 1e3 * min(ks)))
