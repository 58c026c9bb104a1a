"""developer timing of gslnls_mstart_batch at several batch sizes (records stay on the device)"""
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
SIZES = [int(v) for v in sys.argv[1:]] or [8192, 65536, 262144, 1048576]
for total in SIZES:
    shard = torch.zeros(total * K, dtype=torch.float64, device="cuda")
    ms = C.c_float(0)
    def step():
        rc = L.gslnls_mstart_batch(h, 1, ranges.ctypes.data_as(_lib.DP), kd.ctypes.data_as(_lib.DP), 0, total, 0, total, 5, 1e-6,
                                   ci.ctypes.data_as(_lib.IP), cd.ctypes.data_as(_lib.DP), None, C.c_void_p(shard.data_ptr()), 1, C.byref(ms))
        assert rc == 0
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 20
    print("points %8d: %.3f ms per batch (kernel %.3f ms) -> %.1f M fits/s" % (total, el * 1e3, ms.value, total / el / 1e6))
