"""diagnostic: where one steady-state launch of lm_step_kernel spends its cycles (stamps build)"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import c2_data
from gslnls_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "gslnls_amd", os.environ.get("GSLNLS_STAMPS_LIB", "libgslnls_hip_stamps.so"))
import gslnls_amd as A
L = _lib.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
model = int(sys.argv[2]) if len(sys.argv) > 2 else 1
x, y = c2_data(n)
TH = {1: [4.0, 1.2, 0.8], 4: [98.8, 0.0105, 100.5, 67.5, 23.1, 72.0, 179.0, 18.4]}[model]
if model == 4:
    x = 250.0 * (np.arange(n) + 1) / n
    t = TH
    y = t[0] * np.exp(-t[1] * x) + t[2] * np.exp(-(x - t[3]) ** 2 / t[4] ** 2) + t[5] * np.exp(-(x - t[6]) ** 2 / t[7] ** 2)
prob = A.DenseProblem(model, len(TH), x, y)
L.gslnls_debug_stamps.argtypes = [C.c_void_p, C.c_int, _lib.DP, C.c_int, C.POINTER(C.c_ulonglong), _lib.IP]
for jac in (1, 0):
    out = np.zeros(8 * 256 * 16, dtype=np.uint64)
    nrows = C.c_int(0)
    th = np.array(TH) * 1.01
    L.gslnls_debug_stamps(prob._h, jac, th.ctypes.data_as(_lib.DP), 300, out.ctypes.data_as(C.POINTER(C.c_ulonglong)), C.byref(nrows))
    st = out[:8 * nrows.value].reshape(nrows.value, 8).astype(np.int64)
    wpb = nrows.value // 256 if nrows.value >= 256 else nrows.value
    st = st - st[:, :1]  # per wave: cycles since its own entry (counters differ between XCDs)
    t0 = 0
    rel = st
    names = ["entry", "prefetch issued", "state+partials reduced", "advance done", "after barrier", "rows done", "kernel end", "partials published"]
    w0 = rel[0::wpb]       # wave 0 of each block
    wo = np.delete(rel, np.arange(0, nrows.value, wpb), axis=0)
    print("jac=%d rows=%d (100 MHz-ish s_memtime ticks are shader cycles)" % (jac, nrows.value))
    for k, nm in enumerate(names):
        a = w0[:, k]; b = wo[:, k]
        print("  %-24s wave0: med=%7d max=%7d | other waves: med=%7d max=%7d" % (nm, np.median(a), a.max(), np.median(b) if k not in (2, 3) else -1, b.max() if k not in (2, 3) else -1))
    print("  total span (max end - min entry): %d cycles" % (st[:, 6].max() - t0))
    adv = (C.c_ulonglong * 8)()
    if hasattr(L, "gslnls_debug_adv_stamps") and L.gslnls_debug_adv_stamps(adv) == 0:
        a = [int(v) for v in adv]
        print("  block 0 wave 0, cycles after kernel entry: entry barrier passed %d | advance starts %d" % (a[7] - a[6], a[0] - a[6]))
        print("  lm_advance (block 0): rho %d | accept/reject bookkeeping %d | end of iteration + test %d | begin step (solve) %d | whole %d" % (a[1] - a[0], a[2] - a[1], a[3] - a[2], a[4] - a[3], a[4] - a[0]))
prob.close()
