"""diagnostic: where one step of the resident lm_fit_kernel spends its time (stamps build, 100 MHz ticks = 10 ns)"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import c2_data
from gslnls_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "gslnls_amd", "libgslnls_hip_stamps.so")
import gslnls_amd as A
L = _lib.lib()
L.gslnls_debug_stamps.argtypes = [C.c_void_p, C.c_int, _lib.DP, C.c_int, C.POINTER(C.c_ulonglong), _lib.IP]
for n in [int(a) for a in sys.argv[1:]] or [512, 4096, 1_000_000]:
    x, y = c2_data(n)
    prob = A.DenseProblem(1, 3, x, y)
    out = np.zeros(8 * 2 * 256, dtype=np.uint64)
    nrows = C.c_int(0)
    th = np.array([4.0, 1.2, 0.8])
    rc = L.gslnls_debug_stamps(prob._h, 1, th.ctypes.data_as(_lib.DP), -1, out.ctypes.data_as(C.POINTER(C.c_ulonglong)), C.byref(nrows))
    G = nrows.value // 2
    st = out[:8 * nrows.value].reshape(G, 2, 8).astype(np.int64)
    t0 = st[:, 0, 0].min()
    rel = (st - t0) * 10  # ns since the first workgroup left B1
    rel[st == 0] = -1
    print("n=%d G=%d rc=%d fast=%s (ns after the first workgroup left B1 of the stamped step)" % (n, G, rc, os.environ.get("GSLNLS_PERSIST_FAST", "0")))
    for role, names in ((0, ["B1 left", "rows done", "own sums published", "totals picked up", "advance+publish done", "", "", "B2 left"]),
                        (1, ["B1 left", "rows done", "own sums published", "leader: members gathered", "leader: group total stored",
                             "leader: other groups gathered", "leader: totals stored", "B2 left"])):
        for k, nm in enumerate(names):
            v = rel[:, role, k]
            v = v[v >= 0]
            if len(v):
                print("  wave %d  %-32s min=%6d med=%6d max=%6d (n=%d)" % (role, nm, v.min(), np.median(v), v.max(), len(v)))
    prob.close()
