"""developer aid: wall time per damped solve of the large path's device factorisation (csrc/mchol_device.hip) at
p = 500, 1000, 2000 -- the one-launch back substitution against the launch-per-block form (GSLNLS_LARGE_BACK_BLOCKS=1),
with the two solutions compared bit for bit.  and the solve with J^T J resident on the device.
Usage: python scripts/dev_time_mchol.py [reps [p ...]]"""
import sys, os, time, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gslnls_amd import _lib
L = _lib.lib()
DP = C.POINTER(C.c_double)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sizes = [int(a) for a in sys.argv[2:]] or [500, 1000, 2000, 333]
rng = np.random.default_rng(7)
for p in sizes:
    J = rng.standard_normal((p + 50, p))
    A = np.ascontiguousarray(J.T @ J)
    d = np.sqrt(np.diag(A)).copy()
    rhs = rng.standard_normal(p)
    out = {}
    for mode in ("one launch", "per block"):
        if mode == "per block":
            os.environ["GSLNLS_LARGE_BACK_BLOCKS"] = "1"
        else:
            os.environ.pop("GSLNLS_LARGE_BACK_BLOCKS", None)
        sol = np.zeros(p)
        args = (p, A.ctypes.data_as(DP), d.ctypes.data_as(DP), 1e-3, rhs.ctypes.data_as(DP), sol.ctypes.data_as(DP))
        rc = L.gslnls_debug_mchol_solve(*args)
        t0 = time.perf_counter()
        for _ in range(reps):
            rc = L.gslnls_debug_mchol_solve(*args) or rc
        el = (time.perf_counter() - t0) / reps
        M = A + 1e-3 * np.diag(d * d)
        out[mode] = sol.copy()
        print("p = %4d  %-10s rc %d  %.3f ms per solve (upload included)  rel. residual %.2e" %
              (p, mode, rc, el * 1e3, np.linalg.norm(M @ sol - rhs) / np.linalg.norm(rhs)))
    print("          identical bits: %s" % bool(np.array_equal(out["one launch"], out["per block"])))
    os.environ.pop("GSLNLS_LARGE_BACK_BLOCKS", None)
    dA = C.c_void_p()
    L.gslnls_debug_device_alloc(C.byref(dA), A.nbytes)
    L.gslnls_debug_device_copy(dA, A.ctypes.data_as(C.c_void_p), A.nbytes, 1)
    sol = np.zeros(p)
    rargs = (p, dA, d.ctypes.data_as(DP), 1e-3, rhs.ctypes.data_as(DP), sol.ctypes.data_as(DP))
    rc = L.gslnls_debug_mchol_solve_resident(*rargs)
    t0 = time.perf_counter()
    for _ in range(reps):
        rc = L.gslnls_debug_mchol_solve_resident(*rargs) or rc
    el = (time.perf_counter() - t0) / reps
    print("          J^T J resident (the lm step's call): rc %d  %.3f ms per solve, same bits %s" %
          (rc, el * 1e3, bool(np.array_equal(sol, out["one launch"]))))
    L.gslnls_debug_device_free(dA)
