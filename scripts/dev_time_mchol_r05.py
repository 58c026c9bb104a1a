"""developer aid (round 5): the damped solve with J^T J resident on the device (the lm step's call) -- wall time per solve,
device milliseconds by HIP events, the relative residual, and the solution compared bit for bit with the previous forms of
the kernels (GSLNLS_LARGE_PANEL_V1=1: the one-wavefront diagonal block; GSLNLS_LARGE_LOOKAHEAD=1: second stream; GSLNLS_LARGE_BACK_V1=1: the one-workgroup back substitution; GSLNLS_LARGE_STEP_V1=1: panel and trailing update as two launches per step).
Usage: python scripts/dev_time_mchol_r05.py [reps [p ...]]"""
import sys, os, time, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gslnls_amd import _lib
L = _lib.lib()
DP = C.POINTER(C.c_double)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sizes = [int(a) for a in sys.argv[2:]] or [65, 100, 128, 200, 333, 500, 1000, 2000]
rng = np.random.default_rng(7)
SWITCHES = ["GSLNLS_LARGE_SMALL_OFF", "GSLNLS_LARGE_BACK_STEPWISE", "GSLNLS_LARGE_STEP_V1", "GSLNLS_LARGE_PANEL_V1", "GSLNLS_LARGE_LOOKAHEAD", "GSLNLS_LARGE_BACK_V1", "GSLNLS_LARGE_BACKUPD_V1"]
for p in sizes:
    J = rng.standard_normal((p + 50, p))
    A = np.ascontiguousarray(J.T @ J)
    d = np.sqrt(np.diag(A)).copy()
    rhs = rng.standard_normal(p)
    M = A + 1e-3 * np.diag(d * d)
    dA = C.c_void_p()
    L.gslnls_debug_device_alloc(C.byref(dA), A.nbytes)
    L.gslnls_debug_device_copy(dA, A.ctypes.data_as(C.c_void_p), A.nbytes, 1)
    base = None
    modes = ["default"] + SWITCHES + ["all previous"]
    if os.environ.get("GSLNLS_DEV_ONLY_DEFAULT"):
        modes = ["default"]
    for mode in modes:
        for s in SWITCHES:
            os.environ.pop(s, None)
        if mode == "all previous":
            for s in ("GSLNLS_LARGE_PANEL_V1", "GSLNLS_LARGE_BACK_V1"):
                os.environ[s] = "1"
        elif mode != "default":
            os.environ[mode] = "1"
        sol = np.zeros(p)
        rargs = (p, dA, d.ctypes.data_as(DP), 1e-3, rhs.ctypes.data_as(DP), sol.ctypes.data_as(DP))
        rc = L.gslnls_debug_mchol_solve_resident(*rargs)
        t0 = time.perf_counter()
        for _ in range(reps):
            rc = L.gslnls_debug_mchol_solve_resident(*rargs) or rc
        el = (time.perf_counter() - t0) / reps
        dev = []
        L.gslnls_debug_mchol_timing(1)  # (the event pair around a solve's kernels: off in the product path, 6 us per solve)
        for _ in range(reps):
            rc = L.gslnls_debug_mchol_solve_resident(*rargs) or rc
            dev.append(L.gslnls_debug_mchol_last_device_ms())
        L.gslnls_debug_mchol_timing(0)
        if base is None:
            base = sol.copy()
        print("p = %4d  %-26s rc %d  wall %.3f ms  device %.3f ms (min %.3f)  rel. residual %.2e  same bits as default: %s" % (
            p, mode, rc, el * 1e3, float(np.mean(dev)), float(np.min(dev)), np.linalg.norm(M @ sol - rhs) / np.linalg.norm(rhs),
            bool(np.array_equal(sol, base))), flush=True)
    for s in SWITCHES:
        os.environ.pop(s, None)
    # the host routine that served p < 400 until round 5
    sol = np.zeros(p)
    hargs = (p, A.ctypes.data_as(DP), d.ctypes.data_as(DP), 1e-3, rhs.ctypes.data_as(DP), sol.ctypes.data_as(DP))
    L.gslnls_debug_host_mchol_solve(*hargs)
    t0 = time.perf_counter()
    nh = max(1, reps // 4)
    for _ in range(nh):
        L.gslnls_debug_host_mchol_solve(*hargs)
    print("          host routine: %.3f ms per solve, max |device - host| / |host| = %.2e" % (
        (time.perf_counter() - t0) / nh * 1e3, float(np.max(np.abs(base - sol)) / np.max(np.abs(sol)))), flush=True)
    L.gslnls_debug_device_free(dA)
