"""developer aid: the one-shot gslnls_nls() at C2 (n = 1e6, p = 3) as .Call(C_nls) delivers it -- pageable host buffers in,
resid + grad + covar out into freshly allocated pageable arrays -- wall clock per call and the library's own breakdown"""
import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gslnls_amd import _lib
from gslnls_amd.control import gsl_nls_control, pack_control

L = _lib.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
rng = np.random.Generator(np.random.PCG64(20250927))
x = 3.0 * np.arange(n) / (n - 1)
y = 5.0 * np.exp(-1.5 * x) + 1.0 + 0.25 * rng.standard_normal(n)
ci, cd = pack_control(gsl_nls_control(solver="cholesky"))
ci = (C.c_int * 15)(*ci)
cd = (C.c_double * 11)(*cd)
start = (C.c_double * 3)(1.0, 1.0, 0.0)
has_prof = hasattr(L, "gslnls_last_call_profile")
if has_prof:
    L.gslnls_last_call_profile.restype = C.c_int
    L.gslnls_last_call_profile.argtypes = [_lib.DP, C.c_int]
import mmap
for jac in (1, 0):
    for want in ("vectors", "vectors-fresh-pages", "scalars"):
        ts, profs = [], []
        for rep in range(14):
            model = _lib.Model(1, 3, 1, x.ctypes.data_as(C.c_void_p), 0)
            par = np.empty(3)
            covar = np.empty(9)
            res = _lib.Result()
            t0 = time.perf_counter()
            if want == "vectors":
                resid = np.empty(n)          # (glibc recycles the previous repetition's pages: already touched)
                grad = np.empty((n, 3), order="F")
            elif want == "vectors-fresh-pages":
                # a fresh anonymous mapping per call: pages the process has never touched, as a large Rf_allocVector's
                m1, m2 = mmap.mmap(-1, 8 * n), mmap.mmap(-1, 24 * n)
                resid = np.frombuffer(m1, dtype=np.float64)
                grad = np.frombuffer(m2, dtype=np.float64).reshape((n, 3), order="F")
            if want != "scalars":
                res.resid = resid.ctypes.data_as(_lib.DP)
                res.grad = grad.ctypes.data_as(_lib.DP)
            res.par = par.ctypes.data_as(_lib.DP)
            res.covar = covar.ctypes.data_as(_lib.DP)
            rc = L.gslnls_nls(C.byref(model), y.ctypes.data_as(C.c_void_p), n, jac, 0, start, 0, None, 0, None, ci, cd, None, 0,
                              None, C.byref(res))
            t1 = time.perf_counter()
            ts.append(1e3 * (t1 - t0))
            if has_prof:
                pr = (C.c_double * 16)()
                L.gslnls_last_call_profile(pr, 16)
                profs.append(list(pr))
            assert rc == 0, rc
        ts = np.array(ts[4:])
        line = "jac=%d out=%s: median %.3f ms  min %.3f  max %.3f (10 calls after 4 warm-up)  niter %d loop_ms %.3f" % (
            jac, want, np.median(ts), ts.min(), ts.max(), res.niter, res.loop_ms)
        if has_prof:
            pm = np.median(np.array(profs[4:]), axis=0)
            line += "\n    breakdown (median ms): " + "  ".join("%s %.3f" % (k, v) for k, v in zip(
                ("create", "h2d", "loop", "finalize", "d2h", "destroy", "total"), pm))
        print(line)
        if want != "scalars":
            print("    par", par, "ssr", res.ssr, "resid[0:2]", resid[:2], "grad[0]", grad[0])
