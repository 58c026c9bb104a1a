"""ctypes binding of the CPU ORACLE (oracle/_build/libgslref.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, bench.py's cpu_baseline leg
and __graft_entry__.smoke(); never from gslnls_amd/.  The call surface mirrors
what the reference's R layer hands to .Call(C_nls) / .Call(C_nls_large)
(src/nls.c:54, src/nls_large.c:66).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

F_T = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_void_p, C.POINTER(C.c_double))
DF_T = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_void_p, C.POINTER(C.c_double))
FVV_T = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p, C.POINTER(C.c_double))
DFL_T = C.CFUNCTYPE(C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p,
                    C.POINTER(C.c_double), C.POINTER(C.c_double))

DP = C.POINTER(C.c_double)
IP = C.POINTER(C.c_int)

MODEL_EXPDECAY, MODEL_MISRA1A, MODEL_GAUSSPK, MODEL_GAUSS1, MODEL_GLMEXP = 1, 2, 3, 4, 5

STATUS = {0: "success", -1: "failure", -2: "the iteration has not converged yet",
          9: "problem with user-supplied function", 11: "exceeded max number of iterations",
          27: "iteration is not making progress towards solution"}


class Problem(C.Structure):
    _fields_ = [("n", C.c_int), ("p", C.c_int), ("f", F_T), ("df", DF_T), ("fvv", FVV_T),
                ("params", C.c_void_p), ("start", DP), ("mstart", C.c_int), ("swts", DP),
                ("swts_mat", DP), ("lupars", DP), ("control_int", IP), ("control_dbl", DP),
                ("has_start", IP), ("loss_rho", C.c_int), ("loss_cc", DP)]


class Result(C.Structure):
    _fields_ = [("par", DP), ("covar", DP), ("resid", DP), ("grad", DP), ("niter", C.c_int),
                ("conv", C.c_int), ("ssr", C.c_double), ("ssrtol", C.c_double), ("neval", C.c_int * 3),
                ("info", C.c_int), ("chisq_init", C.c_double),
                ("irls_weights", DP), ("irls_psi", DP), ("irls_dpsi", DP),
                ("irls_sigma", C.c_double), ("irls_tol", C.c_double),
                ("irls_status", C.c_int), ("irls_niter", C.c_int),
                ("partrace", DP), ("ssrtrace", DP),
                ("mstart_nsp", C.c_int), ("mstart_nwsp", C.c_int), ("mstart_iters", C.c_int),
                ("mstart_stop", C.c_int), ("mstart_ssropt", C.c_double)]


class LargeProblem(C.Structure):
    _fields_ = [("n", C.c_int), ("p", C.c_int), ("f", F_T), ("df", DFL_T), ("params", C.c_void_p),
                ("start", DP), ("weights", DP), ("control_int", IP), ("control_dbl", DP)]


class LargeResult(C.Structure):
    _fields_ = [("par", DP), ("covar", DP), ("resid", DP), ("niter", C.c_int), ("conv", C.c_int),
                ("ssr", C.c_double), ("ssrtol", C.c_double), ("neval", C.c_int * 4), ("info", C.c_int),
                ("chisq_init", C.c_double), ("partrace", DP), ("ssrtrace", DP)]


class RowData(C.Structure):
    _fields_ = [("n", C.c_int), ("nx", C.c_int), ("x", DP), ("y", DP), ("model", C.c_int), ("p", C.c_int)]


def build(force=False):
    """compile oracle/_build/libgslref.so with gcc (make)."""
    out = os.path.join(_HERE, "_build", "libgslref.so")
    if force or not os.path.exists(out) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(out)
            for f in os.listdir(_HERE) if f.endswith((".c", ".h"))):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return out


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.gslref_nls.argtypes = [C.POINTER(Problem), C.POINTER(Result)]
        L.gslref_nls.restype = C.c_int
        L.gslref_nls_large.argtypes = [C.POINTER(LargeProblem), C.POINTER(LargeResult)]
        L.gslref_nls_large.restype = C.c_int
        L.gslref_sobol.argtypes = [C.c_int, C.c_int, C.c_int, DP]
        L.gslref_halton.argtypes = [C.c_int, C.c_int, C.c_int, DP]
        L.gslref_mcholesky_decomp.argtypes = [C.c_int, DP, IP]
        L.gslref_mcholesky_solve.argtypes = [C.c_int, DP, IP, DP, DP]
        L.gslref_det_cholesky_jtj.argtypes = [C.c_int, C.c_int, DP]
        L.gslref_det_cholesky_jtj.restype = C.c_double
        L.gslref_median.argtypes = [DP, C.c_int]
        L.gslref_median.restype = C.c_double
        L.gslref_mad.argtypes = [DP, C.c_int]
        L.gslref_mad.restype = C.c_double
        L.gslref_psi.argtypes = [C.c_double, DP, C.c_int]
        L.gslref_psi.restype = C.c_double
        L.gslref_psip.argtypes = [C.c_double, DP, C.c_int]
        L.gslref_psip.restype = C.c_double
        L.gslref_hat_values.argtypes = [C.c_int, C.c_int, DP, DP]
        L.gslref_cooks_d.argtypes = [C.c_int, C.c_int, DP, DP, DP]
        L.gslref_set_device_exp.argtypes = [C.c_int]
        L.gslref_set_device_exp.restype = None
        L.gslref_device_exp_array.argtypes = [DP, DP, C.c_int]
        L.gslref_device_exp_array.restype = None
        _LIB = L
    return _LIB


def _dp(a):
    return a.ctypes.data_as(DP) if a is not None else None


# ---- gsl_nls_control() defaults, R/nls.R:1186-1229 ------------------------
ALGORITHMS = {"lm": 0, "lmaccel": 1, "dogleg": 2, "ddogleg": 3, "subspace2D": 4, "cgst": 5}
SCALES = {"more": 0, "levenberg": 1, "marquardt": 2}
SOLVERS = {"qr": 0, "cholesky": 1, "svd": 2}
FDTYPES = {"forward": 0, "center": 1}
LOSSES = ["default", "huber", "barron", "bisquare", "welsh", "optimal", "hampel", "ggw", "lqq"]
LOSS_CC = {"default": [0.0], "huber": [1.345], "barron": [1.0, 1.345], "bisquare": [4.685061],
           "welsh": [2.11], "optimal": [1.060158], "hampel": [0.9016085],
           "ggw": [1.387, 1.5, 1.063], "lqq": [1.473, 0.982, 1.5]}
EPS = float(np.finfo(float).eps)


def control(**kw):
    """gsl_nls_control(): same names and defaults as R/nls.R:1186-1192."""
    c = dict(maxiter=100, scale="more", solver="qr", fdtype="forward", factor_up=2.0, factor_down=3.0,
             avmax=0.75, h_df=EPS ** 0.5, h_fvv=0.02, xtol=EPS ** 0.5, ftol=EPS ** 0.5, gtol=EPS ** 0.5,
             mstart_n=30, mstart_p=5, mstart_q=None, mstart_r=4.0, mstart_s=2, mstart_tol=0.25,
             mstart_maxiter=10, mstart_maxstart=250, mstart_minsp=1, irls_maxiter=50, irls_xtol=EPS ** 0.25)
    for k, v in kw.items():
        if k not in c:
            raise KeyError(k)
        c[k] = v
    if c["mstart_q"] is None:
        c["mstart_q"] = c["mstart_n"] // 10
    return c


def pack_control(ctrl, algorithm="lm", trace=False, startisnum=True, any_missing_range=False):
    """control_int[15] / control_dbl[11] exactly as R/nls.R:693-713 packs them."""
    ci = np.array([ctrl["maxiter"], int(trace), ALGORITHMS[algorithm], SCALES[ctrl["scale"]],
                   SOLVERS[ctrl["solver"]], FDTYPES[ctrl["fdtype"]], ctrl["mstart_n"], ctrl["mstart_p"],
                   ctrl["mstart_q"], ctrl["mstart_s"], ctrl["mstart_maxiter"], ctrl["mstart_maxstart"],
                   ctrl["mstart_minsp"], int(startisnum), ctrl["irls_maxiter"]], dtype=np.int32)
    r = ctrl["mstart_r"] * (10.0 if any_missing_range else 1.0)
    cd = np.array([ctrl["factor_up"], ctrl["factor_down"], ctrl["avmax"], ctrl["h_df"], ctrl["h_fvv"],
                   ctrl["xtol"], ctrl["ftol"], ctrl["gtol"], r, ctrl["mstart_tol"], ctrl["irls_xtol"]],
                  dtype=np.float64)
    return ci, cd


def _wrap_callbacks(n, p, fn, jac, fvv):
    """fn(theta)->residual-free model values are NOT assumed: fn returns f = model - y already
    unless y is given by the caller; here fn returns the residual vector (length n)."""
    def f_cb(xp, _params, out):
        try:
            th = np.ctypeslib.as_array(xp, shape=(p,)).copy()
            with np.errstate(all="ignore"):
                v = np.asarray(fn(th), dtype=np.float64)
            if v.shape != (n,):
                return 9
            v = np.where(np.isfinite(v), v, np.inf)
            np.ctypeslib.as_array(out, shape=(n,))[:] = v
            return 0
        except Exception:  # noqa
            return 9

    def df_cb(xp, _params, out):
        try:
            th = np.ctypeslib.as_array(xp, shape=(p,)).copy()
            with np.errstate(all="ignore"):
                J = np.asarray(jac(th), dtype=np.float64)
            if J.shape != (n, p) or not np.all(np.isfinite(J)):
                return 9
            np.ctypeslib.as_array(out, shape=(n * p,))[:] = J.reshape(-1)
            return 0
        except Exception:  # noqa
            return 9

    def fvv_cb(xp, vp, _params, out):
        try:
            th = np.ctypeslib.as_array(xp, shape=(p,)).copy()
            v = np.ctypeslib.as_array(vp, shape=(p,)).copy()
            with np.errstate(all="ignore"):
                r = np.asarray(fvv(th, v), dtype=np.float64)
            if r.shape != (n,) or not np.all(np.isfinite(r)):
                return 9
            np.ctypeslib.as_array(out, shape=(n,))[:] = r
            return 0
        except Exception:  # noqa
            return 9

    return (F_T(f_cb), DF_T(df_cb) if jac is not None else C.cast(None, DF_T),
            FVV_T(fvv_cb) if fvv is not None else C.cast(None, FVV_T))


def nls(n, p, start, fn=None, jac=None, fvv=None, rowdata=None, use_jac=True, use_fvv=False,
        algorithm="lm", ctrl=None, trace=False, weights=None, weights_matrix=None,
        lower=None, upper=None, loss="default", loss_cc=None, has_start=None):
    """Run the oracle behind gsl_nls().

    Either pass Python callbacks fn/jac/fvv (fn returns the residual model - y), or
    rowdata=dict(model=MODEL_*, x=(n,nx) array, y=(n,)) to use the plain-C row models.
    start: length-p vector (single start) or (2,p) array [lower;upper] (multi-start, NaN allowed
    only through has_start=False entries that the R layer would have filled, R/nls.R:399-437).
    """
    L = lib()
    ctrl = ctrl or control()
    start = np.asarray(start, dtype=np.float64)
    mstart = start.ndim == 2
    prob = Problem()
    keep = []
    if rowdata is not None:
        X = np.asfortranarray(np.asarray(rowdata["x"], dtype=np.float64).reshape(n, -1))
        Y = np.ascontiguousarray(rowdata["y"], dtype=np.float64)
        rd = RowData(n, X.shape[1], _dp(X), _dp(Y), rowdata["model"], p)
        keep += [X, Y, rd]
        prob.f = C.cast(L.gslref_model_f, F_T)
        prob.df = C.cast(L.gslref_model_df, DF_T) if use_jac else C.cast(None, DF_T)
        prob.fvv = C.cast(L.gslref_model_fvv, FVV_T) if use_fvv else C.cast(None, FVV_T)
        prob.params = C.cast(C.pointer(rd), C.c_void_p)
    else:
        cbs = _wrap_callbacks(n, p, fn, jac, fvv)
        keep += list(cbs)
        prob.f, prob.df, prob.fvv = cbs
        prob.params = None
    prob.n, prob.p = n, p
    if mstart:
        st = np.ascontiguousarray(start.T.reshape(-1))  # 2 x p col-major: lo0,hi0,lo1,hi1...
        hs = np.ones(2 * p, dtype=np.int32) if has_start is None else np.ascontiguousarray(
            np.asarray(has_start, dtype=np.int32).T.reshape(-1))
        any_missing = bool(np.any(hs == 0))
    else:
        st = np.ascontiguousarray(start)
        hs = np.ones(2 * p, dtype=np.int32)
        any_missing = False
    keep += [st, hs]
    prob.start, prob.mstart, prob.has_start = _dp(st), int(mstart), hs.ctypes.data_as(IP)
    sw = swm = None
    if weights is not None:
        sw = np.sqrt(np.asarray(weights, dtype=np.float64))
    if weights_matrix is not None:
        W = np.asarray(weights_matrix, dtype=np.float64)
        swm = np.asfortranarray(np.linalg.cholesky(W))  # t(chol(W)) in R == lower factor
    prob.swts, prob.swts_mat = _dp(sw), _dp(swm)
    lu = None
    if lower is not None or upper is not None:
        lo = np.full(p, -np.inf) if lower is None else np.asarray(lower, dtype=np.float64)
        up = np.full(p, np.inf) if upper is None else np.asarray(upper, dtype=np.float64)
        lu = np.ascontiguousarray(np.stack([lo, up], axis=1).reshape(-1))
    prob.lupars = _dp(lu)
    ci, cd = pack_control(ctrl, algorithm, trace, True, any_missing)
    prob.control_int, prob.control_dbl = ci.ctypes.data_as(IP), _dp(cd)
    prob.loss_rho = LOSSES.index(loss)
    cc = np.asarray(loss_cc if loss_cc is not None else LOSS_CC[loss], dtype=np.float64)
    prob.loss_cc = _dp(cc)
    keep += [sw, swm, lu, ci, cd, cc]

    res = Result()
    maxiter = ctrl["maxiter"]
    out = dict(par=np.zeros(p), covar=np.zeros((p, p), order="F"), resid=np.zeros(n),
               grad=np.zeros((n, p), order="F"), irls_weights=np.zeros(n), irls_psi=np.zeros(n),
               irls_dpsi=np.zeros(n))
    res.par, res.covar, res.resid, res.grad = _dp(out["par"]), _dp(out["covar"]), _dp(out["resid"]), _dp(out["grad"])
    res.irls_weights, res.irls_psi, res.irls_dpsi = _dp(out["irls_weights"]), _dp(out["irls_psi"]), _dp(out["irls_dpsi"])
    if trace:
        out["partrace"] = np.full((maxiter + 1, p), np.nan, order="F")
        out["ssrtrace"] = np.full(maxiter + 1, np.nan)
        res.partrace, res.ssrtrace = _dp(out["partrace"]), _dp(out["ssrtrace"])
    status = L.gslref_nls(C.byref(prob), C.byref(res))
    out.update(niter=res.niter, conv=res.conv, status=STATUS.get(res.conv, str(res.conv)), ssr=res.ssr,
               ssrtol=res.ssrtol, neval=dict(f=res.neval[0], J=res.neval[1], fvv=res.neval[2]),
               info=res.info, chisq_init=res.chisq_init, ret=status,
               irls=dict(irls_sigma=res.irls_sigma, irls_tol=res.irls_tol, irls_status=res.irls_status,
                         irls_niter=res.irls_niter, irls_conv=res.irls_status),
               mstart=dict(nsp=res.mstart_nsp, nwsp=res.mstart_nwsp, iters=res.mstart_iters,
                           stop=res.mstart_stop, ssropt=res.mstart_ssropt))
    if trace:
        out["partrace"] = out["partrace"][:res.niter + 1]
        out["ssrtrace"] = out["ssrtrace"][:res.niter + 1]
    del keep
    return out


def nls_large(n, p, start, fn=None, dfl=None, rowdata=None, algorithm="cgst", ctrl=None, weights=None,
              trace=False):
    """Run the oracle behind gsl_nls_large().  dfl(trans, theta, u, want_v, want_jtj) -> (v, JTJ)."""
    L = lib()
    ctrl = ctrl or control()
    prob = LargeProblem()
    keep = []
    if rowdata is not None:
        X = np.asfortranarray(np.asarray(rowdata["x"], dtype=np.float64).reshape(n, -1))
        Y = np.ascontiguousarray(rowdata["y"], dtype=np.float64)
        rd = RowData(n, X.shape[1], _dp(X), _dp(Y), rowdata["model"], p)
        keep += [X, Y, rd]
        prob.f = C.cast(L.gslref_model_f, F_T)
        prob.df = C.cast(L.gslref_model_dfl, DFL_T)
        prob.params = C.cast(C.pointer(rd), C.c_void_p)
    else:
        f_cb, _, _ = _wrap_callbacks(n, p, fn, None, None)

        def dfl_cb(trans, xp, up, _params, vp, jp):
            try:
                th = np.ctypeslib.as_array(xp, shape=(p,)).copy()
                u = None
                if up:
                    u = np.ctypeslib.as_array(up, shape=((n if trans else p),)).copy()
                v, JTJ = dfl(int(trans), th, u, bool(vp), bool(jp))
                if vp:
                    np.ctypeslib.as_array(vp, shape=((p if trans else n),))[:] = v
                if jp:
                    np.ctypeslib.as_array(jp, shape=(p * p,))[:] = np.asarray(JTJ).reshape(-1)
                return 0
            except Exception:  # noqa
                return 9
        d_cb = DFL_T(dfl_cb)
        keep += [f_cb, d_cb]
        prob.f, prob.df, prob.params = f_cb, d_cb, None
    prob.n, prob.p = n, p
    st = np.ascontiguousarray(start, dtype=np.float64)
    prob.start = _dp(st)
    w = None if weights is None else np.ascontiguousarray(weights, dtype=np.float64)
    prob.weights = _dp(w)
    # control_int / control_dbl of C_nls_large, R/nls_large.R:560-584
    ci = np.array([ctrl["maxiter"], int(trace), ALGORITHMS[algorithm], SCALES[ctrl["scale"]],
                   FDTYPES[ctrl["fdtype"]], -2, 0], dtype=np.int32)
    cd = np.array([ctrl["factor_up"], ctrl["factor_down"], ctrl["avmax"], ctrl["h_df"], ctrl["h_fvv"],
                   ctrl["xtol"], ctrl["ftol"], ctrl["gtol"]], dtype=np.float64)
    prob.control_int, prob.control_dbl = ci.ctypes.data_as(IP), _dp(cd)
    res = LargeResult()
    out = dict(par=np.zeros(p), covar=np.zeros((p, p), order="F"), resid=np.zeros(n))
    res.par, res.covar, res.resid = _dp(out["par"]), _dp(out["covar"]), _dp(out["resid"])
    maxiter = ctrl["maxiter"]
    if trace:
        out["partrace"] = np.full((maxiter + 1, p), np.nan, order="F")
        out["ssrtrace"] = np.full(maxiter + 1, np.nan)
        res.partrace, res.ssrtrace = _dp(out["partrace"]), _dp(out["ssrtrace"])
    status = L.gslref_nls_large(C.byref(prob), C.byref(res))
    out.update(niter=res.niter, conv=res.conv, status=STATUS.get(res.conv, str(res.conv)), ssr=res.ssr,
               ssrtol=res.ssrtol, neval=dict(f=res.neval[0], dfu=res.neval[1], df2=res.neval[2], fvv=res.neval[3]),
               info=res.info, chisq_init=res.chisq_init, ret=status)
    del keep, st, w
    return out


class device_exp:
    """with gslref.device_exp(): the oracle's row models evaluate exp with the device's arithmetic (gslref_models.c) --
    finite-difference runs of oracle and device then walk the same trajectory bit for bit"""

    def __enter__(self):
        lib().gslref_set_device_exp(1)
        return self

    def __exit__(self, *a):
        lib().gslref_set_device_exp(0)


def gexp(x):
    """the device's exp on a numpy array (for Python model closures handed to the oracle)"""
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    lib().gslref_device_exp_array(_dp(x.reshape(-1)), _dp(out.reshape(-1)), x.size)
    return out


def sobol(dim, npts, skip=0):
    out = np.zeros((npts, dim))
    lib().gslref_sobol(dim, skip, npts, _dp(out))
    return out


def halton(dim, npts, skip=0):
    out = np.zeros((npts, dim))
    lib().gslref_halton(dim, skip, npts, _dp(out))
    return out
