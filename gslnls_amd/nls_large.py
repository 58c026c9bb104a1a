"""gsl_nls_large(): host-side mirror of R/nls_large.R:124-627 over the C ABI (gslnls_nls_large).

control_int[7] / control_dbl[8] are packed as R/nls_large.R:560-584 does; the
solver is always the normal-equations one (R/nls_large.R:547).  The model is a registered row model
(formula) or the dense GLM family exp(A theta) with the matrix resident in HBM.
"""
import ctypes as C

import numpy as np

from . import _lib
from . import formula as F
from .control import FDTYPES, LARGE_ALGORITHMS, SCALES, gsl_nls_control
from .nls import GslNlsFit, _normalise_start

DP, IP = _lib.DP, _lib.IP
MODEL_GLMEXP = 5


def pack_control_large(ctrl, algorithm="lm", trace=False):
    if algorithm not in LARGE_ALGORITHMS:
        raise ValueError("'algorithm' should be one of %s" % ", ".join(LARGE_ALGORITHMS))
    ci = np.array([ctrl["maxiter"], int(bool(trace)), LARGE_ALGORITHMS.index(algorithm), SCALES.index(ctrl["scale"]),
                   FDTYPES.index(ctrl["fdtype"]), -2, 0], dtype=np.int32)
    cd = np.array([ctrl["factor_up"], ctrl["factor_down"], ctrl["avmax"], ctrl["h_df"], ctrl["h_fvv"], ctrl["xtol"],
                   ctrl["ftol"], ctrl["gtol"]], dtype=np.float64)
    return ci, cd


class LargeProblem:
    """model + data resident in HBM for gsl_nls_large (gslnls_large_create)"""

    def __init__(self, model_id, p, x, y, weights=None, row_major=False):
        self.n = int(len(y))
        self.p = int(p)
        x = np.asarray(x, dtype=np.float64).reshape(self.n, -1)
        # row models: column-major n x nx; GLM family: row-major n x p
        self._x = np.ascontiguousarray(x) if (row_major or model_id == MODEL_GLMEXP) else np.asfortranarray(x)
        self._y = np.ascontiguousarray(y, dtype=np.float64)
        self._w = None if weights is None else np.ascontiguousarray(weights, dtype=np.float64)
        m = _lib.Model(int(model_id), self.p, x.shape[1], self._x.ctypes.data_as(C.c_void_p), 0)
        err = C.c_int(0)
        self._h = _lib.lib().gslnls_large_create(C.byref(m), self._y.ctypes.data_as(C.c_void_p), self.n,
                                                 None if self._w is None else self._w.ctypes.data_as(C.c_void_p),
                                                 C.byref(err))
        if not self._h:
            _lib.check(err.value)
            raise RuntimeError("gslnls_large_create failed: %s" % _lib.strerror(err.value))

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().gslnls_large_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa
            pass

    def solve(self, start, algorithm="cgst", control=None, trace=False, want_resid=True):
        ctrl = control if (control is not None and len(control) >= 23) else gsl_nls_control(**(control or {}))
        ci, cd = pack_control_large(ctrl, algorithm, trace)
        p, n = self.p, self.n
        st = np.ascontiguousarray(start, dtype=np.float64)
        out = dict(par=np.zeros(p), covar=np.zeros((p, p), order="F"))
        res = _lib.LargeResult()
        res.par, res.covar = out["par"].ctypes.data_as(DP), out["covar"].ctypes.data_as(DP)
        if want_resid:
            out["resid"] = np.zeros(n)
            res.resid = out["resid"].ctypes.data_as(DP)
        if trace:
            out["partrace"] = np.full((ctrl["maxiter"] + 1, p), np.nan, order="F")
            out["ssrtrace"] = np.full(ctrl["maxiter"] + 1, np.nan)
            res.partrace, res.ssrtrace = out["partrace"].ctypes.data_as(DP), out["ssrtrace"].ctypes.data_as(DP)
        rc = _lib.lib().gslnls_large_solve(self._h, st.ctypes.data_as(DP), ci.ctypes.data_as(IP),
                                           cd.ctypes.data_as(DP), C.byref(res))
        _lib.check(rc)
        out.update(niter=res.niter, conv=res.conv, status=_lib.strerror(res.conv), ssr=res.ssr, ssrtol=res.ssrtol,
                   chisq_init=res.chisq_init, info=res.info, n=n,
                   algorithm=_lib.lib().gslnls_algorithm_name(LARGE_ALGORITHMS.index(algorithm)).decode(),
                   neval=dict(f=res.neval[0], dfu=res.neval[1], df2=res.neval[2], fvv=res.neval[3]),
                   n_passes=res.n_passes, last_pass_ms=res.last_pass_ms)
        if trace:
            out["partrace"] = out["partrace"][:res.niter + 1]
            out["ssrtrace"] = out["ssrtrace"][:res.niter + 1]
        return GslNlsFit(out)

    def time_pass(self, mode, x, u=None, reps=20):
        x = np.ascontiguousarray(x, dtype=np.float64)
        u = np.ascontiguousarray(u if u is not None else x, dtype=np.float64)
        return float(_lib.lib().gslnls_large_time_pass(self._h, int(mode), x.ctypes.data_as(DP), u.ctypes.data_as(DP),
                                                       int(reps)))


def gsl_nls_large(fn, data=None, start=None, algorithm="lm", control=None, trace=False, weights=None, y=None, A=None):
    """gsl_nls_large(fn = y ~ f(x, theta), data, start, algorithm = c("lm", ..., "cgst"), ...)  (R/nls_large.R:124)

    fn: formula string lowering to a registered row model, or "glmexp" with A (n x p) and y.
    """
    if start is None:
        raise ValueError("starting values 'start' are required")
    if weights is not None and np.any(~(np.asarray(weights) > 0)):
        raise ValueError("missing or non-positive weights not allowed")
    if fn == "glmexp":
        A = np.asarray(A, dtype=np.float64)
        st = np.asarray(start, dtype=np.float64)
        prob = LargeProblem(MODEL_GLMEXP, A.shape[1], A, y, weights)
        fit = prob.solve(st, algorithm, control, trace)
        prob.close()
        fit["parnames"] = ["x%d" % (i + 1) for i in range(A.shape[1])]
        return fit
    names, vec, mat, _ = _normalise_start(start)
    if mat is not None:
        raise ValueError("gsl_nls_large has no multi-start")
    lhs, rhs = F.parse_formula(fn)
    low = F.lower(rhs, names)
    if low is None:
        raise NotImplementedError("formula RHS does not match a registered device model: %s" % fn)
    mid, order, xnames = low
    order = np.asarray(order)
    inv = np.argsort(order)
    yv = np.asarray(F.evaluate(lhs, {k: np.asarray(v, dtype=np.float64) for k, v in data.items()}), dtype=np.float64)
    X = np.stack([np.asarray(data[c], dtype=np.float64) for c in xnames], axis=1)
    prob = LargeProblem(mid, len(names), X, yv, weights)
    fit = prob.solve(vec[order], algorithm, control, trace)
    prob.close()
    fit["par"] = fit["par"][inv]
    fit["covar"] = np.asarray(fit["covar"])[np.ix_(inv, inv)]
    if trace:
        fit["partrace"] = fit["partrace"][:, inv]
    fit["parnames"] = names
    return fit
