"""gsl_nls(): host-side mirror of the reference's R entry point (R/nls.R:306-1083).

It does what gsl_nls.formula does before and after `.Call(C_nls, ...)`:
normalise start values / ranges (R/nls.R:399-437), bounds (:539-559) and weights
(:489-495), pack the control vectors (:693-713), call the C ABI, and wrap the returned
list (:723-762).  The call itself goes to libgslnls_hip.so -- the numeric work runs on
the MI355X; nothing here computes any part of the fit.
"""
import ctypes as C
import math

import numpy as np

from . import _lib
from . import formula as F
from .control import ALGORITHMS, LOSSES, gsl_nls_control, gsl_nls_loss, pack_control

DP = _lib.DP
IP = _lib.IP


def _dp(a):
    return None if a is None else a.ctypes.data_as(DP)


class GslNlsFit(dict):
    """The list C_nls returns (src/nls.c:636-645) plus the few accessors the reference's
    tests use (coef, deviance, sigma, df.residual; R/nls_methods.R)."""

    def coef(self):
        return dict(zip(self["parnames"], self["par"]))

    def deviance(self):
        return float(self["ssr"])

    def df_residual(self):
        return int(self["n"] - len(self["par"]))

    def sigma(self):
        return math.sqrt(self.deviance() / self.df_residual())

    def vcov(self):
        return self.sigma() ** 2 * np.asarray(self["covar"])

    @property
    def isConv(self):
        return self["conv"] == 0


def _normalise_start(start, parnames_hint=None):
    """R/nls.R:399-437: vector / list of scalars -> single start; list with length-2 entries or
    2 x p matrix -> multi-start ranges with has_start; NA/inf -> (-0.1, 0.75)."""
    if isinstance(start, dict):
        names = list(start.keys())
        vals = [np.atleast_1d(np.asarray(v, dtype=np.float64)) for v in start.values()]
        if any(len(v) > 1 for v in vals):
            if not all(len(v) in (1, 2) for v in vals):
                raise ValueError("List elements of 'start' must be of length 1 or 2 to specify (multi-start) "
                                 "parameter values or ranges")
            mat = np.stack([np.repeat(v, 2) if len(v) == 1 else v for v in vals], axis=1)  # 2 x p
            vec = None
        else:
            vec = np.array([v[0] for v in vals])
            mat = None
    else:
        arr = np.asarray(start, dtype=np.float64)
        names = list(parnames_hint) if parnames_hint is not None else ["par%d" % (i + 1) for i in range(arr.shape[-1])]
        if arr.ndim == 2:
            if arr.shape[0] != 2:
                raise ValueError("Matrix 'start' must have exactly 2 rows to specify (multi-start) parameter ranges")
            mat, vec = arr.copy(), None
        else:
            mat, vec = None, arr.copy()
    has_start = None
    if mat is None:
        na = ~np.isfinite(vec)
        if na.any():
            mat = np.stack([np.where(na, -0.1, vec), np.where(na, 0.75, vec)], axis=0)
            has_start = np.stack([~na, ~na], axis=0)
    else:
        na = ~np.isfinite(mat)
        mat[0, na[0]] = -0.1
        mat[1, na[1]] = 0.75
        has_start = ~na
    if mat is not None:
        if np.any(mat[0] > mat[1]):
            raise ValueError("Multi-start parameter lower bounds cannot be larger than upper bounds")
        if not np.any(mat[0] < mat[1]):
            vec, mat = mat[0].copy(), None  # degenerate ranges: single start
    return names, vec, mat, has_start


def _bounds(lower, upper, names):
    """R/nls.R:539-559 -> 2 x p column-major [lower, upper] pairs, or None"""
    if lower is None and upper is None:
        return None
    p = len(names)

    def expand(b, fill):
        if b is None:
            return np.full(p, fill)
        if isinstance(b, dict):
            out = np.full(p, fill)
            for k, v in b.items():
                out[names.index(k)] = v
            return out
        b = np.atleast_1d(np.asarray(b, dtype=np.float64))
        return np.full(p, b[0]) if b.size == 1 else b
    lo, up = expand(lower, -np.inf), expand(upper, np.inf)
    if np.all(np.isinf(lo) & (lo < 0)) and np.all(np.isinf(up) & (up > 0)):
        return None  # if(all(is.infinite(.lupars))) .lupars <- NULL (R/nls.R:558-560)
    return np.ascontiguousarray(np.stack([lo, up], axis=1).reshape(-1))


def _ranges_inside_bounds(mat, has_start, lo_b, up_b):
    """R/nls.R:545-557: missing range ends are moved inside the bounds first, then every range has to lie within them"""
    if has_start is not None:
        m0, m1 = ~has_start[0], ~has_start[1]
        if m0.any():
            old = mat[0, m0].copy()
            mat[0, m0] = np.maximum(old, lo_b[m0])
            mat[1, m0] = mat[1, m0] + (mat[0, m0] - old)
        if m1.any():
            mat[1, m1] = np.minimum(mat[1, m1], up_b[m1])
    if np.any(mat[0] < lo_b) or np.any(mat[1] > up_b):
        raise ValueError("Starting parameter ranges must be contained within 'lower' and 'upper' bounds")


class DenseProblem:
    """A model + data resident in HBM (gslnls_dense_create): upload once, solve many times."""

    def __init__(self, model_id, p, x, y, weights=None, expr=None, parnames=None, xnames=None, lowering="auto"):
        """model_id: a registered device model, or _lib.MODEL_EXPR with `expr` (formula right-hand side),
        `parnames` (order of the start vector) and `xnames` (order of the columns of x)"""
        L = _lib.lib()
        x = np.asarray(x, dtype=np.float64)
        self.n = int(np.asarray(y).shape[0])
        self._x = np.asfortranarray(x.reshape(self.n, -1))
        self._y = np.ascontiguousarray(y, dtype=np.float64)
        self._sw = None if weights is None else np.ascontiguousarray(np.sqrt(np.asarray(weights, dtype=np.float64)))
        self.p, self.model_id = int(p), int(model_id)
        m = _lib.Model(self.model_id, self.p, self._x.shape[1], self._x.ctypes.data_as(C.c_void_p), 0)
        if self.model_id == _lib.MODEL_EXPR:
            keep = _lib.set_expr(m, expr, list(parnames), list(xnames), lowering)  # noqa: F841
        err = C.c_int(0)
        self._h = L.gslnls_dense_create(C.byref(m), self._y.ctypes.data_as(C.c_void_p), self.n,
                                        None if self._sw is None else self._sw.ctypes.data_as(C.c_void_p),
                                        C.byref(err))
        if not self._h:
            _lib.check(err.value)
            raise RuntimeError("gslnls_dense_create failed: %s" % _lib.strerror(err.value))

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().gslnls_dense_destroy(self._h)
            self._h = None

    def diagnostics(self, par, jac=True, control=None):
        """hat values and Cook's distances at `par` (hatvalues.gsl_nls / cooks.distance.gsl_nls of the reference,
        R/nls_methods.R, computed on the device from the resident data): returns (hat[n], cooks[n])"""
        ctrl = control if (control is not None and len(control) >= 23) else gsl_nls_control(**(control or {}))
        ci, cd = pack_control(ctrl, "lm", False, True, False)
        th = np.ascontiguousarray(par, dtype=np.float64)
        hat, cooks = np.zeros(self.n), np.zeros(self.n)
        rc = _lib.lib().gslnls_dense_diagnostics(self._h, int(bool(jac)), _dp(th), ci.ctypes.data_as(IP), _dp(cd),
                                                 _dp(hat), _dp(cooks))
        _lib.check(rc)
        if rc != 0:
            raise RuntimeError("diagnostics failed: %s" % _lib.strerror(rc))
        return hat, cooks

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa
            pass

    def solve(self, start, jac=False, fvv=False, algorithm="lm", control=None, trace=False, lupars=None,
              want_vectors=True, chunk=0):
        L = _lib.lib()
        ctrl = control or gsl_nls_control()
        ci, cd = pack_control(ctrl, algorithm, trace)
        p, n = self.p, self.n
        st = np.ascontiguousarray(start, dtype=np.float64)
        out = dict(par=np.zeros(p), covar=np.zeros((p, p), order="F"))
        res = _lib.Result()
        res.par, res.covar = _dp(out["par"]), _dp(out["covar"])
        if want_vectors:
            out["resid"] = np.zeros(n)
            out["grad"] = np.zeros((n, p), order="F")
            res.resid, res.grad = _dp(out["resid"]), _dp(out["grad"])
        if trace:
            out["partrace"] = np.full((ctrl["maxiter"] + 1, p), np.nan, order="F")
            out["ssrtrace"] = np.full(ctrl["maxiter"] + 1, np.nan)
            res.partrace, res.ssrtrace = _dp(out["partrace"]), _dp(out["ssrtrace"])
        rc = L.gslnls_dense_solve(self._h, int(bool(jac)), int(bool(fvv)), _dp(st), _dp(lupars),
                                  ci.ctypes.data_as(IP), _dp(cd), int(chunk), C.byref(res))
        _lib.check(rc)
        return _finish(out, res, trace, algorithm, n)

    def time_pass(self, theta, jac=True, reps=200):
        th = np.ascontiguousarray(theta, dtype=np.float64)
        return float(_lib.lib().gslnls_dense_time_pass(self._h, int(bool(jac)), _dp(th), int(reps)))


def _print_trace(trace):
    """trace = TRUE: what the reference prints while it runs (iteration lines, multi-start / IRLS progress, the summary
    block; src/nls.c:610-630, :980-995) -- collected by the core, printed here once the call is back, as the R shim
    does with Rprintf"""
    if trace:
        import sys
        sys.stdout.write(_lib.trace_text())


def _finish(out, res, trace, algorithm, n):
    out.update(niter=res.niter, conv=res.conv, status=_lib.strerror(res.conv), ssr=res.ssr, ssrtol=res.ssrtol,
               algorithm=_lib.lib().gslnls_algorithm_name(ALGORITHMS.index(algorithm)).decode(),
               neval=dict(f=res.neval[0], J=res.neval[1], fvv=res.neval[2]), info=res.info,
               chisq_init=res.chisq_init, loop_ms=res.loop_ms, n_launches=res.n_launches, n_steps=res.n_steps, code_path=res.code_path, n=n,
               jtj_cond=res.jtj_cond,
               irls=dict(irls_sigma=res.irls_sigma, irls_tol=res.irls_tol, irls_status=res.irls_status,
                         irls_niter=res.irls_niter, irls_conv=res.irls_status),
               mstart=dict(nsp=res.mstart_nsp, nwsp=res.mstart_nwsp, iters=res.mstart_iters,
                           stop=res.mstart_stop, ssropt=res.mstart_ssropt))
    if trace:
        out["partrace"] = out["partrace"][:res.niter + 1]
        out["ssrtrace"] = out["ssrtrace"][:res.niter + 1]
    return GslNlsFit(out)


def _gsl_nls_function(fn, y, start, algorithm, control, jac, fvv, trace, weights, lower, upper, loss, parnames=None):
    """gsl_nls.function (R/nls.R:778-1060): `fn(par)` returns the n model values -- or (values, gradient), the analogue of
    the "gradient" attribute README example 4 uses --, `y` the response, `jac` / `fvv` optional functions jac(par) -> n x p,
    fvv(par, v) -> n.  The closures run on this thread, where the reference runs them (src/nls.c:815-978); every n x p and
    p x p operation of the fit runs on the device (csrc/bd_host.hpp).  Any p <= 4096."""
    if y is None:
        raise ValueError("'y' is required when 'fn' is a function")
    if algorithm not in ALGORITHMS:
        raise ValueError("'algorithm' should be one of %s" % ", ".join(ALGORITHMS))
    loss_cfg = gsl_nls_loss(loss) if isinstance(loss, str) else gsl_nls_loss(**loss)
    ctrl = control if (control is not None and len(control) >= 23) else gsl_nls_control(**(control or {}))
    # start values, ranges or missing values (NaN): R/nls.R:399-437 for functions as for formulas (R/nls.R:842-880)
    names, vec, mat, has_start = _normalise_start(start, parnames)
    p = len(names)
    yv = np.ascontiguousarray(np.asarray(y, dtype=np.float64).reshape(-1))
    n = len(yv)
    first = fn((vec if mat is None else 0.5 * (mat[0] + mat[1])).copy())
    grad_in_fn = isinstance(first, tuple)
    if jac is True and not grad_in_fn:
        raise ValueError("jac = True needs a model function that returns (values, gradient)")
    jac_fn = jac if callable(jac) else ((lambda th: fn(th)[1]) if (grad_in_fn and jac is not False) else None)
    fvv_fn = fvv if callable(fvv) else None
    errors = []

    def f_cb(theta, pp, out, nn, _user):
        try:
            v = fn(np.ctypeslib.as_array(theta, shape=(pp,)).copy())
            v = np.asarray(v[0] if isinstance(v, tuple) else v, dtype=np.float64).reshape(-1)
            if v.size != nn:
                return 1
            np.ctypeslib.as_array(out, shape=(nn,))[:] = v
            return 0
        except Exception as e:  # noqa: BLE001 -- a Python exception must not unwind through the C frames
            errors.append(e)
            return 1

    def jac_cb(theta, pp, out, nn, _user):
        try:
            Jm = jac_fn(np.ctypeslib.as_array(theta, shape=(pp,)).copy())
            if hasattr(Jm, "toarray"):  # (a scipy sparse matrix: asarray would wrap it in a 0-d object array)
                Jm = Jm.toarray()
            Jm = np.asarray(Jm, dtype=np.float64)
            if Jm.shape != (nn, pp):
                return 1
            # (two sequential copies on purpose: the core's buffer is pinned host memory, and a transposing assignment
            # writes it 8 bytes at a time in a scattered order -- measured 57 ms per Jacobian instead of 0.5)
            np.ctypeslib.as_array(out, shape=(nn * pp,))[:] = np.asfortranarray(Jm).reshape(-1, order="F")
            return 0
        except Exception as e:  # noqa: BLE001
            errors.append(e)
            return 1

    def fvv_cb(theta, v, pp, out, nn, _user):
        try:
            r = np.asarray(fvv_fn(np.ctypeslib.as_array(theta, shape=(pp,)).copy(),
                                  np.ctypeslib.as_array(v, shape=(pp,)).copy()), dtype=np.float64).reshape(-1)
            if r.size != nn:
                return 1
            np.ctypeslib.as_array(out, shape=(nn,))[:] = r
            return 0
        except Exception as e:  # noqa: BLE001
            errors.append(e)
            return 1

    sw = None
    if weights is not None:
        w = np.asarray(weights, dtype=np.float64)
        if w.ndim != 1 or len(w) != n or np.any(~(w > 0)):
            raise ValueError("weights of a function model: a vector of n positive values")
        sw = np.ascontiguousarray(np.sqrt(w))
    lu = _bounds(lower, upper, names)
    if lu is not None:
        lo_b, up_b = lu.reshape(p, 2)[:, 0], lu.reshape(p, 2)[:, 1]
        if np.any(lo_b > up_b):
            raise ValueError("Parameter lower bounds cannot be larger than upper bounds")
        if mat is not None:
            _ranges_inside_bounds(mat, has_start, lo_b, up_b)
        elif np.any(vec < lo_b) or np.any(vec > up_b):
            raise ValueError("Starting parameters must be contained within 'lower' and/or 'upper' bounds")
    any_missing = bool(has_start is not None and not np.all(has_start))
    ci, cd = pack_control(ctrl, algorithm, trace, True, any_missing)
    out = dict(par=np.zeros(p), covar=np.zeros((p, p), order="F"), resid=np.zeros(n), grad=np.zeros((n, p), order="F"))
    res = _lib.Result()
    res.par, res.covar, res.resid, res.grad = _dp(out["par"]), _dp(out["covar"]), _dp(out["resid"]), _dp(out["grad"])
    if loss_cfg["rho"] != "default":
        for k in ("irls_weights", "irls_psi", "irls_dpsi"):
            out[k] = np.zeros(n)
            setattr(res, k, _dp(out[k]))
    cc = np.asarray(list(loss_cfg["cc"].values()) or [0.0], dtype=np.float64)
    if trace:
        out["partrace"] = np.full((ctrl["maxiter"] + 1, p), np.nan, order="F")
        out["ssrtrace"] = np.full(ctrl["maxiter"] + 1, np.nan)
        res.partrace, res.ssrtrace = _dp(out["partrace"]), _dp(out["ssrtrace"])
    # (a function-pointer type called without arguments is the NULL pointer)
    cbs = (_lib.FN_CB(f_cb), _lib.JAC_CB(jac_cb) if jac_fn else _lib.JAC_CB(), _lib.FVV_CB(fvv_cb) if fvv_fn else _lib.FVV_CB())
    if mat is not None:
        # start ranges: the multi-start procedure with the closures as the model (src/nls.c:274-532)
        st = np.ascontiguousarray(mat.T.reshape(-1))
        hs = np.ascontiguousarray(has_start.T.reshape(-1).astype(np.int32))
        rc = _lib.lib().gslnls_nls_fn_mstart(n, p, yv.ctypes.data_as(C.c_void_p), cbs[0], cbs[1], cbs[2], None, _dp(st),
                                             hs.ctypes.data_as(IP), None if sw is None else sw.ctypes.data_as(C.c_void_p),
                                             _dp(lu), ci.ctypes.data_as(IP), _dp(cd), LOSSES.index(loss_cfg["rho"]), _dp(cc),
                                             C.byref(res))
    else:
        st = np.ascontiguousarray(vec, dtype=np.float64)
        rc = _lib.lib().gslnls_nls_fn_loss(n, p, yv.ctypes.data_as(C.c_void_p), cbs[0], cbs[1], cbs[2], None, _dp(st),
                                           None if sw is None else sw.ctypes.data_as(C.c_void_p), _dp(lu), ci.ctypes.data_as(IP),
                                           _dp(cd), LOSSES.index(loss_cfg["rho"]), _dp(cc), C.byref(res))
    if errors:
        raise errors[0]
    _lib.check(rc)
    _print_trace(trace)
    fit = _finish(out, res, trace, algorithm, n)
    fit["solver_served"] = bool(_lib.lib().gslnls_solver_served(ci.ctypes.data_as(IP), C.byref(res)))
    fit["parnames"] = names
    fit["weights"] = weights
    return fit


def gsl_nls(fn, data=None, start=None, algorithm="lm", control=None, jac=None, fvv=None, trace=False,
            weights=None, lower=None, upper=None, loss="default", y=None, lowering="auto"):
    """gsl_nls(fn = y ~ f(x, theta), data, start, ...) -- same arguments as R/nls.R:306-316.

    fn      : model formula string 'y ~ rhs': a RHS that matches a hand-written device model
              (gslnls_amd/formula.py) uses it, any other expression is compiled (value + symbolic gradient);
              or an int registry id together with data = {'x': ..., 'y': ...}
    jac/fvv : True -> analytic derivatives on device (R: symbolic stats::deriv), None/False -> FD
    lowering: compiled expressions only: "vm" interpreter; "jit" native code, compiled in process by the HIP runtime's
              own compiler (hiprtc) before the fit, cached on disk; "auto" = interpreter for the first fit of a formula
              while the native code is built on a background thread, native code for the fits after that.
              fit["code_path"] says which one ran (0 hand-written model, 1 interpreter, 2 native, 3 native wide path)
    """
    if start is None:
        raise ValueError("starting values 'start' are required")
    if callable(fn):
        return _gsl_nls_function(fn, y, start, algorithm, control, jac, fvv, trace, weights, lower, upper, loss)
    _ = ALGORITHMS.index(algorithm) if algorithm in ALGORITHMS else (_ for _ in ()).throw(
        ValueError("'algorithm' should be one of %s" % ", ".join(ALGORITHMS)))
    ctrl = control if (control is not None and len(control) >= 23) else gsl_nls_control(**(control or {}))
    loss_cfg = gsl_nls_loss(loss) if isinstance(loss, str) else gsl_nls_loss(**loss)
    names, vec, mat, has_start = _normalise_start(start)
    p = len(names)

    if isinstance(fn, str):
        lhs, rhs = F.parse_formula(fn)
        if lhs is None:
            raise ValueError("formula needs a left-hand side")
        low = F.lower(rhs, names)
        if low is None:
            # not one of the hand-written device models: compile the expression itself
            # (GSLNLS_MODEL_EXPR, csrc/expr_compile.hpp -- the analogue of eval(formula[[3]], ...) and
            # stats::deriv(), R/nls.R:565,588-599)
            xnames = [v for v in F.symbols(rhs) if v not in names and v != "pi"]
            missing = [v for v in xnames if v not in data]
            if missing:
                raise ValueError("formula symbols %s are neither parameters nor data columns" % missing)
            # (more than 512 parameters or 8 data columns: the core refuses the expression -- GSLNLS_E_UNSUPPORTED, before any
            # work -- and the closure route below serves the fit, as for a right-hand side it cannot differentiate.  The
            # in-process compiler does take longer formulas, but not in a time a caller would wait for: measured, round 5,
            # p = 750: 85 s for the first fit)
            mid, order = _lib.MODEL_EXPR, list(range(p))
            expr_text = fn.split("~", 1)[1].strip()
        else:
            mid, order, xnames = low
        yv = np.asarray(F.evaluate(lhs, {k: np.asarray(v, dtype=np.float64) for k, v in data.items()}),
                        dtype=np.float64)
        X = (np.stack([np.asarray(data[c], dtype=np.float64) for c in xnames], axis=1) if xnames
             else np.zeros((len(yv), 0)))
    else:
        mid, order = int(fn), list(range(p))
        yv = np.asarray(data["y"] if y is None else y, dtype=np.float64)
        X = np.asarray(data["x"], dtype=np.float64).reshape(len(yv), -1)
    n = len(yv)
    if X.shape[0] != n:
        raise ValueError("data columns differ in length")
    order = np.asarray(order)
    inv = np.argsort(order)

    sw = None
    sw_is_matrix = 0
    if weights is not None:
        w = np.asarray(weights, dtype=np.float64)
        if w.ndim == 2:
            sw = np.asfortranarray(np.linalg.cholesky(w))  # t(chol(W)), R/nls.R:489-495
            sw_is_matrix = 1
        else:
            if len(w) != n or np.any(~(w > 0)):
                raise ValueError("missing or non-positive weights not allowed")
            sw = np.ascontiguousarray(np.sqrt(w))
    lu = _bounds(lower, upper, names)
    if lu is not None:
        # the reference's checks, R/nls.R:542-557
        lo_b, up_b = lu.reshape(p, 2)[:, 0], lu.reshape(p, 2)[:, 1]
        if np.any(lo_b > up_b):
            raise ValueError("Parameter lower bounds cannot be larger than upper bounds")
        if mat is not None:
            _ranges_inside_bounds(mat, has_start, lo_b, up_b)
        elif np.any(vec < lo_b) or np.any(vec > up_b):
            raise ValueError("Starting parameters must be contained within 'lower' and/or 'upper' bounds")
        lu = np.ascontiguousarray(lu.reshape(p, 2)[order].reshape(-1))

    any_missing = bool(has_start is not None and not np.all(has_start))
    ci, cd = pack_control(ctrl, algorithm, trace, True, any_missing)
    if mat is not None:
        st = np.ascontiguousarray(mat[:, order].T.reshape(-1))
        hs = np.ascontiguousarray(has_start[:, order].T.reshape(-1).astype(np.int32))
    else:
        st = np.ascontiguousarray(vec[order])
        hs = np.ones(2 * p, dtype=np.int32)
    cc = np.asarray(list(loss_cfg["cc"].values()) or [0.0], dtype=np.float64)

    Xf = np.asfortranarray(X)
    yc = np.ascontiguousarray(yv)
    m = _lib.Model(mid, p, Xf.shape[1], Xf.ctypes.data_as(C.c_void_p), 0)
    if mid == _lib.MODEL_EXPR:
        if not isinstance(fn, str):
            raise ValueError("model id %d needs a formula" % mid)
        keep = _lib.set_expr(m, expr_text, names, xnames, lowering)  # noqa: F841 (keeps the C strings alive)
    out = dict(par=np.zeros(p), covar=np.zeros((p, p), order="F"), resid=np.zeros(n),
               grad=np.zeros((n, p), order="F"))
    res = _lib.Result()
    res.par, res.covar, res.resid, res.grad = _dp(out["par"]), _dp(out["covar"]), _dp(out["resid"]), _dp(out["grad"])
    if loss_cfg["rho"] != "default":
        for k in ("irls_weights", "irls_psi", "irls_dpsi"):
            out[k] = np.zeros(n)
            setattr(res, k, _dp(out[k]))
    if trace:
        out["partrace"] = np.full((ctrl["maxiter"] + 1, p), np.nan, order="F")
        out["ssrtrace"] = np.full(ctrl["maxiter"] + 1, np.nan)
        res.partrace, res.ssrtrace = _dp(out["partrace"]), _dp(out["ssrtrace"])
    if trace:
        od = np.ascontiguousarray(order, dtype=np.int32)
        _lib.lib().gslnls_trace_set_order(od.ctypes.data_as(IP), p)  # printed vectors in the caller's parameter order
    rc = _lib.lib().gslnls_nls(C.byref(m), yc.ctypes.data_as(C.c_void_p), n, int(bool(jac)), int(bool(fvv)),
                               _dp(st), int(mat is not None), None if sw is None else sw.ctypes.data_as(C.c_void_p),
                               sw_is_matrix, _dp(lu), ci.ctypes.data_as(IP), _dp(cd), hs.ctypes.data_as(IP),
                               LOSSES.index(loss_cfg["rho"]), _dp(cc), C.byref(res))
    if (rc == _lib.E_UNSUPPORTED and mid == _lib.MODEL_EXPR and not sw_is_matrix and ALGORITHMS.index(algorithm) <= 1
            and lowering == "auto"):  # (an explicit "vm" / "jit" asks for that form of the formula's device code or nothing)
        # The core cannot lower this right-hand side (a function outside stats::deriv's table -- ifelse, pmax, ... --, a
        # comparison, too long a program).  The reference never lowers anything: its .fn closure evaluates the expression
        # (R/nls.R:565) and, where stats::deriv fails, jac stays NULL with a warning (R/nls.R:588-599).  The same here:
        # the expression evaluated by this module's own evaluator becomes the model closure of the callback route
        # (gslnls_nls_fn_loss / gslnls_nls_fn_mstart: every n x p and p x p operation still on the device).
        import warnings
        cols = {k: np.asarray(v, dtype=np.float64) for k, v in data.items()}

        def closure(th, _rhs=rhs, _names=names):
            env = dict(cols)
            env.update(zip(_names, th))
            v = np.asarray(F.evaluate(_rhs, env), dtype=np.float64)
            return np.broadcast_to(v, (n,)).copy() if v.shape != (n,) else v
        if jac:
            warnings.warn("failed to symbolically derive 'jac': the model expression is not in the derivatives table; "
                          "finite differences are used")
        if fvv:
            warnings.warn("failed to symbolically derive 'fvv': the model expression is not in the derivatives table; "
                          "finite differences are used")
        fit = _gsl_nls_function(closure, yv, start, algorithm, ctrl, None, None, trace, weights, lower, upper, loss,
                                parnames=names)
        fit["lowered"] = False
        return fit
    _lib.check(rc)
    _print_trace(trace)
    fit = _finish(out, res, trace, algorithm, n)
    # solver routing rule of the boundary (include/gslnls_core.h): a "qr" / "svd" request is served on the normal
    # equations only while the scaled condition number allows it; the R shim re-runs such a fit through GSL, this
    # mirror has no GSL to fall back to and says so
    fit["solver_served"] = bool(_lib.lib().gslnls_solver_served(ci.ctypes.data_as(IP), C.byref(res)))
    if not fit["solver_served"] and res.conv in (0, 11):
        import warnings
        warnings.warn("solver=%r requested but kappa(S J'J S) = %.3g exceeds %.0e: the normal-equations result may "
                      "carry fewer than 6 digits (the R binding falls through to GSL here)"
                      % (ctrl["solver"], res.jtj_cond, 1e10))
    # back to the caller's parameter order (nothing to do for expression models, whose order is the caller's: at p = 501,
    # n = 20000 the gather of the 80 MB gradient alone was a third of the call)
    if not np.array_equal(inv, np.arange(p)):
        fit["par"] = fit["par"][inv]
        fit["covar"] = np.asarray(fit["covar"])[np.ix_(inv, inv)]
        fit["grad"] = np.asarray(fit["grad"])[:, inv]
        if trace:
            fit["partrace"] = fit["partrace"][:, inv]
    fit["parnames"] = names
    fit["weights"] = weights
    return fit
