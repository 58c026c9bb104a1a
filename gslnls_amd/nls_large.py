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

    def __init__(self, model_id, p, x, y, weights=None, row_major=False, expr=None, parnames=None, xnames=None,
                 lowering="auto"):
        self.n = int(len(y))
        self.p = int(p)
        x = np.asarray(x, dtype=np.float64).reshape(self.n, -1)
        # row models: column-major n x nx; GLM family: row-major n x p
        self._x = np.ascontiguousarray(x) if (row_major or model_id == MODEL_GLMEXP) else np.asfortranarray(x)
        self._y = np.ascontiguousarray(y, dtype=np.float64)
        self._w = None if weights is None else np.ascontiguousarray(weights, dtype=np.float64)
        m = _lib.Model(int(model_id), self.p, x.shape[1], self._x.ctypes.data_as(C.c_void_p), 0)
        if int(model_id) == _lib.MODEL_EXPR:
            keep = _lib.set_expr(m, expr, list(parnames), list(xnames), lowering)  # noqa: F841
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
        if trace:
            # what callback_large and the summary block print (src/nls_large.c:259-273, :715-739), collected by the core
            import sys
            sys.stdout.write(_lib.trace_text())
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


class SparseLargeProblem(LargeProblem):
    """gsl_nls_large(fn = function, jac = function returning a sparse matrix): the callables stay on the host (as the
    R closures do in the reference, src/nls_large.c:426-653), all products with J run on the device.

    fn(theta) -> model values [n]; jac(theta) -> scipy.sparse csr (dgRMatrix) / csc (dgCMatrix) / coo (dgTMatrix)
    matrix or a dense n x p array (dgeMatrix / base matrix)."""

    def __init__(self, fn, jac, y, p, weights=None):
        import scipy.sparse as sp
        self.n, self.p = int(len(y)), int(p)
        self._y = np.ascontiguousarray(y, dtype=np.float64)
        self._w = None if weights is None else np.ascontiguousarray(weights, dtype=np.float64)
        self._keep = None
        self.error = None
        n, pp = self.n, self.p

        def f_cb(theta, p_, out, n_, _user):
            try:
                th = np.ctypeslib.as_array(theta, shape=(p_,)).copy()
                val = np.asarray(fn(th), dtype=np.float64).reshape(-1)
                if val.shape[0] != n_:
                    raise ValueError("fn returned %d values, expected %d" % (val.shape[0], n_))
                np.ctypeslib.as_array(out, shape=(n_,))[:] = val
                return 0
            except Exception as e:  # noqa: the error crosses the C boundary as a status
                self.error = e
                return 1

        def jac_cb(theta, p_, J, _user):
            try:
                th = np.ctypeslib.as_array(theta, shape=(p_,)).copy()
                M = jac(th)
                if not sp.issparse(M):
                    # a dense n x p array (base matrix / dgeMatrix in R): handed over as the block it is, column-major as R
                    # holds it (GSLNLS_SPARSE_DENSE) -- no index arrays are built anywhere
                    D = np.asfortranarray(np.asarray(M, dtype=np.float64).reshape(n, pp))
                    s = J.contents
                    s.format, s.nrow, s.ncol, s.nnz = 3, n, pp, n * pp
                    s.x = D.ctypes.data_as(DP)
                    self._keep = (D,)
                    return 0
                if M.shape != (n, pp):
                    raise ValueError("jac returned a %s matrix, expected %s" % (M.shape, (n, pp)))
                fmt = M.getformat()
                if fmt not in ("csr", "csc", "coo"):
                    M = M.tocsr()
                    fmt = "csr"
                x = np.ascontiguousarray(M.data, dtype=np.float64)
                s = J.contents
                s.nrow, s.ncol, s.nnz = n, pp, x.shape[0]
                s.x = x.ctypes.data_as(DP)
                if fmt == "coo":
                    a = np.ascontiguousarray(M.row, dtype=np.int32)
                    b = np.ascontiguousarray(M.col, dtype=np.int32)
                    s.format, s.i, s.j = 2, a.ctypes.data_as(IP), b.ctypes.data_as(IP)
                else:
                    a = np.ascontiguousarray(M.indptr, dtype=np.int32)
                    b = np.ascontiguousarray(M.indices, dtype=np.int32)
                    s.p = a.ctypes.data_as(IP)
                    if fmt == "csr":
                        s.format, s.j = 0, b.ctypes.data_as(IP)
                    else:
                        s.format, s.i = 1, b.ctypes.data_as(IP)
                self._keep = (x, a, b)  # valid until the next call, as the header asks
                return 0
            except Exception as e:  # noqa
                self.error = e
                return 1

        self._f_cb, self._jac_cb = _lib.LARGE_F_CB(f_cb), _lib.LARGE_JAC_CB(jac_cb)
        err = C.c_int(0)
        self._h = _lib.lib().gslnls_large_create_sparse(
            self.n, self.p, self._y.ctypes.data_as(C.c_void_p),
            None if self._w is None else self._w.ctypes.data_as(C.c_void_p), self._f_cb, self._jac_cb, None,
            C.byref(err))
        if not self._h:
            _lib.check(err.value)
            raise RuntimeError("gslnls_large_create_sparse failed: %s" % _lib.strerror(err.value))

    def solve(self, *a, **k):
        self.error = None
        fit = super().solve(*a, **k)
        if self.error is not None:  # an exception inside fn / jac aborted the fit (status EINVAL)
            raise self.error
        return fit


def gsl_nls_large(fn, data=None, start=None, algorithm="lm", control=None, trace=False, weights=None, y=None, A=None,
                  jac=None, lowering="auto"):
    """gsl_nls_large(fn = y ~ f(x, theta), data, start, algorithm = c("lm", ..., "cgst"), ...)  (R/nls_large.R:124)

    fn: formula string lowering to a registered row model; "glmexp" with A (n x p) and y; or a callable
    fn(theta) -> model values together with jac(theta) -> sparse / dense Jacobian and the response y
    (gsl_nls_large.function, R/nls_large.R:420-627).
    """
    if start is None:
        raise ValueError("starting values 'start' are required")
    if weights is not None and np.any(~(np.asarray(weights) > 0)):
        raise ValueError("missing or non-positive weights not allowed")
    if callable(fn):
        if jac is None or not callable(jac):
            raise NotImplementedError("gsl_nls_large(fn = function) needs jac = function (no finite differences on "
                                      "the large path, R/nls_large.R:470-480 requires it as well)")
        if y is None:
            raise ValueError("response 'y' is required with a function model")
        if isinstance(start, dict):
            names, st = list(start.keys()), np.array([float(np.atleast_1d(v)[0]) for v in start.values()])
        else:
            st = np.asarray(start, dtype=np.float64)
            names = ["par%d" % (i + 1) for i in range(st.shape[0])]
        prob = SparseLargeProblem(fn, jac, y, st.shape[0], weights)
        try:
            fit = prob.solve(st, algorithm, control, trace)
        finally:
            prob.close()
        fit["parnames"] = names
        return fit
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
    expr_kw = {}
    if low is None:
        # any other expression: compiled like in gsl_nls() (GSLNLS_MODEL_EXPR)
        xnames = [v for v in F.symbols(rhs) if v not in names and v != "pi"]
        if len(names) > 64 or len(xnames) > 8 or any(v not in data for v in xnames):
            raise NotImplementedError("formula RHS does not lower to the device: %s" % fn)
        mid, order = _lib.MODEL_EXPR, list(range(len(names)))
        expr_kw = dict(expr=fn.split("~", 1)[1].strip(), parnames=names, xnames=xnames, lowering=lowering)
    else:
        mid, order, xnames = low
    order = np.asarray(order)
    inv = np.argsort(order)
    yv = np.asarray(F.evaluate(lhs, {k: np.asarray(v, dtype=np.float64) for k, v in data.items()}), dtype=np.float64)
    X = (np.stack([np.asarray(data[c], dtype=np.float64) for c in xnames], axis=1) if xnames
         else np.zeros((len(yv), 0)))
    prob = LargeProblem(mid, len(names), X, yv, weights, **expr_kw)
    fit = prob.solve(vec[order], algorithm, control, trace)
    prob.close()
    fit["par"] = fit["par"][inv]
    fit["covar"] = np.asarray(fit["covar"])[np.ix_(inv, inv)]
    if trace:
        fit["partrace"] = fit["partrace"][:, inv]
    fit["parnames"] = names
    return fit
