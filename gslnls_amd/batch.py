"""Batched robust fits: B independent data sets through gsl_nls(loss = ...)'s procedure, one workgroup per
data set (BASELINE config C5).  Thin ctypes front of gslnls_batch_* (include/gslnls_core.h)."""
import ctypes as C

import numpy as np

from . import _lib
from .control import LOSSES, gsl_nls_control, gsl_nls_loss, pack_control

DP, IP = _lib.DP, _lib.IP


class BatchProblem:
    def __init__(self, model_id, p, x, y, weights=None):
        """x: (B, n) or (B, nx, n); y: (B, n); weights: (B, n) or None"""
        y = np.ascontiguousarray(y, dtype=np.float64)
        self.B, self.n = y.shape
        x = np.asarray(x, dtype=np.float64)
        if self.B == 0:
            # an empty block of a sharded job (more ranks than data sets): the handle only joins the final all-gather
            x = x.reshape(0, x.shape[1] if x.ndim == 3 else 1, self.n)
        x = np.ascontiguousarray(x.reshape(self.B, -1, self.n)) if self.B else x
        self.p, self.model_id, self.nx = int(p), int(model_id), x.shape[1]
        sw = None if weights is None else np.ascontiguousarray(np.sqrt(np.asarray(weights, dtype=np.float64)))
        err = C.c_int(0)
        self._h = _lib.lib().gslnls_batch_create(self.model_id, self.p, self.nx, x.ctypes.data_as(C.c_void_p),
                                                 y.ctypes.data_as(C.c_void_p),
                                                 None if sw is None else sw.ctypes.data_as(C.c_void_p), self.n, self.B,
                                                 C.byref(err))
        if not self._h:
            _lib.check(err.value)
            raise RuntimeError("gslnls_batch_create failed: %s" % _lib.strerror(err.value))

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().gslnls_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa
            pass

    def irls_gathered(self, B_total, start, **kw):
        """One process per GPU: this object holds the calling rank's contiguous block of the B_total data sets
        (rank r: [r * ceil(B_total / world), ...)); every rank fits its block, one all-gather completes the outputs on
        every rank (gslnls_batch_irls_gather; communicator: gslnls_amd.dist).  Returns B_total entries."""
        return self.irls(start, _gather_total=int(B_total), **kw)

    def irls(self, start, loss="bisquare", jac=True, fvv=False, algorithm="lm", control=None, lower=None, upper=None,
             lo=0, hi=None, _gather_total=None):
        hi = self.B if hi is None else hi
        ctrl = control if (control is not None and len(control) >= 23) else gsl_nls_control(**(control or {}))
        cfg = gsl_nls_loss(loss) if isinstance(loss, str) else gsl_nls_loss(**loss)
        ci, cd = pack_control(ctrl, algorithm)
        st = np.ascontiguousarray(start, dtype=np.float64)
        cc = np.asarray(list(cfg["cc"].values()) + [0.0, 0.0, 0.0], dtype=np.float64)
        cnt = hi - lo if _gather_total is None else _gather_total
        par = np.zeros((cnt, self.p))
        scal = np.zeros((cnt, 4))
        ints = np.zeros((cnt, 4), dtype=np.int32)
        ms = C.c_float(0)
        lu = None
        if lower is not None or upper is not None:
            lo_ = np.full(self.p, -np.inf) if lower is None else np.asarray(lower, dtype=np.float64)
            up_ = np.full(self.p, np.inf) if upper is None else np.asarray(upper, dtype=np.float64)
            lu = np.ascontiguousarray(np.stack([lo_, up_], axis=1).reshape(-1))
        if _gather_total is not None:
            rc = _lib.lib().gslnls_batch_irls_gather(self._h, _gather_total, int(bool(jac)), int(bool(fvv)),
                                                     st.ctypes.data_as(DP), None if lu is None else lu.ctypes.data_as(DP),
                                                     ci.ctypes.data_as(IP), cd.ctypes.data_as(DP), LOSSES.index(cfg["rho"]),
                                                     cc.ctypes.data_as(DP), par.ctypes.data_as(C.c_void_p),
                                                     scal.ctypes.data_as(C.c_void_p), ints.ctypes.data_as(C.c_void_p),
                                                     C.byref(ms))
        else:
            rc = _lib.lib().gslnls_batch_irls(self._h, lo, hi, int(bool(jac)), int(bool(fvv)), st.ctypes.data_as(DP),
                                              None if lu is None else lu.ctypes.data_as(DP), ci.ctypes.data_as(IP),
                                              cd.ctypes.data_as(DP), LOSSES.index(cfg["rho"]), cc.ctypes.data_as(DP),
                                              par.ctypes.data_as(C.c_void_p), scal.ctypes.data_as(C.c_void_p),
                                              ints.ctypes.data_as(C.c_void_p), C.byref(ms))
        _lib.check(rc)
        if rc != 0:
            raise RuntimeError("gslnls_batch_irls failed: %s" % _lib.strerror(rc))
        lm_p, rw_p = C.c_longlong(0), C.c_longlong(0)
        _lib.lib().gslnls_batch_last_passes(self._h, C.byref(lm_p), C.byref(rw_p))
        self.last_passes = dict(lm=int(lm_p.value), reweight=int(rw_p.value))  # this rank's data sets only
        return dict(par=par, sigma=scal[:, 0], ssr=scal[:, 1], irls_tol=scal[:, 2], chisq_init=scal[:, 3],
                    conv=ints[:, 0], irls_status=ints[:, 1], irls_niter=ints[:, 2], niter=ints[:, 3],
                    kernel_ms=float(ms.value))
