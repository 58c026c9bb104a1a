"""gsl_nls_control() / gsl_nls_loss(): the reference's flag system and its packing into the
positional control vectors that cross the .Call boundary (R/nls.R:693-713, :1186-1229;
R/nls_rho.R:106-144)."""
import math
import warnings

import numpy as np

EPS = float(np.finfo(np.float64).eps)

ALGORITHMS = ("lm", "lmaccel", "dogleg", "ddogleg", "subspace2D")
LARGE_ALGORITHMS = ALGORITHMS + ("cgst",)
SCALES = ("more", "levenberg", "marquardt")
SOLVERS = ("qr", "cholesky", "svd")
FDTYPES = ("forward", "center")
LOSSES = ("default", "huber", "barron", "bisquare", "welsh", "optimal", "hampel", "ggw", "lqq")

_LOSS_DEFAULT_CC = {
    "default": (),
    "huber": (("k", 1.345),),
    "barron": (("alpha", 1.0), ("k", 1.345)),
    "bisquare": (("k", 4.685061),),
    "welsh": (("k", 2.11),),
    "optimal": (("k", 1.060158),),
    "hampel": (("k", 0.9016085),),
    "ggw": (("a", 1.387), ("b", 1.5), ("c", 1.063)),
    "lqq": (("b", 1.473), ("c", 0.982), ("s", 1.5)),
}


def _match_arg(v, choices, what):
    if v not in choices:
        raise ValueError("'%s' should be one of %s" % (what, ", ".join(choices)))
    return v


def gsl_nls_control(maxiter=100, scale="more", solver="qr", fdtype="forward", factor_up=2.0, factor_down=3.0,
                    avmax=0.75, h_df=math.sqrt(EPS), h_fvv=0.02, xtol=math.sqrt(EPS), ftol=math.sqrt(EPS),
                    gtol=math.sqrt(EPS), mstart_n=30, mstart_p=5, mstart_q=None, mstart_r=4.0, mstart_s=2,
                    mstart_tol=0.25, mstart_maxiter=10, mstart_maxstart=250, mstart_minsp=1, irls_maxiter=50,
                    irls_xtol=EPS ** 0.25):
    """Same names, defaults and validation as R/nls.R:1186-1229."""
    if mstart_q is None:
        mstart_q = mstart_n // 10
    _match_arg(scale, SCALES, "scale")
    _match_arg(solver, SOLVERS, "solver")
    _match_arg(fdtype, FDTYPES, "fdtype")
    c = dict(maxiter=maxiter, scale=scale, solver=solver, fdtype=fdtype, factor_up=factor_up,
             factor_down=factor_down, avmax=avmax, h_df=h_df, h_fvv=h_fvv, xtol=xtol, ftol=ftol, gtol=gtol,
             mstart_n=mstart_n, mstart_p=mstart_p, mstart_q=mstart_q, mstart_r=mstart_r, mstart_s=mstart_s,
             mstart_tol=mstart_tol, mstart_maxiter=mstart_maxiter, mstart_maxstart=mstart_maxstart,
             mstart_minsp=mstart_minsp, irls_maxiter=irls_maxiter, irls_xtol=irls_xtol)
    for k in ("maxiter", "mstart_n", "mstart_p", "mstart_q", "mstart_s", "mstart_maxiter", "mstart_maxstart",
              "mstart_minsp", "irls_maxiter"):
        if not (isinstance(c[k], (int, np.integer)) or float(c[k]).is_integer()) or c[k] < 1:
            raise ValueError("%s must be a positive integer" % k)
        c[k] = int(c[k])
    for k in ("factor_up", "factor_down", "avmax", "h_df", "h_fvv", "xtol", "ftol", "gtol", "mstart_tol",
              "irls_xtol"):
        if not c[k] > 0:
            raise ValueError("%s must be positive" % k)
    if not c["mstart_r"] > 1:
        raise ValueError("mstart_r must be larger than 1")
    return c


def gsl_nls_loss(rho="default", cc=None):
    """R/nls_rho.R:106-144: list(rho=, cc=) with default tuning constants."""
    _match_arg(rho, LOSSES, "rho")
    default = _LOSS_DEFAULT_CC[rho]
    names = [k for k, _ in default]
    if cc is None or rho == "default":
        vals = [v for _, v in default]
    else:
        if isinstance(cc, dict):
            if not all(k in cc for k in names):
                raise ValueError("'cc' must be unnamed or include names %s" % ", ".join(names))
            vals = [float(cc[k]) for k in names]
        else:
            vals = [float(v) for v in np.atleast_1d(cc)]
            if len(vals) != len(names):
                raise ValueError("'cc' must be of length %d for function '%s'" % (len(names), rho))
    if rho == "barron" and vals[0] > 2:
        warnings.warn("Robustness parameter (alpha) in Barron loss function cannot be larger than 2")
        vals[0] = 2.0
    return dict(rho=rho, cc=dict(zip(names, vals)))


def pack_control(ctrl, algorithm="lm", trace=False, startisnum=True, any_missing_start=False):
    """.ctrl_int (15) / .ctrl_dbl (11) as R/nls.R:693-713 builds them."""
    _match_arg(algorithm, ALGORITHMS, "algorithm")
    ci = np.array([ctrl["maxiter"], int(bool(trace)), ALGORITHMS.index(algorithm), SCALES.index(ctrl["scale"]),
                   SOLVERS.index(ctrl["solver"]), FDTYPES.index(ctrl["fdtype"]), ctrl["mstart_n"],
                   ctrl["mstart_p"], ctrl["mstart_q"], ctrl["mstart_s"], ctrl["mstart_maxiter"],
                   ctrl["mstart_maxstart"], ctrl["mstart_minsp"], int(bool(startisnum)), ctrl["irls_maxiter"]],
                  dtype=np.int32)
    r = ctrl["mstart_r"] * (10.0 if any_missing_start else 1.0)
    cd = np.array([ctrl["factor_up"], ctrl["factor_down"], ctrl["avmax"], ctrl["h_df"], ctrl["h_fvv"],
                   ctrl["xtol"], ctrl["ftol"], ctrl["gtol"], r, ctrl["mstart_tol"], ctrl["irls_xtol"]],
                  dtype=np.float64)
    return ci, cd
