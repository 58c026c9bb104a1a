"""GPU parity tests of option combinations that the reference's own tests never exercise together: control values off
their defaults, finite differences with custom steps under lmaccel, robust IRLS with weights and bounds, multi-start with
weights and bounds, multi-start with a missing range (dynamic ranges), the scaling rules and trust-region factors on the
large path.  Bar: the oracle's iteration counts and bookkeeping, coefficients to round-off."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FORMULA = "y ~ A*exp(-lam*x) + b"


@pytest.fixture(scope="module")
def amd():
    import gslnls_amd
    from gslnls_amd import _lib
    assert _lib.lib().gslnls_device_count() >= 1, "no MI355X visible: the HIP path cannot be tested"
    return gslnls_amd


@pytest.fixture(scope="module")
def decay():
    rng = np.random.default_rng(1)
    x = np.linspace(0, 3, 200)
    y = 5 * np.exp(-1.5 * x) + 1 + 0.05 * rng.standard_normal(200)
    w = rng.uniform(0.5, 2.0, 200)
    return x, y, w


def _same(fit, o, tol=1e-9):
    assert fit["conv"] == o["conv"] == 0
    assert fit["niter"] == o["niter"], (fit["niter"], o["niter"])
    assert np.max(np.abs(np.asarray(fit["par"]) - o["par"])) <= tol * np.max(np.abs(o["par"]))


@pytest.mark.parametrize("fd", ["forward", "center"])
def test_lmaccel_with_finite_differences_and_custom_steps(amd, gslref, decay, fd):
    """h_df = 1e-6 (src/fdjac.c:36-38) and h_fvv = 0.05 (src/fdfvv.c:35-77) under algorithm = "lmaccel" """
    x, y, w = decay
    ctrl = dict(solver="cholesky", fdtype=fd, h_df=1e-6, h_fvv=0.05)
    fit = amd.gsl_nls(FORMULA, data=dict(x=x, y=y), start=dict(A=1.0, lam=1.0, b=0.0), algorithm="lmaccel", jac=False,
                      fvv=False, control=ctrl)
    o = gslref.nls(200, 3, [1.0, 1.0, 0.0], rowdata=dict(model=gslref.MODEL_EXPDECAY, x=x, y=y), use_jac=False,
                   algorithm="lmaccel", ctrl=gslref.control(**ctrl))
    _same(fit, o)


@pytest.mark.parametrize("alg", ["lm", "lmaccel"])
def test_trust_region_factors_and_avmax_off_their_defaults(amd, gslref, decay, alg):
    x, y, w = decay
    ctrl = dict(solver="cholesky", factor_up=4.0, factor_down=1.5, avmax=0.3)
    fit = amd.gsl_nls(FORMULA, data=dict(x=x, y=y), start=dict(A=1.0, lam=1.0, b=0.0), algorithm=alg, jac=True, control=ctrl)
    o = gslref.nls(200, 3, [1.0, 1.0, 0.0], rowdata=dict(model=gslref.MODEL_EXPDECAY, x=x, y=y), use_jac=True,
                   algorithm=alg, ctrl=gslref.control(**ctrl))
    _same(fit, o)


def test_robust_irls_with_weights_and_bounds(amd, gslref, decay):
    """loss = "bisquare" + weights + an active upper bound on A (trust_trial_step_lu, src/trust.c:9-32)"""
    x, y, w = decay
    fit = amd.gsl_nls(FORMULA, data=dict(x=x, y=y), start=dict(A=1.0, lam=1.0, b=0.6), jac=True, loss="bisquare", weights=w,
                      lower=dict(A=0.0, lam=0.0, b=0.5), upper=dict(A=4.5, lam=3.0, b=2.0), control=dict(solver="cholesky"))
    o = gslref.nls(200, 3, [1.0, 1.0, 0.6], rowdata=dict(model=gslref.MODEL_EXPDECAY, x=x, y=y), use_jac=True,
                   loss="bisquare", weights=w, lower=[0.0, 0.0, 0.5], upper=[4.5, 3.0, 2.0],
                   ctrl=gslref.control(solver="cholesky"))
    _same(fit, o)
    assert fit["irls"]["irls_niter"] == o["irls"]["irls_niter"]


def test_multistart_with_weights_bounds_and_a_missing_range(amd, gslref, decay):
    """start ranges inside the bounds with weights; then one parameter without a range (NA: the reference fills
    (-0.1, 0.75) and lets the range move, R/nls.R:399-437, src/nls_mstart.c:131-138) -- same bookkeeping as the oracle"""
    x, y, w = decay
    ctrl = dict(solver="cholesky", mstart_n=20, mstart_q=3)
    rd = dict(model=gslref.MODEL_EXPDECAY, x=x, y=y)
    fit = amd.gsl_nls(FORMULA, data=dict(x=x, y=y), start=dict(A=[0.5, 5.5], lam=[0.1, 4.0], b=[0.0, 2.0]), jac=True,
                      weights=w, lower=dict(A=0.0, lam=0.0, b=0.0), upper=dict(A=6.0, lam=5.0, b=3.0), control=ctrl)
    o = gslref.nls(200, 3, np.array([[0.5, 0.1, 0.0], [5.5, 4.0, 2.0]]), rowdata=rd, use_jac=True, weights=w,
                   lower=[0.0, 0.0, 0.0], upper=[6.0, 5.0, 3.0], ctrl=gslref.control(**ctrl))
    _same(fit, o)
    assert (fit["mstart"]["nsp"], fit["mstart"]["iters"]) == (o["mstart"]["nsp"], o["mstart"]["iters"])
    fit = amd.gsl_nls(FORMULA, data=dict(x=x, y=y), start=dict(A=[np.nan, np.nan], lam=[0.1, 4.0], b=[0.0, 2.0]), jac=True,
                      control=ctrl)
    o = gslref.nls(200, 3, np.array([[-0.1, 0.1, 0.0], [0.75, 4.0, 2.0]]), rowdata=rd, use_jac=True,
                   ctrl=gslref.control(**ctrl), has_start=np.array([[0, 1, 1], [0, 1, 1]]))
    _same(fit, o)
    assert (fit["mstart"]["nsp"], fit["mstart"]["iters"]) == (o["mstart"]["nsp"], o["mstart"]["iters"])


@pytest.mark.parametrize("alg", ["lm", "cgst"])
@pytest.mark.parametrize("scale", ["more", "levenberg", "marquardt"])
def test_large_path_scaling_rules_and_factors(amd, gslref, scale, alg):
    """gsl_nls_large with every scaling rule (GSL multilarge scaling.c) and factor_up / factor_down off their defaults"""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_large import glm_data
    A, y, th = glm_data(3000, 16)
    kw = dict(scale=scale, factor_up=3.0, factor_down=2.0)
    fit = amd.gsl_nls_large("glmexp", A=A, y=y, start=np.zeros(16), algorithm=alg, control=kw)
    o = gslref.nls_large(3000, 16, np.zeros(16), rowdata=dict(model=gslref.MODEL_GLMEXP, x=A, y=y), algorithm=alg,
                         ctrl=gslref.control(**kw))
    _same(fit, o)
