"""GPU parity of the robust-loss IRLS path (gsl_nls(loss = ...), src/nls_irls.c) through the C ABI."""
import numpy as np
import pytest

from conftest import c2_data

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import gslnls_amd
    from gslnls_amd import _lib
    assert _lib.lib().gslnls_device_count() >= 1
    return gslnls_amd


def test_readme_huber_on_gpu(amd, gslref, readme):
    """README.md:493-505: Huber loss, 8 IRLS iterations, achieved tolerance 0.0001023, last NLS solve 9 iterations"""
    e1 = readme["ex1"]
    h = e1["huber"]
    fit = amd.gsl_nls("y ~ A * exp(-lam * x) + b", data=dict(x=e1["x"], y=e1["y"]), start=dict(A=0, lam=0, b=0),
                      loss="huber")
    assert fit["conv"] == 0 and fit["irls"]["irls_status"] == 0
    assert fit["irls"]["irls_niter"] == h["irls_niter"] and fit["niter"] == h["nls_niter"]
    assert np.allclose(fit["par"], h["coef"], atol=5e-4)
    assert abs(fit["ssr"] - h["wssr"]) < 5e-5 and abs(fit["irls"]["irls_tol"] - h["irls_tol"]) < 5e-8
    # full agreement with the oracle on everything the R list carries
    x, y = np.array(e1["x"]), np.array(e1["y"])
    o = gslref.nls(25, 3, [0.0, 0.0, 0.0], rowdata=dict(model=gslref.MODEL_EXPDECAY, x=x, y=y), use_jac=False,
                   ctrl=gslref.control(solver="cholesky"), loss="huber")
    __import__("conftest").rel_err(fit["par"], o["par"])  # (recorded for the session summary)
    assert np.allclose(fit["par"], o["par"], rtol=1e-6)
    assert abs(fit["irls"]["irls_sigma"] - o["irls"]["irls_sigma"]) < 1e-6 * o["irls"]["irls_sigma"]
    assert np.allclose(fit["irls_weights"], o["irls_weights"], rtol=1e-5, atol=1e-9)
    assert np.allclose(fit["irls_psi"], o["irls_psi"], rtol=1e-5, atol=1e-7)
    assert np.allclose(fit["irls_dpsi"], o["irls_dpsi"], atol=1e-9)
    assert np.allclose(fit["resid"], o["resid"], atol=1e-6)


LOSSES = ["huber", "barron", "bisquare", "welsh", "optimal", "hampel", "ggw", "lqq"]


@pytest.mark.parametrize("loss", LOSSES)
@pytest.mark.parametrize("outlier", [True, False])
def test_unit_tests_5_1_robust_losses(amd, gslref, nist, loss, outlier):
    """unit_tests_gslnls.R:180-225: Misra1a with y[1] <- 25; pass = NLS converged, IRLS converged and
    relative error < 1e-2; here additionally equal to the oracle run of the same loss"""
    q = nist["Misra1a"]
    x, y = np.array(q["data"]["x"]), np.array(q["data"]["y"])
    if outlier:
        y = y.copy()
        y[0] = 25.0
    tgt = np.array(list(q["target"].values()))
    fit = amd.gsl_nls(q["formula"], data=dict(x=x, y=y), start=q["start"], loss=loss, control=dict(solver="cholesky"))
    assert fit["conv"] == 0 and fit["irls"]["irls_status"] == 0
    assert np.max(np.abs(1 - fit["par"] / tgt)) < 1e-2
    o = gslref.nls(14, 2, [500.0, 1e-4], rowdata=dict(model=gslref.MODEL_MISRA1A, x=x, y=y), use_jac=False,
                   ctrl=gslref.control(solver="cholesky"), loss=loss)
    assert fit["irls"]["irls_niter"] == o["irls"]["irls_niter"]
    __import__("conftest").rel_err(fit["par"], o["par"])  # (recorded for the session summary)
    assert np.allclose(fit["par"], o["par"], rtol=1e-5)
    assert np.allclose(fit["irls_weights"], o["irls_weights"], rtol=1e-4, atol=1e-8)


def test_irls_failure_and_weights(amd, gslref, nist):
    """5.1.17: barron alpha = -Inf with irls_xtol = 1e-20 must report non-convergence; 5.1.8: user weights"""
    q = nist["Misra1a"]
    fit = amd.gsl_nls(q["formula"], data=q["data"], start=q["start"], loss=dict(rho="barron", cc=[-np.inf, 1.345]),
                      control=dict(irls_xtol=1e-20))
    assert fit["conv"] != 0 and fit["irls"]["irls_status"] != 0
    tgt = np.array(list(q["target"].values()))
    fit = amd.gsl_nls(q["formula"], data=q["data"], start=q["start"], loss="welsh", weights=np.full(14, 10.0))
    assert fit["conv"] == 0 and fit["irls"]["irls_status"] == 0 and np.max(np.abs(1 - fit["par"] / tgt)) < 1e-2


def test_irls_large_n_median_on_device(amd, gslref):
    """radix-select median at a size where the reference would full-sort: n = 200001 (odd) and 200000 (even),
    2 % gross outliers; sigma and coefficients against the oracle"""
    for n in (200001, 200000):
        x, y = c2_data(n, seed=11)
        rng = np.random.default_rng(3)
        idx = rng.choice(n, n // 50, replace=False)
        y = y.copy()
        y[idx] += 20.0
        prob_fit = amd.gsl_nls("y ~ A * exp(-lam * x) + b", data=dict(x=x, y=y), start=dict(A=1, lam=1, b=0),
                               loss="bisquare", jac=True, control=dict(solver="cholesky"))
        o = gslref.nls(n, 3, [1.0, 1.0, 0.0], rowdata=dict(model=gslref.MODEL_EXPDECAY, x=x, y=y), use_jac=True,
                       ctrl=gslref.control(solver="cholesky"), loss="bisquare")
        assert prob_fit["conv"] == 0 and prob_fit["irls"]["irls_status"] == 0
        assert prob_fit["irls"]["irls_niter"] == o["irls"]["irls_niter"]
        # the fitted points of two correct implementations differ at the sqrt(eps) level, and so do their medians
        assert abs(prob_fit["irls"]["irls_sigma"] - o["irls"]["irls_sigma"]) <= 1e-6 * o["irls"]["irls_sigma"]
        # the device selection itself is exact: recompute the median at the device's own fitted point
        th = prob_fit["par"]
        r = np.abs(th[0] * np.exp(-th[1] * x) + th[2] - y)
        assert abs(prob_fit["irls"]["irls_sigma"] - 1.482602218505602 * np.median(r)) <= 1e-12
        assert np.allclose(prob_fit["par"], o["par"], rtol=1e-6)
        assert np.allclose(prob_fit["par"], [5.0, 1.5, 1.0], atol=0.02)   # the outliers are rejected


@pytest.mark.parametrize("loss,start,outlier", [("hampel", dict(b1=[200, 250], b2=1e-4), False),
                                                ("ggw", dict(b1=np.nan, b2=1e-4), False),
                                                ("huber", dict(b1=[200, 250], b2=[1e-4, 1e-3]), True),
                                                ("bisquare", dict(b1=[100, 400], b2=[1e-4, 1e-3]), True)])
def test_robust_multistart_second_pass(amd, gslref, nist, loss, start, outlier):
    """unit_tests_gslnls.R:212-216 (5.1.12, 5.1.14) + outlier variants: multi-start with a robust loss runs the
    Cook's-distance second pass (src/nls.c:401-509) before the IRLS solve; bookkeeping and result == oracle"""
    q = nist["Misra1a"]
    x, y = np.array(q["data"]["x"]), np.array(q["data"]["y"])
    if outlier:
        y = y.copy()
        y[0] = 25.0
    tgt = np.array(list(q["target"].values()))
    fit = amd.gsl_nls(q["formula"], data=dict(x=x, y=y), start=start, loss=loss, control=dict(solver="cholesky"))
    assert fit["conv"] == 0 and fit["irls"]["irls_status"] == 0
    assert np.max(np.abs(1 - fit["par"] / tgt)) < 1e-2
    from gslnls_amd.nls import _normalise_start
    names, vec, mat, hs = _normalise_start(start)
    o = gslref.nls(14, 2, mat, rowdata=dict(model=gslref.MODEL_MISRA1A, x=x, y=y), use_jac=False,
                   ctrl=gslref.control(solver="cholesky"), loss=loss, has_start=hs.astype(int))
    m, mo = fit["mstart"], o["mstart"]
    assert (m["nsp"], m["nwsp"], m["iters"], m["stop"]) == (mo["nsp"], mo["nwsp"], mo["iters"], mo["stop"])
    assert fit["irls"]["irls_niter"] == o["irls"]["irls_niter"]
    __import__("conftest").rel_err(fit["par"], o["par"])  # (recorded for the session summary)
    assert np.allclose(fit["par"], o["par"], rtol=1e-5)


def test_boundary_fills_the_whole_irls_slot(amd, gslref, nist):
    """the contract the R shim relies on (src/nls.c:756-791): a robust gslnls_nls() call fills irls_weights / psi / dpsi
    and the four scalars; a default-loss call leaves the arrays untouched"""
    import ctypes as C
    from gslnls_amd import _lib
    from gslnls_amd.control import gsl_nls_control, pack_control
    q = nist["Misra1a"]
    x, y = np.array(q["data"]["x"]), np.array(q["data"]["y"]).copy()
    y[0] = 25.0
    n, p = len(y), 2
    X = np.asfortranarray(x.reshape(n, 1))
    m = _lib.Model(2, p, 1, X.ctypes.data_as(C.c_void_p), 0)
    ci, cd = pack_control(gsl_nls_control(solver="cholesky"), "lm")
    st = np.array([500.0, 1e-4])
    cc = np.array([4.685061])
    hs = np.ones(4, dtype=np.int32)
    for rho in (3, 0):
        par, w, psi, dpsi = np.zeros(p), np.full(n, -7.0), np.full(n, -7.0), np.full(n, -7.0)
        res = _lib.Result()
        res.par = par.ctypes.data_as(_lib.DP)
        res.irls_weights, res.irls_psi, res.irls_dpsi = (a.ctypes.data_as(_lib.DP) for a in (w, psi, dpsi))
        rc = _lib.lib().gslnls_nls(C.byref(m), y.ctypes.data_as(C.c_void_p), n, 0, 0, st.ctypes.data_as(_lib.DP), 0, None,
                                   0, None, ci.ctypes.data_as(_lib.IP), cd.ctypes.data_as(_lib.DP),
                                   hs.ctypes.data_as(_lib.IP), rho, cc.ctypes.data_as(_lib.DP), C.byref(res))
        assert rc == 0 and res.conv == 0
        if rho:
            o = gslref.nls(n, p, st, rowdata=dict(model=gslref.MODEL_MISRA1A, x=x, y=y), use_jac=False,
                           ctrl=gslref.control(solver="cholesky"), loss="bisquare")
            assert np.all(w != -7.0) and np.all(psi != -7.0) and np.all(dpsi != -7.0)
            assert np.allclose(w, o["irls_weights"], rtol=1e-5, atol=1e-9)
            assert np.allclose(psi, o["irls_psi"], rtol=1e-5, atol=1e-7)
            assert res.irls_status == 0 and res.irls_niter == o["irls"]["irls_niter"] and res.irls_sigma > 0
            assert abs(res.irls_tol - o["irls"]["irls_tol"]) < 1e-3 * o["irls"]["irls_tol"]
        else:
            assert np.all(w == -7.0) and np.all(psi == -7.0)
        assert np.isfinite(res.jtj_cond) and res.jtj_cond >= 1.0
        assert _lib.lib().gslnls_solver_served(ci.ctypes.data_as(_lib.IP), C.byref(res)) == 1


def test_solver_routing_rule(amd, nist):
    """include/gslnls_core.h: cholesky is always served; qr / svd requests are served on the normal equations while the
    scaled condition number stays below GSLNLS_COND_LIMIT, and flagged for the GSL path beyond it"""
    import warnings
    q = nist["Misra1a"]
    d = dict(x=np.array(q["data"]["x"]), y=np.array(q["data"]["y"]))
    fit = amd.gsl_nls(q["formula"], data=d, start=q["start"])  # default control: solver = "qr"
    assert fit["conv"] == 0 and fit["solver_served"] and 1.0 <= fit["jtj_cond"] < 1e10
    # a nearly collinear pair of columns: f = (a + b) x + 1e-7 b x^2
    x = np.linspace(1.0, 2.0, 50)
    y = 3.0 * x + 1e-7 * 2.0 * x ** 2
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        bad = amd.gsl_nls("y ~ (a + b) * x + 1e-7 * b * x^2", data=dict(x=x, y=y), start=dict(a=1.0, b=1.0), jac=True)
    assert bad["jtj_cond"] > 1e10 and not bad["solver_served"]
    assert any("falls through to GSL" in str(v.message) for v in w)
    ok = amd.gsl_nls("y ~ (a + b) * x + 1e-7 * b * x^2", data=dict(x=x, y=y), start=dict(a=1.0, b=1.0), jac=True,
                     control=dict(solver="cholesky"))
    assert ok["solver_served"]
