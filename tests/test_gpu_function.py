"""GPU parity tests of gsl_nls() on `function` models and on more than 64 parameters: the Jacobian is a matrix in HBM
(csrc/bd_host.hpp, csrc/bd_kernels.hpp), the closures run on the host exactly where the reference runs them
(gsl_f / gsl_df / gsl_fvv, src/nls.c:815-978), everything n x p and p x p on the device.  Through the C ABI
(gslnls_nls_fn) with Python closures, against the oracle's multifit driver with the same closures.

Reference tests mirrored: unit_tests_gslnls.R:74-101 (2.2.x / 2.3.x: "Linear, full rank" as a function, lmaccel with
fvv, weights, central differences, the gradient attribute), README.md:1286-1295 (Madsen, 42 iterations) and
README.md:1067-1079 (example 4: gsl_nls() LM at p = 500, ssr 0.004778845)."""
import time

import numpy as np
import pytest
from conftest import record_parity

pytestmark = pytest.mark.gpu

TOL = float(np.finfo(float).eps ** 0.25)
REPORT = []


@pytest.fixture(scope="module")
def amd():
    import gslnls_amd
    from gslnls_amd import _lib
    assert _lib.lib().gslnls_device_count() >= 1, "no MI355X visible: the HIP path cannot be tested"
    yield gslnls_amd
    if REPORT:
        print("\n[function models] relative error of the coefficients vs the oracle (measured): " + "; ".join(REPORT))


def _rel(a, b):
    from conftest import rel_err
    return rel_err(a, b)  # (recorded: the session summary prints what every comparison measured)


def linear_full_rank(n, p):
    """MGH 'Linear, full rank' (src/test_nls.f90 problem 1; R/nls_test.R): f_i = x_i - (2/n) sum x - 1 (i <= p),
    - (2/n) sum x - 1 beyond"""
    def fn(x):
        s = 2.0 * np.sum(x) / n
        out = np.full(n, -s - 1.0)
        out[:p] += x
        return out

    def jac(x):
        J = np.full((n, p), -2.0 / n)
        J[:p, :p] += np.eye(p)
        return J
    return fn, jac


def test_linear_full_rank_as_a_function_2_2_and_2_3(amd, gslref, mgh):
    q = next(m for m in mgh.values() if m["name"] == "Linear, full rank") if isinstance(mgh, dict) else None
    fn, jac = linear_full_rank(5, 5)
    if q is not None and q["n"] == 10:
        f10, j10 = linear_full_rank(10, 5)
        assert np.allclose(f10(np.array(q["start"])), q["f_start"], atol=1e-14)  # the closure is the catalogue's function
        assert np.allclose(j10(np.array(q["start"])).reshape(-1), q["J_start_rowmajor"], atol=1e-14)
    y = np.zeros(5)
    start = np.zeros(5)
    target = -np.ones(5)
    fvv = lambda th, v: np.zeros(5)  # noqa: E731 -- a linear model
    # 2.2.1: lmaccel, jac, fvv = TRUE, trace
    fit = amd.gsl_nls(fn, y=y, start=start, algorithm="lmaccel", jac=jac, fvv=fvv, trace=True, control=dict(solver="cholesky"))
    ref = gslref.nls(5, 5, start, fn=fn, jac=jac, fvv=fvv, algorithm="lmaccel", trace=True, ctrl=gslref.control(solver="cholesky"))
    assert fit["conv"] == 0 and fit["code_path"] == 4 and np.all(np.abs(fit["par"] - target) <= TOL)
    assert fit["niter"] == ref["niter"] and fit["neval"] == ref["neval"]
    assert np.allclose(fit["partrace"], ref["partrace"], rtol=1e-10, atol=1e-12)
    REPORT.append("2.2.1 %.1e" % _rel(fit["par"], ref["par"]))
    # 2.2.4: lmaccel, weights, central differences (fvv by differences too)
    w = np.full(5, 100.0)
    fit = amd.gsl_nls(fn, y=y, start=start, algorithm="lmaccel", weights=w, control=dict(fdtype="center", solver="cholesky"))
    ref = gslref.nls(5, 5, start, fn=fn, algorithm="lmaccel", weights=w, ctrl=gslref.control(fdtype="center", solver="cholesky"))
    assert fit["conv"] == 0 and np.all(np.abs(fit["par"] - target) <= TOL)
    assert fit["niter"] == ref["niter"] and fit["neval"] == ref["neval"]
    assert fit["ssr"] < 1e-15 and ref["ssr"] < 1e-15  # (a consistent linear system: both end at round-off level)
    # 2.3.1: the gradient travels with the value (the "gradient" attribute of the reference's closure), lmaccel
    fit = amd.gsl_nls(lambda th: (fn(th), jac(th)), y=y, start=start, algorithm="lmaccel", control=dict(solver="cholesky"))
    ref = gslref.nls(5, 5, start, fn=fn, jac=jac, algorithm="lmaccel", ctrl=gslref.control(solver="cholesky"))
    assert fit["conv"] == 0 and np.all(np.abs(fit["par"] - target) <= TOL)
    assert fit["niter"] == ref["niter"] and fit["neval"] == ref["neval"]
    # bounds: 2.3.2's box [target / 2, start] with LM (subspace2D is not lowered): the fit ends on the lower bounds
    fit = amd.gsl_nls(fn, y=y, start=start, jac=jac, lower=target / 2, upper=start, control=dict(solver="cholesky"))
    ref = gslref.nls(5, 5, start, fn=fn, jac=jac, lower=target / 2, upper=start, ctrl=gslref.control(solver="cholesky"))
    assert fit["conv"] == ref["conv"] and fit["niter"] == ref["niter"]
    __import__("conftest").rel_err(fit["par"], ref["par"])  # (recorded for the session summary)
    assert np.allclose(fit["par"], ref["par"], rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize("jacmode", ["analytic", "forward", "center"])
def test_madsen_as_a_function_matches_the_oracle(amd, gslref, pins, jacmode):
    """README.md:1286-1295: Madsen from (3, 1): 42 LM iterations with the analytic Jacobian; the difference Jacobians are
    built from p + 1 (2 p) calls of the closure and charged the same way (src/fdjac.c, App. A.8)"""
    fn = lambda t: np.array([t[0] ** 2 + t[1] ** 2 + t[0] * t[1], np.sin(t[0]), np.cos(t[1])])  # noqa: E731
    jac = lambda t: np.array([[2 * t[0] + t[1], 2 * t[1] + t[0]], [np.cos(t[0]), 0.0], [0.0, -np.sin(t[1])]])  # noqa
    kw = dict(solver="cholesky") if jacmode == "analytic" else dict(solver="cholesky", fdtype=jacmode)
    fit = amd.gsl_nls(fn, y=np.zeros(3), start=[3.0, 1.0], jac=jac if jacmode == "analytic" else None, control=kw, trace=True)
    ref = gslref.nls(3, 2, [3.0, 1.0], fn=fn, jac=jac if jacmode == "analytic" else None, ctrl=gslref.control(**kw), trace=True)
    assert fit["conv"] == 0 and ref["conv"] == 0
    if jacmode == "analytic":
        # the iterations are the README's 42; the last ones sit at round-off level (ssr changes in its last bit: a three-row
        # sum of squares added as a butterfly here, left to right in the oracle), where a trial is accepted or rejected by that bit
        assert fit["niter"] == ref["niter"] == pins["madsen_lm"]["niter"]
        assert fit["neval"]["J"] == ref["neval"]["J"] and abs(fit["neval"]["f"] - ref["neval"]["f"]) <= 16
    else:
        # differences amplify that last bit by 1 / h: the round-off tail is an iteration longer or shorter
        assert abs(fit["niter"] - ref["niter"]) <= 2
    # default stopping rule |dx| < xtol (1 + |x|), xtol = 1.5e-8, on a problem whose residual does not vanish: both stop within
    # ~xtol of the optimum (tests/test_gpu_dense.py::test_c2_matches_oracle has the argument)
    assert _rel(fit["par"], ref["par"]) < 1e-6
    assert abs(fit["ssr"] - ref["ssr"]) <= 1e-10 * ref["ssr"]
    assert np.allclose(fit["covar"], ref["covar"], rtol=1e-5)
    assert np.allclose(fit["resid"], ref["resid"], rtol=0, atol=1e-6)
    assert np.allclose(fit["grad"], ref["grad"], rtol=1e-5, atol=1e-6)
    REPORT.append("Madsen/%s %.1e" % (jacmode, _rel(fit["par"], ref["par"])))


def gaussians(ng, n, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    x = np.linspace(0.0, 10.0 * ng, n)
    amp, mid, wid = rng.uniform(2.0, 6.0, ng), 10.0 * np.arange(ng) + rng.uniform(3.0, 7.0, ng), rng.uniform(1.2, 2.4, ng)
    truth = np.append(np.stack([amp, mid, wid], axis=1).reshape(-1), [0.5])
    p = 3 * ng + 1

    def model(th):
        a, m, s = th[0:3 * ng:3], th[1:3 * ng:3], th[2:3 * ng:3]
        return np.sum(a[None, :] * np.exp(-((x[:, None] - m[None, :]) / s[None, :]) ** 2), axis=1) + th[-1]

    def jac(th):
        a, m, s = th[0:3 * ng:3], th[1:3 * ng:3], th[2:3 * ng:3]
        z = (x[:, None] - m[None, :]) / s[None, :]
        e = np.exp(-z * z)
        J = np.empty((n, p))
        J[:, 0:3 * ng:3] = e
        J[:, 1:3 * ng:3] = a[None, :] * e * 2.0 * z / s[None, :]
        J[:, 2:3 * ng:3] = a[None, :] * e * 2.0 * z * z / s[None, :]
        J[:, -1] = 1.0
        return J
    y = model(truth) + 0.05 * rng.standard_normal(n)
    # amplitudes and widths 2 % off, centres 0.1 off (a relative error of a centre near 300 would be several widths)
    start = truth * (1.0 + 0.02 * np.where(np.arange(p) % 2 == 0, 1.0, -1.0))
    start[1:3 * ng:3] = truth[1:3 * ng:3] + 0.1 * np.where(np.arange(ng) % 2 == 0, 1.0, -1.0)
    return x, y, model, jac, start, truth


@pytest.mark.parametrize("ng,n", [(33, 3000), (66, 5000)])
def test_sum_of_gaussians_p100_and_p199_matches_the_oracle(amd, gslref, ng, n):
    """gsl_nls() beyond 64 parameters (the reference takes any p, src/nls.c:266): p = 100 and p = 199, J^T J in 64 x 64 MFMA
    blocks, the damped solve on the host routine below p = 400 -- same iteration count as the oracle's multifit driver"""
    x, y, model, jac, start, truth = gaussians(ng, n, 4200 + ng)
    p = len(start)
    ctrl = dict(solver="cholesky")
    t0 = time.perf_counter()
    fit = amd.gsl_nls(model, y=y, start=start, jac=jac, control=ctrl)
    wall = time.perf_counter() - t0
    ref = gslref.nls(n, p, start, fn=lambda th: model(th) - y, jac=jac, ctrl=gslref.control(**ctrl))
    assert fit["conv"] == 0 and ref["conv"] == 0 and fit["code_path"] == 4
    # (the last iteration sits at round-off level: whether its trials are accepted is decided by the last bit of ssr)
    assert fit["niter"] == ref["niter"] and abs(fit["neval"]["J"] - ref["neval"]["J"]) <= 1
    assert abs(fit["neval"]["f"] - ref["neval"]["f"]) <= 16
    assert _rel(fit["par"], ref["par"]) < 1e-8
    assert abs(fit["ssr"] - ref["ssr"]) <= 1e-10 * ref["ssr"]
    assert np.allclose(np.diag(fit["covar"]), np.diag(ref["covar"]), rtol=1e-6)
    REPORT.append("p=%d %.1e (%.0f ms, %d iterations)" % (p, _rel(fit["par"], ref["par"]), 1e3 * wall, fit["niter"]))


def penalty1(p):
    a = np.sqrt(1e-5)

    def fn(th):
        return np.concatenate([a * (th - 1.0), [np.sum(th ** 2) - 0.25]])

    def jac(th):
        return np.vstack([a * np.eye(p), 2.0 * th[None, :]])
    return fn, jac


def test_readme_example_4_through_gsl_nls(amd, gslref, monkeypatch):
    """README.md:1067-1079: gsl_nls(fn = f, y = rep(0, p + 1), start = 1:p, control = list(maxiter = 500)), p = 500 -- the
    reference's own dense benchmark call (36.66 s there, hardware unstated).  ssr 0.004778845 either way; the iteration
    count is compared with the oracle's multifit driver twice: with the p x p factorisation by the host routine (the
    oracle's order of operations: within 5 %) and by the device routine that serves p >= 400 (same algorithm, sums in
    another order: the valley is flat, ~200 iterations amplify the last bits of every step, so the count moves by ~10 %)"""
    p = 500
    fn, jac = penalty1(p)
    y = np.zeros(p + 1)
    start = np.arange(1.0, p + 1.0)
    ref = gslref.nls(p + 1, p, start, fn=fn, jac=jac, ctrl=gslref.control(maxiter=500, solver="cholesky"))
    assert ref["conv"] == 0 and abs(ref["ssr"] - 0.004778845) < 5e-10
    for where, tol in (("host", 0.05), ("device", 0.12)):
        monkeypatch.setenv("GSLNLS_LARGE_CHOL_DEVICE_MIN", "0" if where == "host" else "400")
        for rep in range(2):
            t0 = time.perf_counter()
            fit = amd.gsl_nls(lambda th: (fn(th), jac(th)), y=y, start=start, control=dict(maxiter=500, solver="cholesky"))
            wall = time.perf_counter() - t0
        assert fit["conv"] == 0 and fit["code_path"] == 4
        assert abs(fit["ssr"] - 0.004778845) < 5e-10
        assert abs(fit["ssr"] - ref["ssr"]) <= 1e-9 * ref["ssr"]
        assert abs(fit["niter"] - ref["niter"]) <= tol * ref["niter"], (where, fit["niter"], ref["niter"])
        print("\nREADME example 4 through gsl_nls(), factorisation on the %s: p = 500, %d iterations (oracle %d), ssr %.9f, %.0f ms "
              "wall (the README quotes 36.66 s on unstated hardware)" % (where, fit["niter"], ref["niter"], fit["ssr"], 1e3 * wall))


def test_function_model_failures_are_reported_not_hidden(amd):
    """a closure that returns the wrong length is EBADFUNC (src/nls.c:843-844), non-finite values become +Inf and the trial
    is rejected (:846-849), a Python exception comes back as that exception"""
    with pytest.raises(ZeroDivisionError):
        amd.gsl_nls(lambda th: np.array([1.0 / 0, 1.0]), y=np.zeros(2), start=[1.0])
    fit = amd.gsl_nls(lambda th: np.zeros(3), y=np.zeros(2), start=[1.0, 2.0])
    assert fit["conv"] == 9
    # the model is not finite to the right of 2: the steps that land there are rejected, the fit ends on the left
    f = lambda th: np.where(th[0] < 2.0, (th[0] - 1.0) * np.ones(4), np.nan)  # noqa: E731
    fit = amd.gsl_nls(f, y=np.zeros(4), start=[1.9])
    assert fit["conv"] == 0 and abs(fit["par"][0] - 1.0) < 1e-6


@pytest.mark.parametrize("jac", [True, False])
def test_formula_with_100_parameters_matches_the_oracle(amd, gslref, jac):
    """a FORMULA with p = 100 (33 Gaussians + constant): value and symbolic gradient compiled in process, rows written
    as an n x p matrix in HBM, the same fit as the closure form above and as the oracle (analytic and forward differences)"""
    ng, n = 33, 3000
    x, y, model, jacf, start, truth = gaussians(ng, n, 4200 + ng)
    names, terms = [], []
    for k in range(1, ng + 1):
        names += ["a%d" % k, "m%d" % k, "s%d" % k]
        terms.append("a%d*exp(-((x-m%d)/s%d)^2)" % (k, k, k))
    names += ["c0"]
    fit = amd.gsl_nls("y ~ " + " + ".join(terms) + " + c0", data=dict(x=x, y=y), start=dict(zip(names, start)), jac=jac,
                      control=dict(solver="cholesky"), lowering="jit")
    ref = gslref.nls(n, len(start), start, fn=lambda th: model(th) - y, jac=jacf if jac else None,
                     ctrl=gslref.control(solver="cholesky"))
    assert fit["conv"] == 0 and ref["conv"] == 0 and fit["code_path"] == 4
    assert fit["niter"] == ref["niter"]
    if jac:
        assert fit["neval"] == ref["neval"]
    assert _rel(fit["par"], ref["par"]) < 1e-7
    assert abs(fit["ssr"] - ref["ssr"]) <= 1e-9 * ref["ssr"]
    assert np.allclose(fit["resid"], ref["resid"], rtol=0, atol=1e-7)
    REPORT.append("formula p=100 jac=%s %.1e" % (jac, _rel(fit["par"], ref["par"])))


@pytest.mark.parametrize("ng,n", [(22, 2000), (33, 3000), (66, 5000)])
def test_covariance_and_condition_from_the_device_equal_the_host_routines(amd, gslref, monkeypatch, ng, n):
    """Round 5: at a fit's end (J^T J)^-1 and the solver-routing diagnostic kappa(S J^T J S) come from the device (L^-1 column by
    column behind the damped solve's factorisation, X^T X on the matrix cores, two power iterations); the host routines they
    replace stay as the fallback for matrices the natural-order factorisation refuses.  Same fit through both: the
    covariance to 1e-10 of its scale, the condition number to 1e-8, and the oracle's covariance as before."""
    x, y, model, jac, start, truth = gaussians(ng, n, 4300 + ng)
    p = len(start)
    ctrl = dict(solver="cholesky")
    dev = amd.gsl_nls(model, y=y, start=start, jac=jac, control=ctrl)
    monkeypatch.setenv("GSLNLS_BD_HOST_EPILOGUE", "1")
    host = amd.gsl_nls(model, y=y, start=start, jac=jac, control=ctrl)
    monkeypatch.delenv("GSLNLS_BD_HOST_EPILOGUE")
    assert dev["conv"] == 0 and host["conv"] == 0 and dev["code_path"] == 4
    assert np.array_equal(dev["par"], host["par"]) and dev["niter"] == host["niter"]
    cd, ch = np.asarray(dev["covar"]), np.asarray(host["covar"])
    assert np.array_equal(cd, cd.T)
    scale = np.sqrt(np.outer(np.diag(ch), np.diag(ch)))
    err = float(np.max(np.abs(cd - ch) / scale))
    record_parity("matrix path p=%d covariance device vs host" % p, err, 1e-10)
    assert err < 1e-10, err
    assert abs(dev["jtj_cond"] - host["jtj_cond"]) <= 1e-8 * host["jtj_cond"], (dev["jtj_cond"], host["jtj_cond"])
    ref = gslref.nls(n, p, start, fn=lambda th: model(th) - y, jac=jac, ctrl=gslref.control(**ctrl))
    assert np.allclose(np.diag(cd), np.diag(ref["covar"]), rtol=1e-6)


def test_function_model_with_1500_parameters_end_of_fit_on_the_device(amd, monkeypatch):
    """p = 1500 (a model linear in its parameters, so that the answer is known): every panel of the blocked Cholesky, the
    one-launch-per-step kernel with trailing tiles, L^-1 by forward substitution over 24 blocks, X^T X and the power iterations
    at a size no other test reaches -- X^T X cov = I to 1e-12, covariance and condition number equal to the host routines'."""
    p, n = 1500, 1700
    rng = np.random.default_rng(p)
    X = rng.standard_normal((n, p)) + 0.1
    truth = rng.standard_normal(p)
    y = X @ truth
    Xf = np.asfortranarray(X)
    fit = amd.gsl_nls(lambda th: X @ th, y=y, start=np.zeros(p), jac=lambda th: Xf, control=dict(solver="cholesky"))
    assert fit["conv"] == 0 and fit["code_path"] == 4
    assert np.max(np.abs(fit["par"] - truth)) < 1e-8
    cov = np.asarray(fit["covar"])
    assert np.array_equal(cov, cov.T)
    assert np.max(np.abs(X.T @ (X @ cov[:, :16]) - np.eye(p)[:, :16])) < 1e-12
    monkeypatch.setenv("GSLNLS_BD_HOST_EPILOGUE", "1")
    host = amd.gsl_nls(lambda th: X @ th, y=y, start=np.zeros(p), jac=lambda th: Xf, control=dict(solver="cholesky"))
    monkeypatch.delenv("GSLNLS_BD_HOST_EPILOGUE")
    ch = np.asarray(host["covar"])
    err = float(np.max(np.abs(cov - ch) / np.sqrt(np.outer(np.diag(ch), np.diag(ch)))))
    record_parity("matrix path p=1500 covariance device vs host", err, 1e-10)
    assert err < 1e-10
    assert abs(fit["jtj_cond"] - host["jtj_cond"]) <= 1e-8 * host["jtj_cond"]


def test_jtj_in_128_column_blocks_equals_the_64_column_kernel(amd, monkeypatch):
    """From p = 384 and n = 2048 on J^T J is formed in 128 x 128 blocks (bd_syrk128_kernel: twice the products per byte staged);
    GSLNLS_BD_SYRK64=1 keeps the 64-column kernel.  Same fit of a model linear in its 400 parameters through both: the answer
    (known), the same iterations, coefficients to 1e-12, covariance X^T X cov = I."""
    p, n = 400, 2500
    rng = np.random.default_rng(p)
    X = rng.standard_normal((n, p)) + 0.05
    truth = rng.standard_normal(p)
    y = X @ truth
    Xf = np.asfortranarray(X)
    fits = {}
    for mode in ("wide", "narrow"):
        if mode == "narrow":
            monkeypatch.setenv("GSLNLS_BD_SYRK64", "1")
        else:
            monkeypatch.delenv("GSLNLS_BD_SYRK64", raising=False)
        fits[mode] = amd.gsl_nls(lambda th: X @ th, y=y, start=np.zeros(p), jac=lambda th: Xf, control=dict(solver="cholesky"))
    monkeypatch.delenv("GSLNLS_BD_SYRK64", raising=False)
    a, b = fits["wide"], fits["narrow"]
    assert a["conv"] == 0 and b["conv"] == 0 and a["niter"] == b["niter"]
    assert np.max(np.abs(a["par"] - truth)) < 1e-9 and _rel(a["par"], b["par"]) < 1e-12
    cov = np.asarray(a["covar"])
    assert np.max(np.abs(X.T @ (X @ cov[:, :16]) - np.eye(p)[:, :16])) < 1e-12
    record_parity("matrix path p=400 J'J 128- vs 64-column blocks, par", _rel(a["par"], b["par"]), 1e-12)


@pytest.mark.parametrize("loss", ["huber", "bisquare", "welsh", "hampel"])
def test_robust_losses_on_a_function_model_match_the_oracle(amd, gslref, loss):
    """gsl_nls(fn = <function>, loss = ...): the IRLS driver (src/nls_irls.c:412-546) around the matrix-path solve
    (BdFit::irls, csrc/bd_host.hpp) -- median of |r| and the re-weighting on the device, the closures on the host -- against
    the oracle's rho driver with the same closures: an exponential decay with 8 % gross outliers, weights, and a
    Jacobian closure for one half of the cases, forward differences for the other."""
    rng = np.random.Generator(np.random.PCG64(20240 + len(loss)))
    n = 400
    x = np.linspace(0.0, 4.0, n)
    truth = np.array([5.0, 1.3, 0.7])
    y = truth[0] * np.exp(-truth[1] * x) + truth[2] + 0.05 * rng.standard_normal(n)
    bad = rng.choice(n, n // 12, replace=False)
    y[bad] += rng.choice([-1.0, 1.0], bad.size) * rng.uniform(2.0, 6.0, bad.size)
    fn = lambda th: th[0] * np.exp(-th[1] * x) + th[2]  # noqa: E731
    jac = lambda th: np.stack([np.exp(-th[1] * x), -th[0] * x * np.exp(-th[1] * x), np.ones(n)], axis=1)  # noqa: E731
    start = np.array([3.0, 1.0, 0.0])
    w = 1.0 + 0.5 * np.cos(x)
    for use_jac, weights in ((True, None), (False, w)):
        fit = amd.gsl_nls(fn, y=y, start=start, jac=jac if use_jac else None, weights=weights, loss=loss,
                          control=dict(solver="cholesky"))
        ref = gslref.nls(n, 3, start, fn=lambda th: fn(th) - y, jac=jac if use_jac else None, use_jac=use_jac, weights=weights,
                         loss=loss, ctrl=gslref.control(solver="cholesky"))
        assert fit["conv"] == ref["conv"] == 0 and fit["code_path"] == 4
        assert fit["irls"]["irls_status"] == ref["irls"]["irls_status"] == 0
        assert fit["irls"]["irls_niter"] == ref["irls"]["irls_niter"], (loss, use_jac, fit["irls"], ref["irls"])
        assert _rel(fit["par"], ref["par"]) < 1e-7
        assert abs(fit["irls"]["irls_sigma"] / ref["irls"]["irls_sigma"] - 1.0) < 1e-8
        assert np.allclose(fit["irls_weights"], ref["irls_weights"], rtol=1e-6, atol=1e-9)
        # (psi = r / sigma for small residuals: a coefficient difference of 6e-10 on a model of size 5 is 4e-8 of r / sigma)
        assert np.allclose(fit["irls_psi"], ref["irls_psi"], rtol=1e-6, atol=1e-6)
        assert abs(fit["ssr"] / ref["ssr"] - 1.0) < 1e-7
        # the outliers are what the loss is there for: the coefficients come back close to the truth
        assert np.max(np.abs(fit["par"] / truth - 1.0)) < 0.05
        REPORT.append("irls %s jac=%d %.1e" % (loss, use_jac, _rel(fit["par"], ref["par"])))


def test_formula_with_100_parameters_and_a_robust_loss_matches_the_oracle(amd, gslref):
    """the formula form of the matrix path (p = 100) with loss = "huber": gslnls_nls -> bd_formula_nls -> BdFit::irls;
    5 % of the responses are pushed far off, the oracle runs the same IRLS with closures"""
    ng, n = 33, 3000
    x, y, model, jacf, start, truth = gaussians(ng, n, 4200 + ng)
    rng = np.random.Generator(np.random.PCG64(77))
    bad = rng.choice(n, n // 20, replace=False)
    y = y.copy()
    y[bad] += rng.choice([-1.0, 1.0], bad.size) * 5.0
    names, terms = [], []
    for k in range(1, ng + 1):
        names += ["a%d" % k, "m%d" % k, "s%d" % k]
        terms.append("a%d*exp(-((x-m%d)/s%d)^2)" % (k, k, k))
    names += ["c0"]
    fit = amd.gsl_nls("y ~ " + " + ".join(terms) + " + c0", data=dict(x=x, y=y), start=dict(zip(names, start)), jac=True,
                      loss="huber", control=dict(solver="cholesky"), lowering="jit")
    ref = gslref.nls(n, len(start), start, fn=lambda th: model(th) - y, jac=jacf, loss="huber",
                     ctrl=gslref.control(solver="cholesky"))
    assert fit["conv"] == ref["conv"] == 0 and fit["code_path"] == 4
    assert fit["irls"]["irls_niter"] == ref["irls"]["irls_niter"] and fit["irls"]["irls_status"] == ref["irls"]["irls_status"]
    assert _rel(fit["par"], ref["par"]) < 1e-6
    assert abs(fit["irls"]["irls_sigma"] / ref["irls"]["irls_sigma"] - 1.0) < 1e-7
    assert np.allclose(fit["irls_weights"], ref["irls_weights"], rtol=1e-5, atol=1e-8)
    REPORT.append("formula p=100 huber %.1e" % _rel(fit["par"], ref["par"]))
