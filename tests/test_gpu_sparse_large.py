"""GPU tests of gsl_nls_large with a caller-supplied sparse Jacobian (csrc/sparse_large.hpp), through the C ABI.

The reference's own sparse example and pin: penalty function I with p = 500 (README.md:1040-1146: model
c(sqrt(1e-5) (theta - 1), sum(theta^2) - 1/4), Jacobian rbind(Diagonal(sqrt(1e-5)), 2 t(theta)) as dgCMatrix),
residual sum of squares 0.004778845 for both lm and cgst."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import gslnls_amd
    from gslnls_amd import _lib
    assert _lib.lib().gslnls_device_count() >= 1, "no MI355X visible: the HIP path cannot be tested"
    return gslnls_amd


def penalty(p, fmt):
    a = np.sqrt(1e-5)

    def fn(th):
        return np.concatenate([a * (th - 1.0), [np.sum(th ** 2) - 0.25]])

    def jac(th):
        J = sp.vstack([sp.identity(p, format="csr") * a, sp.csr_matrix(2.0 * th.reshape(1, -1))])
        if fmt == "dense":
            return J.toarray()
        if fmt == "coo_dup":
            # triplets with duplicates that must be summed (dgTMatrix semantics): split every entry in two halves
            c = J.tocoo()
            return sp.coo_matrix((np.concatenate([c.data / 2, c.data / 2]),
                                  (np.concatenate([c.row, c.row]), np.concatenate([c.col, c.col]))), shape=c.shape)
        return J.asformat(fmt)
    return fn, jac


def oracle_penalty(gslref, p, algorithm, ctrl=None):
    a = np.sqrt(1e-5)

    def fn(th):
        return np.concatenate([a * (th - 1.0), [np.sum(th ** 2) - 0.25]])

    def dfl(trans, th, u, want_v, want_jtj):
        J = np.vstack([a * np.eye(p), 2.0 * th.reshape(1, -1)])
        v = (J.T @ u if trans else J @ u) if want_v else None
        return v, (J.T @ J if want_jtj else None)
    return gslref.nls_large(p + 1, p, np.arange(1.0, p + 1), fn=fn, dfl=dfl, algorithm=algorithm, ctrl=ctrl)


@pytest.mark.parametrize("fmt", ["csc", "csr", "coo", "coo_dup", "dense"])
def test_penalty_p500_cgst_readme_pin(amd, gslref, pins, fmt):
    p = 500
    fn, jac = penalty(p, fmt)
    fit = amd.gsl_nls_large(fn, y=np.zeros(p + 1), start=np.arange(1.0, p + 1), algorithm="cgst", jac=jac,
                            control=dict(maxiter=500))
    ref = oracle_penalty(gslref, p, "cgst", gslref.control(maxiter=500))
    assert fit["conv"] == 0 and ref["conv"] == 0
    assert abs(fit["ssr"] - 0.004778845) < 5e-10          # README.md:1100-1101
    assert abs(fit["ssr"] - ref["ssr"]) <= 1e-10 * ref["ssr"]
    # flat valley (see the lm test below): the CG recurrences run on the device, whose norms are block reductions
    # where the oracle adds sequentially -- the same 207 iterations end at points whose ssr agree to 1e-10 while a
    # few coordinates differ in the 4th digit; with the host loop (GSLNLS_LARGE_CG=host, sequential norms like the
    # oracle's) they agree to 1e-6
    __import__("conftest").rel_err(fit["par"], ref["par"])  # (recorded for the session summary)
    assert np.allclose(fit["par"], ref["par"], rtol=2e-3)
    assert fit["niter"] == ref["niter"]


def test_penalty_p500_cgst_host_loop_matches_oracle_coordinates(amd, gslref, monkeypatch):
    monkeypatch.setenv("GSLNLS_LARGE_CG", "host")
    p = 500
    fn, jac = penalty(p, "csc")
    fit = amd.gsl_nls_large(fn, y=np.zeros(p + 1), start=np.arange(1.0, p + 1), algorithm="cgst", jac=jac,
                            control=dict(maxiter=500))
    ref = oracle_penalty(gslref, p, "cgst", gslref.control(maxiter=500))
    assert fit["niter"] == ref["niter"] and fit["neval"]["dfu"] > 0
    __import__("conftest").rel_err(fit["par"], ref["par"])  # (recorded for the session summary)
    assert np.allclose(fit["par"], ref["par"], rtol=1e-6, atol=1e-9)


def test_penalty_p500_lm_sparse_jtj(amd, gslref):
    p = 500
    fn, jac = penalty(p, "csc")
    fit = amd.gsl_nls_large(fn, y=np.zeros(p + 1), start=np.arange(1.0, p + 1), algorithm="lm", jac=jac,
                            control=dict(maxiter=500))
    ref = oracle_penalty(gslref, p, "lm", gslref.control(maxiter=500))
    assert fit["conv"] == 0 and ref["conv"] == 0
    assert abs(fit["ssr"] - 0.004778845) < 5e-10          # README.md:1068-1069
    assert abs(fit["ssr"] - ref["ssr"]) <= 1e-9 * ref["ssr"]
    # the valley is flat (singular values sqrt(1e-5) against 2|theta|): both stop by xtol at points whose ssr
    # agree to 1e-9 while the coordinates still differ in the 4th digit (the true minimiser has all theta_i equal)
    __import__("conftest").rel_err(fit["par"], ref["par"])  # (recorded for the session summary)
    assert np.allclose(fit["par"], ref["par"], rtol=2e-3)
    # ~230 iterations creeping along the valley until a step is shorter than xtol: the count moves with the last bits
    # of the damped solves (host factorisation 228, device factorisation 220-221, oracle 232)
    assert abs(fit["niter"] - ref["niter"]) <= 0.08 * ref["niter"]
    # covariance = (J^T J)^-1 from the dense J^T J assembled on the device
    th = fit["par"]
    J = np.vstack([np.sqrt(1e-5) * np.eye(p), 2.0 * th.reshape(1, -1)])
    assert np.allclose(np.asarray(fit["covar"]), np.linalg.inv(J.T @ J), rtol=1e-6)


def test_penalty_p5_small(amd, gslref):
    """the same model at the size of the reference's large-path unit tests (unit_tests_gslnls.R:109-131: n, p ~ 5)"""
    p = 5
    fn, jac = penalty(p, "csc")
    fit = amd.gsl_nls_large(fn, y=np.zeros(p + 1), start=np.arange(1.0, p + 1), algorithm="cgst", jac=jac)
    ref = oracle_penalty(gslref, p, "cgst")
    assert fit["conv"] == ref["conv"] == 0
    __import__("conftest").rel_err(fit["par"], ref["par"])  # (recorded for the session summary)
    assert np.allclose(fit["par"], ref["par"], rtol=1e-6)
    assert fit["niter"] == ref["niter"]


def test_weighted_banded_problem_long_rows_and_columns(amd):
    """a random banded + one dense row + one dense column Jacobian (exercises the thread-per-segment kernel, the
    long-segment list and row weighting): linear model, so the answer is the weighted least-squares solution"""
    rng = np.random.Generator(np.random.PCG64(7))
    n, p = 4000, 600
    A = sp.diags([rng.standard_normal(n), rng.standard_normal(n - 1), rng.standard_normal(n - 2)], [0, -1, -2],
                 shape=(n, p), format="lil")
    A[n - 1, :] = rng.standard_normal(p)          # dense row (600 entries > SP_LONG)
    A[:, 3] = rng.standard_normal((n, 1))         # dense column (4000 entries)
    for k in range(p):                            # make sure every column is hit below the band as well
        A[p + (k * 5) % (n - p), k] = 1.0 + rng.random()
    A = A.tocsr()
    truth = rng.standard_normal(p)
    w = 0.5 + rng.random(n)
    y = A @ truth + 0.01 * rng.standard_normal(n)
    fit = amd.gsl_nls_large(lambda th: A @ th, y=y, start=np.zeros(p), algorithm="cgst", jac=lambda th: A, weights=w,
                            control=dict(maxiter=200))
    # Weights on the large path scale f only (GSL's multilarge eval_f; gsl_df_large hands J over unweighted,
    # src/nls_large.c:629-646): the fit stops where A^T (sqrt(w) o (A theta - y)) = 0, i.e. at the least-squares solution
    # with weights sqrt(w) -- what the oracle's multilarge driver converges to as well (the linear model makes it a closed
    # form; rounds 1-2 scaled the rows of J too and landed on the properly weighted solution, which the reference does not)
    S = sp.diags(np.sqrt(w))
    sol = np.linalg.solve((A.T @ S @ A).toarray(), A.T @ (np.sqrt(w) * y))
    assert fit["conv"] == 0
    # (the trust-region ratio compares a decrease of sum w r^2 with a model built from the unweighted J: the iteration
    # stops by gtol / xtol a little earlier than on a consistent problem)
    assert np.allclose(fit["par"], sol, rtol=2e-4, atol=2e-5)
    r = np.sqrt(w) * (A @ fit["par"] - y)
    assert np.allclose(np.asarray(fit["resid"]), r, rtol=1e-9, atol=1e-12)
    assert abs(fit["ssr"] - r @ r) <= 1e-10 * (r @ r)


def test_device_cg_equals_host_cg(amd, monkeypatch):
    """The Steihaug-Toint step with its p-sized recurrences on the device (csrc/sparse_cg.hpp, the default) against
    the host loop it replaces (GSLNLS_LARGE_CG=host: one round trip per CG iteration): same iterations, same
    evaluation counts (the products the reference would have made), results equal to rounding of the norms."""
    p = 500
    fn, jac = penalty(p, "csc")
    rng = np.random.Generator(np.random.PCG64(11))
    n2, p2 = 3000, 400
    A = sp.random(n2, p2, density=0.02, random_state=np.random.RandomState(5), format="csr") + sp.eye(n2, p2, format="csr")
    y2 = A @ rng.standard_normal(p2) + 0.01 * rng.standard_normal(n2)
    out = {}
    for mode in ("device", "host"):
        if mode == "host":
            monkeypatch.setenv("GSLNLS_LARGE_CG", "host")
        else:
            monkeypatch.delenv("GSLNLS_LARGE_CG", raising=False)
        a = amd.gsl_nls_large(fn, y=np.zeros(p + 1), start=np.arange(1.0, p + 1), algorithm="cgst", jac=jac,
                              control=dict(maxiter=500))
        b = amd.gsl_nls_large(lambda th: A @ th, y=y2, start=np.zeros(p2), algorithm="cgst", jac=lambda th: A,
                              control=dict(maxiter=100))
        out[mode] = (a, b)
    for k in (0, 1):
        d, h = out["device"][k], out["host"][k]
        assert d["conv"] == h["conv"] == 0
        assert d["niter"] == h["niter"]
        # products with J: the same count, or one CG iteration more or less where ||r|| / ||g|| crosses 1e-6 within
        # rounding of the norms (seen: 95 against 93 on the linear problem)
        assert d["neval"]["f"] == h["neval"]["f"] and d["neval"]["df2"] == h["neval"]["df2"]
        assert abs(d["neval"]["dfu"] - h["neval"]["dfu"]) <= 2
        assert abs(d["ssr"] - h["ssr"]) <= 1e-10 * h["ssr"]
        assert np.allclose(d["par"], h["par"], rtol=2e-3 if k == 0 else 1e-8, atol=1e-11)  # k = 0: the flat valley


def test_callback_errors_surface(amd):
    def bad_fn(th):
        raise RuntimeError("model blew up")
    with pytest.raises(RuntimeError, match="model blew up"):
        amd.gsl_nls_large(bad_fn, y=np.zeros(3), start=np.ones(2), algorithm="cgst", jac=lambda th: np.ones((3, 2)))


def test_weighted_sparse_fit_follows_the_oracles_multilarge_driver(amd, gslref):
    """gsl_nls_large(fn, jac, weights) against the oracle's multilarge driver with the same weights (which scale f only):
    same iterations, same coefficients -- and NOT the properly weighted least-squares solution"""
    rng = np.random.Generator(np.random.PCG64(5))
    n, p = 400, 12
    A = (sp.random(n, p, density=0.3, random_state=7, format="csr")
         + sp.vstack([sp.identity(p), sp.csr_matrix((n - p, p))])).tocsr()
    truth = rng.standard_normal(p)
    w = 0.5 + rng.random(n)
    y = A @ truth + 0.05 * rng.standard_normal(n)
    Ad = A.toarray()

    def dfl(trans, th, u, want_v, want_jtj):
        v = None
        if want_v:
            v = Ad.T @ u if trans else Ad @ u
        return v, (Ad.T @ Ad if want_jtj else None)
    for alg in ("cgst", "lm"):
        fit = amd.gsl_nls_large(lambda th: A @ th, y=y, start=np.zeros(p), algorithm=alg, jac=lambda th: A, weights=w,
                                control=dict(maxiter=200))
        ref = gslref.nls_large(n, p, np.zeros(p), fn=lambda th: Ad @ th - y, dfl=dfl, algorithm=alg,
                               ctrl=gslref.control(maxiter=200), weights=w)
        assert fit["conv"] == 0 and ref["conv"] == 0
        assert abs(fit["niter"] - ref["niter"]) <= 1
        __import__("conftest").rel_err(fit["par"], ref["par"])  # (recorded for the session summary)
        assert np.allclose(fit["par"], ref["par"], rtol=1e-7, atol=1e-9)
        assert abs(fit["ssr"] - ref["ssr"]) <= 1e-9 * ref["ssr"]
        wls = np.linalg.solve(Ad.T @ np.diag(w) @ Ad, Ad.T @ (w * y))
        assert np.max(np.abs(np.asarray(fit["par"]) - wls)) > 1e-5


def test_weighted_lm_at_p_450_through_the_device_factorisation(amd, gslref):
    """penalty function I at p = 450 with observation weights and algorithm = "lm": every damped solve runs on the device
    (p >= 400) from J^T J as it sits there (unweighted rows: the weights scale f only) -- the oracle's iterations to 8 %
    (a flat valley, cf. the p = 500 test), its ssr to 1e-6"""
    p = 450
    fn, jac = penalty(p, "csc")
    rng = np.random.Generator(np.random.PCG64(450))
    w = rng.uniform(0.5, 2.0, p + 1)
    fit = amd.gsl_nls_large(fn, y=np.zeros(p + 1), start=np.arange(1.0, p + 1), algorithm="lm", jac=jac, weights=w,
                            control=dict(maxiter=500))
    a = np.sqrt(1e-5)

    def dfl(trans, th, u, want_v, want_jtj):
        J = np.vstack([a * np.eye(p), 2.0 * th.reshape(1, -1)])
        v = (J.T @ u if trans else J @ u) if want_v else None
        return v, (J.T @ J if want_jtj else None)
    ref = gslref.nls_large(p + 1, p, np.arange(1.0, p + 1), fn=fn, dfl=dfl, algorithm="lm", ctrl=gslref.control(maxiter=500),
                           weights=w)
    assert fit["conv"] == 0 and ref["conv"] == 0
    # (with weights on f only the iteration does not minimise sum w r^2 consistently: the stopping points in the flat
    # valley agree to ~ 1e-7 in ssr)
    assert abs(fit["ssr"] - ref["ssr"]) <= 1e-6 * ref["ssr"]
    assert abs(fit["niter"] - ref["niter"]) <= 0.08 * ref["niter"] + 1
    __import__("conftest").rel_err(fit["par"], ref["par"])  # (recorded for the session summary)
    assert np.allclose(fit["par"], ref["par"], rtol=5e-3)


def test_lm_with_jtj_kept_on_the_device_is_the_fit_with_the_host_copy(amd, monkeypatch):
    """From the device-factorisation threshold on, J^T J of the sparse operators stays where sp_jtj_kernel formed it: the
    damped solve reads it in place and the row sums of the predicted reduction v^T J^T J v are taken there
    (mchol_symv_kernel: the host loop's products in the host loop's order, rounded separately).  GSLNLS_LARGE_JTJ_HOST=1
    brings the 2 MB back to the host at every accepted point as before: same iterations, same coefficients, bit for bit."""
    p = 500
    fn, jac = penalty(p, "csc")
    fits = {}
    for mode in ("device", "host"):
        if mode == "host":
            monkeypatch.setenv("GSLNLS_LARGE_JTJ_HOST", "1")
        else:
            monkeypatch.delenv("GSLNLS_LARGE_JTJ_HOST", raising=False)
        fits[mode] = amd.gsl_nls_large(fn, y=np.zeros(p + 1), start=np.arange(1.0, p + 1), algorithm="lm", jac=jac,
                                       control=dict(maxiter=500))
    monkeypatch.delenv("GSLNLS_LARGE_JTJ_HOST", raising=False)
    a, b = fits["device"], fits["host"]
    assert a["conv"] == b["conv"] == 0 and a["niter"] == b["niter"]
    assert np.array_equal(np.asarray(a["par"]), np.asarray(b["par"])) and a["ssr"] == b["ssr"]
    assert abs(a["ssr"] - 0.004778845) < 5e-10


@pytest.mark.parametrize("alg,dense", [("cgst", False), ("lm", False), ("cgst", True)])
def test_large_path_covariance_from_the_device_equals_the_host_routine(amd, monkeypatch, alg, dense):
    """gsl_multilarge_nlinear_covar at a fit's end (src/nls_large.c:255): from p = 65 on (J^T J)^-1 comes from the device -- the
    damped solve's factorisation, L^-1 column by column, X^T X on the matrix cores (round 5; the host's factor-and-invert
    was 15 ms of every call at p = 500).  GSLNLS_BD_HOST_EPILOGUE=1 keeps the host routine: same fit, covariance to 1e-10
    of its scale."""
    p = 500
    fn, jac = penalty(p, "csc")
    if dense:
        Jd = np.asfortranarray(np.vstack([np.sqrt(1e-5) * np.eye(p), np.zeros((1, p))]))

        def jac(th, _J=Jd):  # noqa: F811
            _J[p, :] = 2.0 * th
            return _J
    fits = {}
    for mode in ("device", "host"):
        if mode == "host":
            monkeypatch.setenv("GSLNLS_BD_HOST_EPILOGUE", "1")
        else:
            monkeypatch.delenv("GSLNLS_BD_HOST_EPILOGUE", raising=False)
        fits[mode] = amd.gsl_nls_large(fn, y=np.zeros(p + 1), start=np.arange(1.0, p + 1), algorithm=alg, jac=jac,
                                       control=dict(maxiter=500))
    monkeypatch.delenv("GSLNLS_BD_HOST_EPILOGUE", raising=False)
    a, b = fits["device"], fits["host"]
    assert a["conv"] == b["conv"] == 0 and a["niter"] == b["niter"]
    assert np.array_equal(np.asarray(a["par"]), np.asarray(b["par"]))
    ca, cb = np.asarray(a["covar"]), np.asarray(b["covar"])
    assert np.all(np.isfinite(ca)) and np.array_equal(ca, ca.T)
    scale = np.sqrt(np.outer(np.diag(cb), np.diag(cb)))
    err = float(np.max(np.abs(ca - cb) / scale))
    from conftest import record_parity
    record_parity("large path p=500 %s%s covariance device vs host" % (alg, " dense" if dense else ""), err, 1e-10)
    assert err < 1e-10, err
