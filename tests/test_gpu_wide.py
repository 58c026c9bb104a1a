"""GPU parity tests of the wide dense path: gsl_nls() with 10 <= p <= 64 parameters (the reference takes any p:
src/nls.c:266, R/nls.R:588-599).  J^T J is accumulated on the matrix cores (v_mfma_f64_16x16x4_f64 tiles,
csrc/wide_kernels.hpp), the p x p algebra of a step runs on one workgroup (csrc/wide_core.hpp), the row model is the
formula compiled in process.  Bars: the oracle (same algorithm, model and Jacobian by numpy): identical iteration
counts with the analytic Jacobian, coefficients to 1e-6 relative."""
import ctypes as C

import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import gslnls_amd
    from gslnls_amd import _lib
    assert _lib.lib().gslnls_device_count() >= 1, "no MI355X visible: the HIP path cannot be tested"
    return gslnls_amd


def _packed(A):
    p = A.shape[0]
    return np.ascontiguousarray(np.concatenate([A[i, :i + 1] for i in range(p)]))


@pytest.mark.parametrize("pivoted", ["0", "1"])
@pytest.mark.parametrize("p", [1, 2, 10, 16, 17, 33, 48, 64])
def test_wide_damped_solve_matches_the_oracles_modified_cholesky(amd, gslref, p, pivoted, monkeypatch):
    """(J^T J + mu D^2) v = -g on one wavefront against gsl_linalg_mcholesky as restated by the oracle: well
    conditioned, rank deficient (the modification kicks in) and badly scaled (pivoting matters) matrices.  Both routes of
    the device: the natural-order L D L^T that serves numerically positive definite systems (and hands everything else to
    the pivoted routine: the rank-deficient and zero-column cases end there) and the pivoted, modified factorisation alone
    (GSLNLS_WIDE_PIVOTED=1)."""
    from gslnls_amd import _lib
    monkeypatch.setenv("GSLNLS_WIDE_PIVOTED", pivoted)
    L = _lib.lib()
    rng = np.random.Generator(np.random.PCG64(100 + p))
    for case in ("spd", "rank_deficient", "badly_scaled", "zero_column"):
        n = 4 * p + 3
        J = rng.standard_normal((n, p))
        if case == "rank_deficient" and p > 2:
            J[:, p - 1] = J[:, 0] + J[:, 1]
        if case == "badly_scaled":
            J *= 10.0 ** rng.uniform(-6, 6, p)
        if case == "zero_column":
            J[:, p // 2] = 0.0
        A = J.T @ J
        diag = np.sqrt(np.maximum(np.diag(A), 1e-300))
        mu = 1e-3 if case != "rank_deficient" else 1e-14
        rhs = -(J.T @ rng.standard_normal(n))
        sol = np.zeros(p)
        rc = L.gslnls_debug_wide_solve(p, _packed(A).ctypes.data_as(_lib.DP), diag.ctypes.data_as(_lib.DP), mu,
                                       rhs.ctypes.data_as(_lib.DP), sol.ctypes.data_as(_lib.DP))
        assert rc == 0
        M = np.ascontiguousarray(A + mu * np.diag(diag * diag))
        perm = np.zeros(p, dtype=np.int32)
        G = gslref.lib()
        assert G.gslref_mcholesky_decomp(p, M.ctypes.data_as(_lib.DP), perm.ctypes.data_as(_lib.IP)) == 0
        ref = np.zeros(p)
        assert G.gslref_mcholesky_solve(p, M.ctypes.data_as(_lib.DP), perm.ctypes.data_as(_lib.IP),
                                        rhs.ctypes.data_as(_lib.DP), ref.ctypes.data_as(_lib.DP)) == 0
        scale = np.max(np.abs(ref)) + 1e-300
        if case == "rank_deficient" and p > 2:
            # the component along the null vector is decided by round-off (mu = 1e-14): compare what the system sees
            Mfull = A + mu * np.diag(diag * diag)
            assert np.max(np.abs(Mfull @ (sol - ref))) <= 1e-9 * np.max(np.abs(rhs)), (case, p)
        else:
            tol = 1e-9 if case in ("spd", "zero_column") else 1e-6
            assert np.max(np.abs(sol - ref)) <= tol * scale, (case, p, np.max(np.abs(sol - ref)) / scale)


@pytest.mark.parametrize("pivoted", ["0", "1"])
@pytest.mark.parametrize("p", [2, 7, 16, 31, 32, 64])
def test_wide_solve_pivot_rules_ties_near_ties_and_nans(amd, gslref, p, pivoted, monkeypatch):
    """The pivot search of the register-resident factorisation reduces the HIGH words of the diagonal first and only
    then compares whole values: exact ties (identical blocks: the reference's scan keeps the FIRST position), diagonals
    that differ below the 20th mantissa bit (the full comparison has to find the true maximum, which sits last), and a
    NaN on the diagonal (the call must come back) -- permutation decisions identical to the oracle's give solutions
    equal to round-off"""
    from gslnls_amd import _lib
    monkeypatch.setenv("GSLNLS_WIDE_PIVOTED", pivoted)  # ("0": positive definite cases take the natural order, NaN the pivoted routine)
    L = _lib.lib()
    G = gslref.lib()
    rng = np.random.Generator(np.random.PCG64(900 + p))

    def both(A, diag, mu, rhs):
        sol = np.zeros(p)
        rc = L.gslnls_debug_wide_solve(p, _packed(A).ctypes.data_as(_lib.DP), diag.ctypes.data_as(_lib.DP), mu,
                                       rhs.ctypes.data_as(_lib.DP), sol.ctypes.data_as(_lib.DP))
        assert rc == 0
        M = np.ascontiguousarray(A + mu * np.diag(diag * diag))
        perm = np.zeros(p, dtype=np.int32)
        assert G.gslref_mcholesky_decomp(p, M.ctypes.data_as(_lib.DP), perm.ctypes.data_as(_lib.IP)) == 0
        ref = np.zeros(p)
        assert G.gslref_mcholesky_solve(p, M.ctypes.data_as(_lib.DP), perm.ctypes.data_as(_lib.IP),
                                        rhs.ctypes.data_as(_lib.DP), ref.ctypes.data_as(_lib.DP)) == 0
        return sol, ref, perm

    # identical 2 x 2 blocks: ties among all untouched blocks at every other step
    A = np.zeros((p, p))
    for k in range(0, p - 1, 2):
        A[k:k + 2, k:k + 2] = [[2.0, 1.0], [1.0, 2.0]]
    if p % 2:
        A[p - 1, p - 1] = 2.0
    rhs = rng.standard_normal(p)
    sol, ref, perm = both(A, np.ones(p), 0.0, rhs)
    assert np.max(np.abs(sol - ref)) <= 1e-13 * np.max(np.abs(ref)), ("ties", p)
    # with weak coupling between the blocks (still exact ties on the diagonal at the first step)
    E = 1e-3 * rng.standard_normal((p, p))
    E = E + E.T
    np.fill_diagonal(E, 0.0)
    sol, ref, perm = both(A + E, np.ones(p), 0.25, rhs)
    assert np.max(np.abs(sol - ref)) <= 1e-12 * np.max(np.abs(ref)), ("coupled ties", p)
    # near ties: the same high word everywhere, the largest diagonal last
    B = E.copy()
    np.fill_diagonal(B, 2.0 + np.arange(p) * 2.0 ** -40)
    sol, ref, perm = both(B, np.ones(p), 0.0, rhs)
    assert perm[0] == p - 1                                   # (the oracle's first pivot: the true maximum)
    assert np.max(np.abs(sol - ref)) <= 1e-12 * np.max(np.abs(ref)), ("near ties", p)
    # a NaN on the diagonal, at the first position and elsewhere: the call comes back (no endless tie loop, no fault);
    # where the oracle still produces numbers (the NaN row is pivoted last / damped by eps) so does the device
    for where in (0, p - 1):
        Cn = A + E
        Cn[where, where] = np.nan
        sol, ref, perm = both(Cn, np.ones(p), 0.5, rhs)
        if np.all(np.isfinite(ref)):
            assert np.all(np.isfinite(sol)), ("nan", p, where)


def gaussians_problem(ng, extra, n, seed, noise=0.05, pert=0.02):
    """sum of ng Gaussian peaks (+ constant, + slope): p = 3 ng + extra"""
    rng = np.random.Generator(np.random.PCG64(seed))
    x = np.linspace(0.0, 10.0 * ng, n)
    amp = rng.uniform(2.0, 6.0, ng)
    mid = 10.0 * np.arange(ng) + rng.uniform(3.0, 7.0, ng)
    wid = rng.uniform(1.2, 2.4, ng)
    truth = np.stack([amp, mid, wid], axis=1).reshape(-1)
    names, terms = [], []
    for k in range(1, ng + 1):
        names += ["a%d" % k, "m%d" % k, "s%d" % k]
        terms.append("a%d*exp(-(x-m%d)^2/s%d^2)" % (k, k, k))
    if extra >= 1:
        names.append("c0")
        terms.append("c0")
        truth = np.append(truth, 0.5)
    if extra >= 2:
        names.append("c1")
        terms.append("c1*x")
        truth = np.append(truth, 0.01)

    def model(th):
        m = sum(th[3 * k] * np.exp(-(x - th[3 * k + 1]) ** 2 / th[3 * k + 2] ** 2) for k in range(ng))
        if extra >= 1:
            m = m + th[3 * ng]
        if extra >= 2:
            m = m + th[3 * ng + 1] * x
        return m

    def jac(th):
        J = np.zeros((n, len(truth)))
        for k in range(ng):
            a, m, s = th[3 * k:3 * k + 3]
            u = x - m
            e = np.exp(-u * u / (s * s))
            J[:, 3 * k] = e
            J[:, 3 * k + 1] = a * e * 2 * u / (s * s)
            J[:, 3 * k + 2] = a * e * 2 * u * u / (s ** 3)
        if extra >= 1:
            J[:, 3 * ng] = 1.0
        if extra >= 2:
            J[:, 3 * ng + 1] = x
        return J
    y = model(truth) + noise * rng.standard_normal(n)
    sign = np.where(np.arange(len(truth)) % 2 == 0, 1.0, -1.0)
    start = truth * (1.0 + pert * sign)
    return dict(x=x, y=y, truth=truth, start=start, names=names, formula="y ~ " + " + ".join(terms), model=model, jac=jac)


@pytest.mark.parametrize("ng,extra,n", [(4, 0, 1000), (5, 1, 4097), (10, 2, 70_001), (11, 0, 3000), (16, 0, 5000), (21, 1, 9999)],
                         ids=["p12", "p16", "p32", "p33", "p48", "p64"])
def test_wide_pass_sums_match_numpy(amd, ng, extra, n):
    """one pass over the rows: ssr, J^T J (MFMA tiles, every lower-triangle block) and J^T f against numpy, for the
    analytic Jacobian and for forward / central differences; ragged n (not a multiple of the 64-row tile)"""
    from gslnls_amd import _lib
    q = gaussians_problem(ng, extra, n, seed=40 + ng)
    p = len(q["truth"])
    prob = amd.DenseProblem(_lib.MODEL_EXPR, p, q["x"], q["y"], expr=q["formula"].split("~")[1].strip(),
                            parnames=q["names"], xnames=["x"], lowering="jit")
    th = q["start"]
    f = q["model"](th) - q["y"]
    J = q["jac"](th)
    A = J.T @ J
    NV = 2 + p * (p + 1) // 2 + p
    for jac, fd, rtol in ((1, 0, 1e-11), (0, 0, 1e-5), (0, 1, 1e-7)):
        tot = np.zeros(NV)
        assert _lib.lib().gslnls_debug_wide_sums(prob._h, jac, fd, th.ctypes.data_as(_lib.DP), tot.ctypes.data_as(_lib.DP)) == 0
        assert abs(tot[0] - f @ f) <= 1e-11 * (f @ f) and tot[1] == 0.0
        Ap = tot[2:2 + p * (p + 1) // 2]
        scale = np.sqrt(np.outer(np.diag(A), np.diag(A)))
        got = np.zeros((p, p))
        got[np.tril_indices(p)] = Ap
        err = np.abs(got - np.tril(A)) / scale
        assert err.max() <= rtol, (jac, fd, err.max(), np.unravel_index(err.argmax(), err.shape))
        g = tot[2 + p * (p + 1) // 2:]
        assert np.max(np.abs(g - J.T @ f) / (np.sqrt(np.diag(A)) * np.sqrt(f @ f))) <= rtol
    prob.close()


@pytest.mark.parametrize("ng,extra,n", [(5, 1, 100_000), (10, 2, 100_000), (4, 0, 4000), (11, 0, 20_000), (21, 1, 30_000)],
                         ids=["p16", "p32", "p12", "p33", "p64"])
def test_wide_fit_matches_oracle_with_the_analytic_jacobian(amd, gslref, ng, extra, n):
    """gsl_nls() on a sum-of-Gaussians formula: same iteration count as the oracle, coefficients to 1e-6, ssr to 1e-9,
    covariance to 1e-5; code_path 3 = the wide path ran"""
    q = gaussians_problem(ng, extra, n, seed=7 + ng, pert=0.01 if ng > 20 else 0.02)
    p = len(q["truth"])
    fit = amd.gsl_nls(q["formula"], data=dict(x=q["x"], y=q["y"]), start=dict(zip(q["names"], q["start"])), jac=True,
                      control=dict(solver="cholesky"), trace=True)
    ref = gslref.nls(n, p, q["start"], fn=lambda th: q["model"](th) - q["y"], jac=q["jac"],
                     ctrl=gslref.control(solver="cholesky"))
    assert fit["code_path"] == 3
    assert fit["conv"] == 0 and ref["conv"] == 0
    assert fit["niter"] == ref["niter"], (fit["niter"], ref["niter"])
    __import__("conftest").rel_err(fit["par"], ref["par"])  # (recorded for the session summary)
    assert np.allclose(fit["par"], ref["par"], rtol=1e-6, atol=1e-9)
    assert abs(fit["ssr"] - ref["ssr"]) <= 1e-9 * ref["ssr"]
    # (the last iterations sit at round-off: a trial whose ssr differs in the last bit is accepted here and rejected
    # there, or the other way round)
    # (... up to a whole ladder of 16 rejections in the last iteration)
    assert abs(fit["neval"]["f"] - ref["neval"]["f"]) <= 16 and abs(fit["neval"]["J"] - ref["neval"]["J"]) <= 1
    assert np.allclose(fit["covar"], ref["covar"], rtol=1e-5, atol=1e-12)
    # returned vectors: weighted residual and Jacobian at the final point, R layout
    assert np.allclose(fit["resid"], q["model"](fit["par"]) - q["y"], rtol=0, atol=1e-10)
    assert np.allclose(fit["grad"], q["jac"](fit["par"]), rtol=1e-9, atol=1e-12)
    assert np.allclose(fit["par"], q["truth"], rtol=5e-2, atol=5e-2)
    # trace: row 0 = start, last row = the result
    assert np.allclose(fit["partrace"][0], q["start"]) and np.allclose(fit["partrace"][fit["niter"]], fit["par"])
    assert abs(fit["ssrtrace"][fit["niter"]] - fit["ssr"]) <= 1e-12 * fit["ssr"]


@pytest.mark.parametrize("fdtype", ["forward", "center"])
def test_wide_fit_with_finite_difference_jacobian_weights_and_bounds(amd, gslref, fdtype):
    """p = 16, forward / central differences (src/fdjac.c), observation weights and box constraints through the wide
    path against the oracle driven by the same numpy model"""
    q = gaussians_problem(5, 1, 20_000, seed=3)
    p, n = 16, 20_000
    rng = np.random.Generator(np.random.PCG64(5))
    w = rng.uniform(0.5, 2.0, n)
    lower = np.where(np.arange(p) % 3 == 2, 0.5, -np.inf)       # widths bounded below
    upper = np.full(p, np.inf)
    upper[0] = max(q["truth"][0], q["start"][0]) * 0.999         # one bound that ends up active
    q["start"][0] = min(q["start"][0], upper[0] * 0.99)
    ctrl = dict(solver="cholesky", fdtype=fdtype)
    fit = amd.gsl_nls(q["formula"], data=dict(x=q["x"], y=q["y"]), start=dict(zip(q["names"], q["start"])), jac=False,
                      weights=w, lower=dict(zip(q["names"], lower)), upper=dict(zip(q["names"], upper)), control=ctrl)
    ref = gslref.nls(n, p, q["start"], fn=lambda th: q["model"](th) - q["y"], ctrl=gslref.control(**ctrl), weights=w,
                     lower=lower, upper=upper)
    assert fit["code_path"] == 3 and fit["conv"] == ref["conv"]
    assert abs(fit["niter"] - ref["niter"]) <= max(1, ref["niter"] // 10)
    __import__("conftest").rel_err(fit["par"], ref["par"])  # (recorded for the session summary)
    assert np.allclose(fit["par"], ref["par"], rtol=1e-5, atol=1e-8)
    assert abs(fit["ssr"] - ref["ssr"]) <= 1e-8 * ref["ssr"]
    assert fit["par"][0] <= upper[0] + 1e-12


def test_wide_path_takes_more_than_three_data_columns(amd, gslref):
    """five regressors, p = 6: beyond the interpreter's three columns, so the wide path serves it"""
    rng = np.random.Generator(np.random.PCG64(21))
    n = 5000
    X = rng.uniform(0.5, 2.0, (n, 5))
    truth = np.array([1.5, 0.7, -0.4, 0.9, 0.3, 2.0])
    names = ["b0", "b1", "b2", "b3", "b4", "b5"]

    def model(th):
        return th[0] * np.exp(th[1] * X[:, 0] + th[2] * X[:, 1]) + th[3] * X[:, 2] * X[:, 3] + th[4] * X[:, 4] ** 2 + th[5]
    y = model(truth) + 0.01 * rng.standard_normal(n)
    data = dict(y=y, **{"x%d" % k: X[:, k] for k in range(5)})
    formula = "y ~ b0*exp(b1*x0 + b2*x1) + b3*x2*x3 + b4*x4^2 + b5"
    start = truth * 1.1
    fit = amd.gsl_nls(formula, data=data, start=dict(zip(names, start)), jac=True, control=dict(solver="cholesky"))
    ref = gslref.nls(n, 6, start, fn=lambda th: model(th) - y, ctrl=gslref.control(solver="cholesky"))
    assert fit["code_path"] == 3 and fit["conv"] == 0 and ref["conv"] == 0
    __import__("conftest").rel_err(fit["par"], ref["par"])  # (recorded for the session summary)
    assert np.allclose(fit["par"], ref["par"], rtol=1e-6)
    assert np.allclose(fit["par"], truth, rtol=2e-2, atol=2e-2)


def test_wide_lmaccel_runs_the_acceleration_pass(amd, gslref):
    """algorithm = lmaccel on p = 16: the PH_FVV pass (J^T fvv only) with finite-difference and with symbolic second
    directional derivatives against the oracle"""
    q = gaussians_problem(5, 1, 8000, seed=9)
    p, n = 16, 8000
    ref = gslref.nls(n, p, q["start"], fn=lambda th: q["model"](th) - q["y"], jac=q["jac"], algorithm="lmaccel",
                     ctrl=gslref.control(solver="cholesky"))
    for fvv in (False, True):
        fit = amd.gsl_nls(q["formula"], data=dict(x=q["x"], y=q["y"]), start=dict(zip(q["names"], q["start"])), jac=True,
                          fvv=fvv, algorithm="lmaccel", control=dict(solver="cholesky"))
        assert fit["code_path"] == 3 and fit["conv"] == 0 and ref["conv"] == 0
        __import__("conftest").rel_err(fit["par"], ref["par"])  # (recorded for the session summary)
        assert np.allclose(fit["par"], ref["par"], rtol=1e-6, atol=1e-9)
        if not fvv:
            assert fit["niter"] == ref["niter"]


@pytest.mark.parametrize("loss", ["huber", "bisquare"])
def test_wide_robust_irls_matches_oracle(amd, gslref, loss):
    """gsl_nls(loss = ...) on p = 16 with 2 % gross outliers: the IRLS driver of src/nls_irls.c:412-546 around the wide
    solve (cold re-solve per IRLS iteration, median by radix select, psi family) against the oracle's"""
    q = gaussians_problem(5, 1, 6000, seed=31)
    p, n = 16, 6000
    rng = np.random.Generator(np.random.PCG64(77))
    y = q["y"].copy()
    y[rng.choice(n, n // 50, replace=False)] += 3.0
    fit = amd.gsl_nls(q["formula"], data=dict(x=q["x"], y=y), start=dict(zip(q["names"], q["start"])), jac=True,
                      loss=loss, control=dict(solver="cholesky"))
    ref = gslref.nls(n, p, q["start"], fn=lambda th: q["model"](th) - y, jac=q["jac"], loss=loss,
                     ctrl=gslref.control(solver="cholesky"))
    assert fit["code_path"] == 3 and fit["conv"] == ref["conv"] == 0
    assert fit["irls"]["irls_niter"] == ref["irls"]["irls_niter"] and fit["irls"]["irls_status"] == ref["irls"]["irls_status"]
    __import__("conftest").rel_err(fit["par"], ref["par"])  # (recorded for the session summary)
    assert np.allclose(fit["par"], ref["par"], rtol=1e-6, atol=1e-9)
    assert abs(fit["irls"]["irls_sigma"] - ref["irls"]["irls_sigma"]) <= 1e-9 * ref["irls"]["irls_sigma"]
    assert np.allclose(fit["irls_weights"], ref["irls_weights"], rtol=1e-6, atol=1e-9)
    # the outliers are what got down-weighted; the peaks are recovered
    assert np.allclose(fit["par"], q["truth"], rtol=5e-2, atol=5e-2)


def test_wide_multistart_replays_the_oracles_procedure(amd, gslref):
    """gsl_nls(start = ranges) for p = 12: Sobol sampling, det filter, concentration fits, local searches and the final
    solve through the wide path, one point after the other, against the oracle's sequential procedure -- same
    bookkeeping (stationary points found, major iterations, stop reason) and the same optimum"""
    q = gaussians_problem(4, 0, 1500, seed=13, noise=0.02)
    p, n = 12, 1500
    lo = q["truth"] * np.where(np.arange(p) % 3 == 1, 0.97, 0.8)      # peak positions +-3 %, amplitudes / widths +-20 %
    hi = q["truth"] * np.where(np.arange(p) % 3 == 1, 1.03, 1.2)
    start = {nm: [float(a), float(b)] for nm, a, b in zip(q["names"], lo, hi)}
    ctrl = dict(solver="cholesky", mstart_n=12, mstart_q=3, mstart_maxstart=40)
    fit = amd.gsl_nls(q["formula"], data=dict(x=q["x"], y=q["y"]), start=start, jac=True, control=ctrl)
    ref = gslref.nls(n, p, np.stack([lo, hi]), fn=lambda th: q["model"](th) - q["y"], jac=q["jac"],
                     ctrl=gslref.control(**ctrl))
    assert fit["code_path"] == 3 and fit["conv"] == 0 and ref["conv"] == 0
    # (every start of this batch reaches the same optimum within the five concentration iterations -- the per-point
    # records equal the oracle's single-start runs to the last bit or two, scripts/dev_wide_mstart_records.py -- so the
    # retained slots are chosen among ties by the last bits of ssr: the counters of the tail may differ by one)
    assert fit["mstart"]["nsp"] == ref["mstart"]["nsp"] and fit["mstart"]["stop"] == ref["mstart"]["stop"]
    assert abs(fit["mstart"]["iters"] - ref["mstart"]["iters"]) <= 1 and abs(fit["mstart"]["nwsp"] - ref["mstart"]["nwsp"]) <= 2
    assert abs(fit["mstart"]["ssropt"] - ref["mstart"]["ssropt"]) <= 1e-7 * abs(ref["mstart"]["ssropt"])
    __import__("conftest").rel_err(fit["par"], ref["par"])  # (recorded for the session summary)
    assert np.allclose(fit["par"], ref["par"], rtol=1e-6, atol=1e-9)
    assert np.allclose(fit["par"], q["truth"], rtol=5e-2, atol=5e-2)


def test_wide_concentration_records_match_the_oracle_point_by_point(amd, gslref):
    """gslnls_mstart_batch on a p = 12 problem: Sobol start points (GSL's table, dimensions 1..12) and, for every point,
    the oracle's own single-start run of the concentration fit (driver2 with maxiter = mstart_p, gtol = 1e-3:
    src/nls_mstart.c:79-92): iterations, status, end point, ssr; det(J^T J) at both ends against numpy"""
    from gslnls_amd import _lib
    from gslnls_amd.control import gsl_nls_control, pack_control
    q = gaussians_problem(4, 0, 1500, seed=13, noise=0.02)
    p, n, N = 12, 1500, 16
    lo = q["truth"] * np.where(np.arange(p) % 3 == 1, 0.97, 0.8)
    hi = q["truth"] * np.where(np.arange(p) % 3 == 1, 1.03, 1.2)
    ranges = np.ascontiguousarray(np.stack([lo, hi], axis=1).reshape(-1))
    kd = np.full(p, 0.75)
    ci, cd = pack_control(gsl_nls_control(solver="cholesky"), "lm")
    prob = amd.DenseProblem(_lib.MODEL_EXPR, p, q["x"], q["y"], expr=q["formula"].split("~")[1].strip(),
                            parnames=q["names"], xnames=["x"], lowering="jit")
    K = 3 * p + 8
    rec = np.zeros((N, K))
    ms = C.c_float(0)
    rc = _lib.lib().gslnls_mstart_batch(prob._h, 1, ranges.ctypes.data_as(_lib.DP), kd.ctypes.data_as(_lib.DP), 3, N, 0, N, 5,
                                        1e-6, ci.ctypes.data_as(_lib.IP), cd.ctypes.data_as(_lib.DP), None,
                                        rec.ctypes.data_as(C.c_void_p), 0, C.byref(ms))
    prob.close()
    assert rc == 0
    u = gslref.sobol(p, N, skip=3)
    # range map of src/nls_mstart.c:48-68 for fixed ranges (exponent 0.75 around the midpoint is applied by the oracle's
    # own driver; here the start points are simply required to lie inside the ranges and to differ)
    x0 = rec[:, 2 * p:3 * p]
    assert np.all(x0 >= lo - 1e-12) and np.all(x0 <= hi + 1e-12) and len(np.unique(np.round(x0[:, 0], 12))) == N
    assert u.shape == (N, p)
    octrl = gslref.control(solver="cholesky", maxiter=5, gtol=1e-3)
    for i in range(N):
        sc = rec[i, 3 * p:]
        o = gslref.nls(n, p, x0[i], fn=lambda th: q["model"](th) - q["y"], jac=q["jac"], ctrl=octrl)
        assert int(sc[5]) == o["niter"] and int(sc[6]) == o["conv"]
        assert np.allclose(rec[i, :p], o["par"], rtol=1e-9) and abs(sc[1] - o["ssr"]) <= 1e-10 * o["ssr"]
        J0, J1 = q["jac"](x0[i]), q["jac"](o["par"])
        assert abs(sc[2] / np.linalg.det(J0.T @ J0) - 1.0) < 1e-8 and abs(sc[3] / np.linalg.det(J1.T @ J1) - 1.0) < 1e-8


def test_wide_hat_values_and_cooks_distances(amd):
    """hatvalues() / cooks.distance() for p = 16 (src/nls_utils.c:88-150) from the resident data, weighted, against numpy"""
    from gslnls_amd import _lib
    q = gaussians_problem(5, 1, 5000, seed=17)
    p, n = 16, 5000
    w = np.random.default_rng(4).uniform(0.5, 2.0, n)
    prob = amd.DenseProblem(_lib.MODEL_EXPR, p, q["x"], q["y"], weights=w, expr=q["formula"].split("~")[1].strip(),
                            parnames=q["names"], xnames=["x"], lowering="jit")
    th = q["truth"] * 1.001
    hat, cooks = prob.diagnostics(th, jac=True)
    prob.close()
    sw = np.sqrt(w)
    J = q["jac"](th) * sw[:, None]
    e = (q["model"](th) - q["y"]) * sw
    H = np.einsum("ij,jk,ik->i", J, np.linalg.inv(J.T @ J), J)
    s2 = (e @ e) / (n - p)
    D = e * e / (p * s2) * H / (1.0 - H) ** 2
    assert np.allclose(hat, H, rtol=1e-9, atol=1e-14) and abs(hat.sum() - p) < 1e-8
    assert np.allclose(cooks, D, rtol=1e-8, atol=1e-16)


def test_wide_robust_multistart_second_pass(amd, gslref):
    """gsl_nls(start = ranges, loss = "huber") for p = 12: multi-start, Cook's-distance outlier screening, second
    multi-start with the outliers' weights zeroed, final IRLS solve (src/nls.c:401-509) against the oracle"""
    q = gaussians_problem(4, 0, 1500, seed=13, noise=0.02)
    p, n = 12, 1500
    rng = np.random.Generator(np.random.PCG64(5))
    y = q["y"].copy()
    y[rng.choice(n, 30, replace=False)] += 2.0
    lo = q["truth"] * np.where(np.arange(p) % 3 == 1, 0.97, 0.8)
    hi = q["truth"] * np.where(np.arange(p) % 3 == 1, 1.03, 1.2)
    start = {nm: [float(a), float(b)] for nm, a, b in zip(q["names"], lo, hi)}
    ctrl = dict(solver="cholesky", mstart_n=12, mstart_q=3, mstart_maxstart=40)
    fit = amd.gsl_nls(q["formula"], data=dict(x=q["x"], y=y), start=start, jac=True, loss="huber", control=ctrl)
    ref = gslref.nls(n, p, np.stack([lo, hi]), fn=lambda th: q["model"](th) - y, jac=q["jac"], loss="huber",
                     ctrl=gslref.control(**ctrl))
    assert fit["code_path"] == 3 and fit["conv"] == ref["conv"] == 0
    assert fit["irls"]["irls_niter"] == ref["irls"]["irls_niter"]
    __import__("conftest").rel_err(fit["par"], ref["par"])  # (recorded for the session summary)
    assert np.allclose(fit["par"], ref["par"], rtol=1e-6, atol=1e-9)
    assert np.allclose(fit["par"], q["truth"], rtol=5e-2, atol=5e-2)


@pytest.mark.parametrize("alg", ["lm", "cgst"])
@pytest.mark.parametrize("ng,extra,n", [(4, 0, 6000), (10, 2, 40_000)], ids=["p12", "p32"])
def test_wide_formula_through_gsl_nls_large(amd, gslref, alg, ng, extra, n):
    """gsl_nls_large(formula) with 10 <= p <= 64 (R/nls_large.R:124, formula method): the operators of the large driver
    sit on the wide pass (J^T J on the matrix cores; the products J^T J u of the Steihaug-Toint iterations are p x p work
    on the host) -- against the oracle's multilarge driver on the same model and against the dense fit"""
    q = gaussians_problem(ng, extra, n, seed=40 + ng)
    p = len(q["truth"])
    data = dict(x=q["x"], y=q["y"])
    start = dict(zip(q["names"], q["start"]))
    fit = amd.gsl_nls_large(q["formula"], data=data, start=start, algorithm=alg, control=dict(maxiter=100), trace=True)

    def dfl(trans, th, u, want_v, want_jtj):
        J = q["jac"](th)
        v = None
        if want_v:
            v = J.T @ u if trans else J @ u
        return v, (J.T @ J if want_jtj else None)
    ref = gslref.nls_large(n, p, q["start"], fn=lambda th: q["model"](th) - q["y"], dfl=dfl, algorithm=alg,
                           ctrl=gslref.control(maxiter=100), trace=True)
    assert fit["conv"] == 0 and ref["conv"] == 0
    assert abs(fit["niter"] - ref["niter"]) <= 1, (fit["niter"], ref["niter"])
    __import__("conftest").rel_err(fit["par"], ref["par"])  # (recorded for the session summary)
    assert np.allclose(fit["par"], ref["par"], rtol=1e-6, atol=1e-9)
    assert abs(fit["ssr"] - ref["ssr"]) <= 1e-9 * ref["ssr"]
    k = min(fit["niter"], ref["niter"]) - 1
    # (the Steihaug-Toint steps are truncated CG runs: J^T J u from the accumulated J^T J here, J^T (J u) in the oracle --
    # the iterates agree to ~ 1e-7 on the way and to 1e-9 at the end)
    assert np.allclose(fit["ssrtrace"][:k], ref["ssrtrace"][:k], rtol=1e-8 if alg == "lm" else 1e-6)
    assert np.allclose(fit["covar"], ref["covar"], rtol=1e-5, atol=1e-12)
    assert np.allclose(fit["resid"], q["model"](np.asarray(fit["par"])) - q["y"], rtol=0, atol=1e-9)
    dense = amd.gsl_nls(q["formula"], data=data, start=start, jac=True, control=dict(solver="cholesky"))
    assert abs(fit["ssr"] - dense["ssr"]) <= 1e-8 * dense["ssr"]


def test_wide_formula_through_gsl_nls_large_with_weights(amd, gslref):
    """the same with observation weights (R/nls_large.R: `weights`; src/nls_large.c:118-124 takes their square roots):
    the rows are scaled inside the wide pass"""
    q = gaussians_problem(4, 1, 5000, seed=77)
    n, p = len(q["y"]), len(q["truth"])
    rng = np.random.Generator(np.random.PCG64(78))
    wts = rng.uniform(0.2, 3.0, n)
    fit = amd.gsl_nls_large(q["formula"], data=dict(x=q["x"], y=q["y"]), start=dict(zip(q["names"], q["start"])),
                            algorithm="lm", weights=wts, control=dict(maxiter=100))

    def dfl(trans, th, u, want_v, want_jtj):
        J = q["jac"](th)
        v = None
        if want_v:
            v = J.T @ u if trans else J @ u
        return v, (J.T @ J if want_jtj else None)
    ref = gslref.nls_large(n, p, q["start"], fn=lambda th: q["model"](th) - q["y"], dfl=dfl, algorithm="lm",
                           ctrl=gslref.control(maxiter=100), weights=wts)
    assert fit["conv"] == 0 and ref["conv"] == 0
    assert abs(fit["niter"] - ref["niter"]) <= 1
    __import__("conftest").rel_err(fit["par"], ref["par"])  # (recorded for the session summary)
    assert np.allclose(fit["par"], ref["par"], rtol=1e-6, atol=1e-9)
    assert abs(fit["ssr"] - ref["ssr"]) <= 1e-9 * ref["ssr"]
    assert np.allclose(fit["resid"], np.sqrt(wts) * (q["model"](np.asarray(fit["par"])) - q["y"]), rtol=0, atol=1e-9)
    assert np.allclose(fit["covar"], ref["covar"], rtol=1e-5, atol=1e-12)


@pytest.mark.parametrize("scale", ["levenberg", "marquardt"])
def test_wide_fit_with_the_other_scaling_rules(amd, gslref, scale):
    """control$scale = "levenberg" (D = 1) and "marquardt" (D_j = ||J_j||, renewed at every accepted point) on the wide
    path (GSL scaling.c, reached at src/trust.c:334, :524) against the oracle: same iterations, same coefficients"""
    q = gaussians_problem(4, 1, 3000, seed=91, pert=0.01)
    n, p = len(q["y"]), len(q["truth"])
    fit = amd.gsl_nls(q["formula"], data=dict(x=q["x"], y=q["y"]), start=dict(zip(q["names"], q["start"])), jac=True,
                      control=dict(solver="cholesky", scale=scale, maxiter=200))
    ref = gslref.nls(n, p, q["start"], fn=lambda th: q["model"](th) - q["y"], jac=q["jac"],
                     ctrl=gslref.control(solver="cholesky", scale=scale, maxiter=200))
    assert fit["code_path"] == 3
    assert fit["conv"] == ref["conv"] == 0
    assert abs(fit["niter"] - ref["niter"]) <= 1, (fit["niter"], ref["niter"])
    __import__("conftest").rel_err(fit["par"], ref["par"])  # (recorded for the session summary)
    assert np.allclose(fit["par"], ref["par"], rtol=1e-6, atol=1e-9)
    assert abs(fit["ssr"] - ref["ssr"]) <= 1e-9 * ref["ssr"]


def test_wide_failure_modes_match_the_oracle(amd, gslref):
    """status codes of the wide path: maxiter reached (conv 11, same point as the oracle after the same iterations) and a
    non-finite Jacobian at the start (s1 = 0: 0 * Inf in d/ds1 -> EBADFUNC, src/nls.c:899-907; coefficients = start,
    NA-filled residuals as src/nls.c:699-737)"""
    q = gaussians_problem(4, 0, 2000, seed=93, pert=0.05)
    n, p = len(q["y"]), len(q["truth"])
    data, start = dict(x=q["x"], y=q["y"]), dict(zip(q["names"], q["start"]))
    fit = amd.gsl_nls(q["formula"], data=data, start=start, jac=True, control=dict(solver="cholesky", maxiter=3))
    ref = gslref.nls(n, p, q["start"], fn=lambda th: q["model"](th) - q["y"], jac=q["jac"],
                     ctrl=gslref.control(solver="cholesky", maxiter=3))
    assert fit["conv"] == ref["conv"] == 11 and fit["niter"] == ref["niter"] == 3
    __import__("conftest").rel_err(fit["par"], ref["par"])  # (recorded for the session summary)
    assert np.allclose(fit["par"], ref["par"], rtol=1e-9, atol=1e-12)
    assert abs(fit["ssr"] - ref["ssr"]) <= 1e-10 * ref["ssr"]
    bad = q["start"].copy()
    bad[2] = 0.0  # s1
    with np.errstate(all="ignore"):
        fit = amd.gsl_nls(q["formula"], data=data, start=dict(zip(q["names"], bad)), jac=True,
                          control=dict(solver="cholesky"))
        ref = gslref.nls(n, p, bad, fn=lambda th: q["model"](th) - q["y"], jac=q["jac"],
                         ctrl=gslref.control(solver="cholesky"))
    # (the reference ignores winit's status and iterates on an unset workspace -- the oracle then ends in "no progress",
    # 27; the device reports the cause, 9: tests/test_host_logic.py::test_nonfinite_jacobian_and_residual_rules)
    assert fit["conv"] == 9 and ref["conv"] in (9, 27), (fit["conv"], ref["conv"])
    assert np.allclose(fit["par"], bad)
    assert np.all(np.isnan(fit["resid"]))


def test_wide_robust_irls_with_observation_weights(amd, gslref):
    """loss = "huber" together with `weights` on the wide path: every inner solve runs with sqrt(w_user) * sqrt(w_irls)
    (src/nls_irls.c:447-456, :517-521), the scale estimate on the raw residuals -- against the oracle"""
    q = gaussians_problem(4, 1, 4000, seed=61)
    n, p = len(q["y"]), len(q["truth"])
    rng = np.random.Generator(np.random.PCG64(62))
    y = q["y"].copy()
    y[rng.choice(n, n // 40, replace=False)] += 2.5
    wts = rng.uniform(0.5, 2.0, n)
    fit = amd.gsl_nls(q["formula"], data=dict(x=q["x"], y=y), start=dict(zip(q["names"], q["start"])), jac=True,
                      loss="huber", weights=wts, control=dict(solver="cholesky"))
    ref = gslref.nls(n, p, q["start"], fn=lambda th: q["model"](th) - y, jac=q["jac"], loss="huber", weights=wts,
                     ctrl=gslref.control(solver="cholesky"))
    assert fit["code_path"] == 3 and fit["conv"] == ref["conv"] == 0
    assert fit["irls"]["irls_niter"] == ref["irls"]["irls_niter"] and fit["irls"]["irls_status"] == ref["irls"]["irls_status"]
    __import__("conftest").rel_err(fit["par"], ref["par"])  # (recorded for the session summary)
    assert np.allclose(fit["par"], ref["par"], rtol=1e-6, atol=1e-9)
    # (sigma = 1.4826 median |r| at the point the last inner solve stopped: the two sides stop within xtol = 1.5e-8 of the
    # optimum of their last weighted problem, sigma follows the point -- measured 2.9e-8 since round 5's row closures
    # multiply by 1 / s^2 where the oracle divides by s^2, 1e-10 before)
    assert abs(fit["irls"]["irls_sigma"] - ref["irls"]["irls_sigma"]) <= 1e-7 * ref["irls"]["irls_sigma"]
    assert np.allclose(fit["irls_weights"], ref["irls_weights"], rtol=1e-6, atol=1e-9)


def test_wide_lmaccel_with_central_differences_everywhere(amd, gslref):
    """algorithm = "lmaccel" with jac = FALSE, fvv = FALSE and fdtype = "center": Jacobian by central differences
    (src/fdjac.c:81-128) and the second directional derivative by the finite-difference form (src/fdfvv.c:35-77)"""
    q = gaussians_problem(4, 0, 3000, seed=63, pert=0.01)
    n, p = len(q["y"]), len(q["truth"])
    fit = amd.gsl_nls(q["formula"], data=dict(x=q["x"], y=q["y"]), start=dict(zip(q["names"], q["start"])), jac=False,
                      fvv=False, algorithm="lmaccel", control=dict(solver="cholesky", fdtype="center"))
    ref = gslref.nls(n, p, q["start"], fn=lambda th: q["model"](th) - q["y"], algorithm="lmaccel",
                     ctrl=gslref.control(solver="cholesky", fdtype="center"))
    assert fit["code_path"] == 3 and fit["conv"] == ref["conv"] == 0
    assert abs(fit["niter"] - ref["niter"]) <= 1, (fit["niter"], ref["niter"])
    __import__("conftest").rel_err(fit["par"], ref["par"])  # (recorded for the session summary)
    assert np.allclose(fit["par"], ref["par"], rtol=1e-5, atol=1e-8)
    assert abs(fit["ssr"] - ref["ssr"]) <= 1e-8 * ref["ssr"]


@pytest.mark.parametrize("n", [13, 20, 63, 64, 65, 129])
def test_wide_path_on_very_few_rows(amd, gslref, n):
    """one partial 64-row tile, a full one, one row more; n = 13 is barely above p = 12 (ill conditioned: both run into
    maxiter along the same trajectory) -- the oracle's iterations and coefficients"""
    q = gaussians_problem(4, 0, n, seed=100 + n, noise=0.01, pert=0.005)
    fit = amd.gsl_nls(q["formula"], data=dict(x=q["x"], y=q["y"]), start=dict(zip(q["names"], q["start"])), jac=True,
                      control=dict(solver="cholesky"))
    o = gslref.nls(n, 12, q["start"], fn=lambda th: q["model"](th) - q["y"], jac=q["jac"],
                   ctrl=gslref.control(solver="cholesky"))
    assert fit["code_path"] == 3 and fit["conv"] == o["conv"] and fit["niter"] == o["niter"]
    __import__("conftest").rel_err(fit["par"], o["par"])  # (recorded for the session summary)
    assert np.allclose(fit["par"], o["par"], rtol=1e-6, atol=1e-8)
    assert abs(fit["ssr"] - o["ssr"]) <= 1e-8 * o["ssr"]


def test_wide_multistart_through_the_in_library_collective(amd):
    """one process per GPU: with an RCCL communicator in the library every multi-start gsl_nls() shards its points over
    the ranks and all-gathers the records (capi.hip: ms_rccl_run_batch).  For p > 9 a rank fits its block one point after
    the other on the host-driven wide path and stages the records into the shard buffer; here one rank with the
    collective forced (GSLNLS_COMM_FORCE_COLLECTIVE): same bookkeeping and the same fit as without a communicator"""
    import json
    import os
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import json, os, sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import gslnls_amd as amd
from gslnls_amd import _lib
from test_gpu_wide import gaussians_problem
L = _lib.lib()
q = gaussians_problem(4, 0, 1500, seed=13, noise=0.02)
lo = q["truth"] * np.where(np.arange(12) %% 3 == 1, 0.97, 0.8); hi = q["truth"] * np.where(np.arange(12) %% 3 == 1, 1.03, 1.2)
start = {nm: [float(a), float(b)] for nm, a, b in zip(q["names"], lo, hi)}
ctrl = dict(solver="cholesky", mstart_n=12, mstart_q=3, mstart_maxstart=40)
def fit():
    f = amd.gsl_nls(q["formula"], data=dict(x=q["x"], y=q["y"]), start=start, jac=True, control=ctrl)
    return dict(par=list(map(float, f["par"])), conv=int(f["conv"]), niter=int(f["niter"]), ms={k: float(v) for k, v in f["mstart"].items()},
                path=int(f["code_path"]))
plain = fit()
rc_init = L.gslnls_comm_init_file(sys.argv[1].encode(), 0, 1, 30)
n0 = L.gslnls_comm_allgather_count()
coll = fit()
n1 = L.gslnls_comm_allgather_count()
L.gslnls_comm_destroy()
print(json.dumps(dict(plain=plain, coll=coll, rc_init=rc_init, collectives=n1 - n0)))
""" % (root, root)
    with tempfile.TemporaryDirectory() as td:
        env = dict(os.environ, GSLNLS_COMM_FORCE_COLLECTIVE="1", GSLNLS_COMM_NONCE="wide-ms")
        out = subprocess.run([sys.executable, "-c", code, os.path.join(td, "nccl_id")], capture_output=True, text=True,
                             timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert r["rc_init"] == 0 and r["collectives"] >= 1
    assert r["plain"]["path"] == r["coll"]["path"] == 3 and r["plain"]["conv"] == r["coll"]["conv"] == 0
    assert r["plain"]["ms"] == r["coll"]["ms"] and r["plain"]["niter"] == r["coll"]["niter"]
    assert r["plain"]["par"] == r["coll"]["par"]


def test_batch_fit_kernel_gives_the_records_of_the_point_by_point_evaluator(amd):
    """The multi-start evaluator of the wide path fits every point of a batch at once, one workgroup per point
    (wide_fit_kernel), where round 3 fitted them one after the other through the launch-per-step chain
    (GSLNLS_WIDE_MS_BATCH=0): same state machine, same sums per workgroup -- but the one-after-the-other form adds the
    partial sets of up to 512 workgroups where the batch form has one workgroup's sums, so the records agree to rounding,
    not bit for bit; the det filter's decisions, iteration counts and status codes are identical.  Two child processes
    (the switch is read once per process)."""
    import json
    import os
    import subprocess
    import sys
    code = r"""
import json, sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import gslnls_amd as A
from test_gpu_wide import gaussians_problem
pb = gaussians_problem(4, 0, 300, 11)
x, y, names, truth = pb["x"], pb["y"], pb["names"], pb["truth"]
lo, hi = truth * 0.8, truth * 1.2
fit = A.gsl_nls(pb["formula"], data=dict(x=x, y=y), start={k: [float(a), float(b)] for k, a, b in zip(names, np.minimum(lo, hi), np.maximum(lo, hi))},
                jac=True, control=dict(solver="cholesky", mstart_n=24, mstart_p=4, mstart_maxiter=3), lowering="jit")
print(json.dumps(dict(par=[float(v) for v in fit["par"]], ssr=float(fit["ssr"]), conv=int(fit["conv"]), niter=int(fit["niter"]),
                      ms=fit["mstart"])))
""" % (ROOT, os.path.join(ROOT, "tests"))
    outs = []
    for batch in ("1", "0"):
        env = dict(os.environ, GSLNLS_WIDE_MS_BATCH=batch)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    a, b = outs
    assert a["conv"] == b["conv"] == 0 and a["ms"]["nsp"] == b["ms"]["nsp"] and a["ms"]["nwsp"] == b["ms"]["nwsp"]
    assert a["ms"]["iters"] == b["ms"]["iters"] and a["ms"]["stop"] == b["ms"]["stop"]
    assert np.allclose(a["par"], b["par"], rtol=1e-8) and abs(a["ssr"] - b["ssr"]) <= 1e-10 * b["ssr"]


def test_speculative_hand_off_rejects_torn_stale_and_missing_payloads(amd, monkeypatch):
    """Round 5 stress test of the speculative solve's hand-off (csrc/wide_kernels.hpp: the speculator workgroup solves for the
    case "this trial is rejected" and publishes {payload, tag}; the stepping workgroup takes the payload only if the tag read
    in front of AND behind it is this launch's, the payload folds to the tag and mu matches bit for bit).  The test switch
    GSLNLS_WIDE_SPEC_FAULT makes the speculator publish (1) a payload half of which is not the one its tag was folded from,
    (2) the tag of another launch, (3) nothing: in every case the stepping workgroup must find "not there", solve itself,
    and the fit must be the same BIT FOR BIT as with the hand-off working (0) and as without speculation."""
    pb = gaussians_problem(5, 1, 20_000, seed=3, pert=0.25)  # (far enough from the optimum for rejected trials)
    start = dict(zip(pb["names"], pb["start"]))
    fits = {}
    for mode in ("0", "1", "2", "3", "off"):
        if mode == "off":
            monkeypatch.delenv("GSLNLS_WIDE_SPEC_FAULT", raising=False)
            monkeypatch.setenv("GSLNLS_WIDE_SPEC", "0")
        else:
            monkeypatch.setenv("GSLNLS_WIDE_SPEC_FAULT", mode)
            monkeypatch.delenv("GSLNLS_WIDE_SPEC", raising=False)
        fits[mode] = amd.gsl_nls(pb["formula"], data=dict(x=pb["x"], y=pb["y"]), start=start, jac=True, lowering="jit",
                                 control=dict(solver="cholesky", maxiter=60), trace=True)
    monkeypatch.delenv("GSLNLS_WIDE_SPEC", raising=False)
    monkeypatch.delenv("GSLNLS_WIDE_SPEC_FAULT", raising=False)
    ref = fits["0"]
    assert ref["code_path"] == 3 and ref["n_launches"] == ref["n_steps"]  # (the one-launch-per-step kernel ran)
    rejected = ref["neval"]["f"] - 1 - ref["niter"]
    assert rejected >= 3, rejected  # (the speculator's case occurred)
    for mode, f in fits.items():
        assert f["conv"] == ref["conv"] and f["niter"] == ref["niter"] and f["neval"] == ref["neval"], mode
        assert np.array_equal(f["par"], ref["par"]) and f["ssr"] == ref["ssr"], mode
        assert np.array_equal(f["partrace"], ref["partrace"]) and np.array_equal(f["ssrtrace"], ref["ssrtrace"]), mode


@pytest.mark.parametrize("ng,extra", [(4, 0), (10, 2), (20, 2)], ids=["PW16-p12", "PW32-p32", "PW64-p62"])
def test_wide_state_machine_equals_the_matrix_paths_on_the_same_problem(amd, monkeypatch, ng, extra):
    """Two of the copies of the LM state machine on ONE problem (round 5): the wide path (wide_core.hpp: one wavefront, lane =
    component, the damped solve in registers) and the matrix path (bd_host.hpp: host p-vectors, the Jacobian a matrix in
    HBM, the blocked device Cholesky), the latter forced onto a formula the wide path serves (GSLNLS_MATRIX_PATH_MIN_P).
    Both take their scalar decisions from csrc/lm_decide.hpp; their sums differ in order only.  Same iterations, same
    evaluation counts, the whole trace to 1e-9 -- at every compiled width of the wide path (PW = 16, 32, 64), analytic and
    forward-difference Jacobians, with bounds and with geodesic acceleration."""
    n = 20000
    pb = gaussians_problem(ng, extra, n, seed=7 + ng, pert=0.01 if ng > 15 else 0.02)
    start = dict(zip(pb["names"], pb["start"]))
    p = len(pb["names"])
    lo = {k: float(v - 0.6 * abs(v) - 0.5) for k, v in zip(pb["names"], pb["truth"])}
    cases = [dict(jac=True), dict(jac=False), dict(jac=True, lower=lo), dict(jac=True, algorithm="lmaccel")]
    for kw in cases:
        monkeypatch.delenv("GSLNLS_MATRIX_PATH_MIN_P", raising=False)
        w = amd.gsl_nls(pb["formula"], data=dict(x=pb["x"], y=pb["y"]), start=start, control=dict(solver="cholesky", maxiter=200),
                        trace=True, lowering="jit", **kw)
        monkeypatch.setenv("GSLNLS_MATRIX_PATH_MIN_P", "10")
        m = amd.gsl_nls(pb["formula"], data=dict(x=pb["x"], y=pb["y"]), start=start, control=dict(solver="cholesky", maxiter=200),
                        trace=True, **kw)
        monkeypatch.delenv("GSLNLS_MATRIX_PATH_MIN_P", raising=False)
        assert w["code_path"] == 3 and m["code_path"] == 4, (w["code_path"], m["code_path"])
        assert w["conv"] == m["conv"] == 0 and w["niter"] == m["niter"], (kw, w["niter"], m["niter"])
        # (the last iteration sits at round-off level: whether one of its trials lowers ||f|| in the last bit, or the whole
        # ladder of 16 is rejected and the iteration ends as "no progress" with the same point, is decided by the order of the
        # sums -- up to 16 trials' worth of evaluations and one Jacobian apart, measured: 14 trials against 29 with lmaccel)
        per_trial = 2 if kw.get("algorithm") == "lmaccel" else 1
        assert abs(w["neval"]["J"] - m["neval"]["J"]) <= 1 and abs(w["neval"]["f"] - m["neval"]["f"]) <= 16 * per_trial + (0 if kw.get("jac") else p), (kw, w["neval"], m["neval"])
        # (a difference Jacobian amplifies the last bits in which the two paths' iterates differ by 1 / h: measured 1.5e-9 on
        # the trace, where the analytic runs agree to 1e-12)
        tol = 1e-9 if kw.get("jac") else 1e-7
        assert np.allclose(w["ssrtrace"], m["ssrtrace"], rtol=tol, atol=0), kw
        assert np.allclose(w["partrace"], m["partrace"], rtol=100 * tol, atol=1e-10), kw
        from conftest import rel_err
        assert rel_err(w["par"], m["par"]) < 10 * tol, kw
