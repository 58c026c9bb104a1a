"""GPU parity of the gsl_nls_large path (gsl_multilarge_nlinear: lm on the normal equations, Steihaug-Toint CG)
through the C ABI: matrix-free EVAL / fused J^T J u passes against numpy, whole fits against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = float(np.finfo(float).eps ** 0.25)


@pytest.fixture(scope="module")
def amd():
    import gslnls_amd
    from gslnls_amd import _lib
    assert _lib.lib().gslnls_device_count() >= 1
    return gslnls_amd


def glm_data(n, p, seed=20250928):
    """a_ij ~ U(-1,1)/sqrt(p), theta* ~ N(0, 0.25^2), y = f(theta*)(1 + 0.01 N(0,1)), theta0 = 0"""
    rng = np.random.Generator(np.random.PCG64(seed))
    A = rng.uniform(-1.0, 1.0, size=(n, p)) / np.sqrt(p)
    th = 0.25 * rng.standard_normal(p)
    y = np.exp(A @ th) * (1.0 + 0.01 * rng.standard_normal(n))
    return A, y, th


@pytest.mark.parametrize("p,n", [(64, 5000), (64, 33), (32, 4097), (16, 1000), (16, 7), (32, 2)])
def test_glm_passes_match_numpy(amd, p, n):
    """one EVAL pass and one fused J^T J u pass (ragged n, several tile shapes) against dense numpy algebra"""
    import ctypes as C
    from gslnls_amd import _lib
    A, y, th = glm_data(n, p)
    rng = np.random.default_rng(1)
    x = 0.1 * rng.standard_normal(p)
    u = rng.standard_normal(p)
    w = rng.uniform(0.5, 2.0, n)
    prob = amd.LargeProblem(5, p, A, y, weights=w)
    # a zero-iteration "fit" is not exposed; use the solver with maxiter = 1 from x and compare its first quantities
    m = np.exp(A @ x)
    f = np.sqrt(w) * (m - y)
    J = m[:, None] * A
    fit = prob.solve(x, "cgst", dict(maxiter=1), trace=True)
    assert abs(fit["chisq_init"] - f @ f) <= 1e-11 * (f @ f)
    # time_pass runs EVAL at x then `reps` passes; use it to exercise both kernels on the same data, then compare a
    # full LM step: the first lm iteration solves (J^T J + mu D^2) v = -J^T f with quantities from the EVAL pass
    fit_lm = prob.solve(x, "lm", dict(maxiter=1), trace=True)
    g = J.T @ f
    JTJ = J.T @ J
    D = np.sqrt(np.diag(JTJ))
    mu = 1e-3 * np.max(np.diag(JTJ) / D ** 2)
    v = np.linalg.solve(JTJ + mu * np.diag(D ** 2), -g)
    f1 = np.sqrt(w) * (np.exp(A @ (x + v)) - y)
    if f1 @ f1 < f @ f:  # first trial accepted (always, for this smooth problem)
        assert np.allclose(fit_lm["partrace"][1], x + v, rtol=1e-8, atol=1e-11)
        assert abs(fit_lm["ssrtrace"][1] - f1 @ f1) <= 1e-10 * (f1 @ f1)
    assert prob.time_pass(0, x) > 0 and prob.time_pass(1, x, u) > 0
    prob.close()


@pytest.mark.parametrize("alg", ["cgst", "lm"])
@pytest.mark.parametrize("p,n", [(64, 20000), (64, 20003), (64, 131), (16, 3000), (32, 5001)])
def test_glm_fit_matches_oracle(amd, gslref, alg, p, n):
    A, y, th = glm_data(n, p)
    fit = amd.gsl_nls_large("glmexp", A=A, y=y, start=np.zeros(p), algorithm=alg, trace=True)
    o = gslref.nls_large(n, p, np.zeros(p), rowdata=dict(model=gslref.MODEL_GLMEXP, x=A, y=y), algorithm=alg,
                         trace=True)
    assert fit["conv"] == 0 and o["conv"] == 0
    assert fit["niter"] == o["niter"]
    __import__("conftest").rel_err(fit["par"], o["par"])  # (recorded for the session summary)
    assert np.allclose(fit["par"], o["par"], rtol=1e-6, atol=1e-9)
    assert abs(fit["ssr"] - o["ssr"]) <= 1e-9 * o["ssr"]
    assert np.allclose(fit["par"], th, atol=0.05)                     # recovers the generating parameters
    # same trajectory: ssr after every outer iteration (the last iteration sits at round-off level, where the
    # number of rejected trials -- and with it the evaluation counters -- depends on the last bits)
    k = fit["niter"]
    assert np.allclose(fit["ssrtrace"][:k], o["ssrtrace"][:k], rtol=1e-9)
    assert np.allclose(fit["partrace"][:k], o["partrace"][:k], rtol=1e-6, atol=1e-9)
    assert abs(fit["neval"]["df2"] - o["neval"]["df2"]) <= 1 and fit["neval"]["f"] >= k + 1
    assert np.allclose(fit["covar"], o["covar"], rtol=1e-5, atol=1e-12)
    assert np.allclose(fit["resid"], o["resid"], atol=1e-7)


@pytest.mark.parametrize("alg", ["cgst", "lm"])
def test_one_shot_entry_equals_create_solve_destroy(amd, alg):
    """gslnls_nls_large is the C_nls_large counterpart (src/nls_large.c:66-75: one call, everything built and torn
    down inside it); it must give bit for bit what gslnls_large_create / _solve / _destroy give, weights included"""
    import ctypes as C
    from gslnls_amd import _lib
    from gslnls_amd.control import gsl_nls_control
    from gslnls_amd.nls_large import pack_control_large
    n, p = 3001, 16
    A, y, th = glm_data(n, p)
    w = np.random.default_rng(3).uniform(0.5, 2.0, n)
    prob = amd.LargeProblem(5, p, A, y, weights=w)
    ref = prob.solve(np.zeros(p), alg)
    prob.close()
    ci, cd = pack_control_large(gsl_nls_control(), alg, False)
    Ac, yc, wc, st = np.ascontiguousarray(A), np.ascontiguousarray(y), np.ascontiguousarray(w), np.zeros(p)
    m = _lib.Model(5, p, p, Ac.ctypes.data_as(C.c_void_p), 0)
    out = dict(par=np.zeros(p), covar=np.zeros((p, p), order="F"), resid=np.zeros(n))
    res = _lib.LargeResult()
    res.par, res.covar, res.resid = (out[k].ctypes.data_as(_lib.DP) for k in ("par", "covar", "resid"))
    rc = _lib.lib().gslnls_nls_large(C.byref(m), yc.ctypes.data_as(C.c_void_p), n, st.ctypes.data_as(_lib.DP),
                                     wc.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(_lib.IP), cd.ctypes.data_as(_lib.DP),
                                     C.byref(res))
    assert rc == 0 and res.conv == 0 and res.niter == ref["niter"]
    assert np.array_equal(out["par"], ref["par"]) and res.ssr == ref["ssr"]
    assert np.array_equal(out["covar"], ref["covar"]) and np.array_equal(out["resid"], ref["resid"])
    assert [res.neval[k] for k in range(4)] == [ref["neval"][k] for k in ("f", "dfu", "df2", "fvv")]
    # a model the large path does not know is refused by the one-shot entry like by create
    bad = _lib.Model(77, p, p, Ac.ctypes.data_as(C.c_void_p), 0)
    assert _lib.lib().gslnls_nls_large(C.byref(bad), yc.ctypes.data_as(C.c_void_p), n, st.ctypes.data_as(_lib.DP), None,
                                       ci.ctypes.data_as(_lib.IP), cd.ctypes.data_as(_lib.DP), C.byref(res)) == _lib.E_UNSUPPORTED


def test_unit_tests_3_x_large(amd, gslref, nist):
    """unit_tests_gslnls.R:108-131: gsl_nls_large on Misra1a (3.1.1 lm, 3.1.4 weights = 1) and Steihaug-Toint"""
    q = nist["Misra1a"]
    tgt = np.array(list(q["target"].values()))
    x, y = np.array(q["data"]["x"]), np.array(q["data"]["y"])
    for alg, w in (("lm", None), ("lm", np.ones(14)), ("cgst", None)):
        fit = amd.gsl_nls_large(q["formula"], data=q["data"], start=q["start"], algorithm=alg, weights=w, trace=True)
        assert fit["conv"] == 0 and np.all(np.abs(fit["par"] - tgt) <= TOL), (alg, fit["par"])
        o = gslref.nls_large(14, 2, [500.0, 1e-4], rowdata=dict(model=gslref.MODEL_MISRA1A, x=x, y=y), algorithm=alg,
                             weights=w)
        __import__("conftest").rel_err(fit["par"], o["par"])  # (recorded for the session summary)
        assert fit["niter"] == o["niter"] and np.allclose(fit["par"], o["par"], rtol=1e-6)
        assert fit["algorithm"] == ("steihaug-toint" if alg == "cgst" else "levenberg-marquardt")
    with pytest.raises(NotImplementedError):
        amd.gsl_nls_large(q["formula"], data=q["data"], start=q["start"], algorithm="dogleg")


@pytest.mark.parametrize("name", ["Misra1b", "Thurber", "Gauss2"])
@pytest.mark.parametrize("alg", ["lm", "cgst"])
def test_large_path_on_compiled_expressions(amd, nist, name, alg):
    """gsl_nls_large(formula) for formulas without a hand-written device model: the matrix-free passes run on the
    compiled expression (R/nls_large.R:124 formula method); certified NIST values at the reference's tolerance"""
    q = nist[name]
    data = {k: np.asarray(v, dtype=np.float64) for k, v in q["data"].items()}
    fit = amd.gsl_nls_large(q["formula"], data=data, start=q["start"], algorithm=alg, control=dict(maxiter=200))
    tgt = np.array(list(q["target"].values()))
    assert fit["conv"] == 0
    err = np.abs(np.asarray(fit["par"]) - tgt)
    assert np.all((err <= TOL) | (err <= 2e-5 * np.abs(tgt))), (fit["par"], tgt)
    dense = amd.gsl_nls(q["formula"], data=data, start=q["start"], jac=True, control=dict(solver="cholesky"))
    assert abs(fit["ssr"] - dense["ssr"]) <= 1e-7 * dense["ssr"]


def test_c3_full_size_properties(amd):
    """BASELINE configs[2] at full size (n = 1e7, p = 64, A = 5.12 GB resident): properties that need no CPU
    reference -- the cgst fit recovers the generating parameters, a second run is bitwise identical (fixed-shape
    reductions), and lm (full J^T J through the fp64 MFMA SYRK + host Cholesky) lands on the same optimum as the
    matrix-free Steihaug-Toint path."""
    n, p = 10_000_000, 64
    A, y, th = glm_data(n, p)
    prob = amd.LargeProblem(5, p, A, y)
    del A
    x0 = np.zeros(p)
    a = prob.solve(x0, "cgst", want_resid=False)
    b = prob.solve(x0, "cgst", want_resid=False)
    c = prob.solve(x0, "lm", want_resid=False)
    prob.close()
    assert a["conv"] == 0 and c["conv"] == 0
    assert np.array_equal(a["par"], b["par"]) and a["ssr"] == b["ssr"] and a["niter"] == b["niter"]
    # y = f(theta*) (1 + 0.01 N(0,1)): the estimate sits within a few standard errors (~ 0.01 / sqrt(n / p)) of theta*
    assert np.max(np.abs(a["par"] - th)) < 1e-3
    assert np.max(np.abs(c["par"] - a["par"])) < 1e-6
    assert abs(c["ssr"] - a["ssr"]) <= 1e-9 * a["ssr"]
    # relative residual variance 0.01^2 on responses exp(a.theta) = O(1)
    assert 0.5e-4 < a["ssr"] / n < 2e-4


@pytest.mark.parametrize("pivoted", ["0", "1"])
@pytest.mark.parametrize("p", [3, 65, 130, 500, 777, 1100, 2050])
def test_device_modified_cholesky_matches_the_oracle(amd, gslref, p, pivoted, monkeypatch):
    """The damped normal equations of the gsl_nls_large lm step on the device (csrc/mchol_device.hip: panels of pivot
    steps by one workgroup, grid-wide trailing updates, implicit permutation) against gsl_linalg_mcholesky as restated
    by the oracle: well conditioned, badly scaled (pivoting matters), rank deficient (the modification kicks in; what the
    system sees is compared), identical diagonal blocks (exact ties: the first position wins) and a zero column.
    pivoted = 0 (default): the natural-order blocked Cholesky serves the numerically positive definite cases and hands the
    rank-deficient and zero-column ones to the pivoted routine; pivoted = 1: the pivoted routine alone."""
    from gslnls_amd import _lib
    monkeypatch.setenv("GSLNLS_LARGE_CHOL_PIVOTED", pivoted)
    L = _lib.lib()
    G = gslref.lib()
    rng = np.random.Generator(np.random.PCG64(4200 + p))
    for case in ("spd", "badly_scaled", "rank_deficient", "ties", "zero_column"):
        n = 2 * p + 3
        J = rng.standard_normal((n, p))
        if case == "badly_scaled":
            J *= 10.0 ** rng.uniform(-5, 5, p)
        if case == "rank_deficient" and p > 2:
            J[:, p - 1] = J[:, 0] + J[:, 1]
        if case == "zero_column":
            J[:, p // 2] = 0.0
        A = np.ascontiguousarray(J.T @ J)
        if case == "ties":
            A = np.zeros((p, p))
            for k in range(0, p - 1, 2):
                A[k:k + 2, k:k + 2] = [[2.0, 1.0], [1.0, 2.0]]
            if p % 2:
                A[p - 1, p - 1] = 2.0
            E = 1e-3 * rng.standard_normal((p, p))
            E = E + E.T
            np.fill_diagonal(E, 0.0)
            A = np.ascontiguousarray(A + E)
        diag = np.sqrt(np.maximum(np.diag(A), 1e-300))
        mu = 1e-14 if case == "rank_deficient" else 1e-3
        rhs = rng.standard_normal(p) * np.max(np.abs(A))
        sol = np.zeros(p)
        assert L.gslnls_debug_mchol_solve(p, A.ctypes.data_as(_lib.DP), diag.ctypes.data_as(_lib.DP), mu,
                                          rhs.ctypes.data_as(_lib.DP), sol.ctypes.data_as(_lib.DP)) == 0
        M = np.ascontiguousarray(A + mu * np.diag(diag * diag))
        Mf = M.copy()
        perm = np.zeros(p, dtype=np.int32)
        assert G.gslref_mcholesky_decomp(p, M.ctypes.data_as(_lib.DP), perm.ctypes.data_as(_lib.IP)) == 0
        ref = np.zeros(p)
        assert G.gslref_mcholesky_solve(p, M.ctypes.data_as(_lib.DP), perm.ctypes.data_as(_lib.IP),
                                        rhs.ctypes.data_as(_lib.DP), ref.ctypes.data_as(_lib.DP)) == 0
        scale = np.max(np.abs(ref)) + 1e-300
        if case == "rank_deficient" and p > 2:
            # (the solution is ~ 1 / mu along the null vector: what the system sees, relative to |M| |sol|)
            assert np.max(np.abs(Mf @ (sol - ref))) <= 1e-8 * np.max(np.abs(rhs)) + 1e-12 * np.max(np.abs(Mf)) * scale, (case, p)
        else:
            tol = 1e-6 if case == "badly_scaled" else 1e-9
            assert np.max(np.abs(sol - ref)) <= tol * scale, (case, p, np.max(np.abs(sol - ref)) / scale)


@pytest.mark.parametrize("p", [70, 333, 500, 1030])
def test_back_substitution_in_one_launch_and_resident_matrix_give_the_same_bits(amd, monkeypatch, p):
    """csrc/mchol_device.hip: cholb_backall_kernel (the whole back substitution in one launch, y in LDS, the block's triangle
    in registers) performs the sums of the launch-per-block form in the same order -- the two solutions are compared bit
    for bit; and the solve with J^T J resident in device memory (how the lm step calls it: diag and rhs go up, nothing
    else) equals the solve that uploads the damped matrix."""
    import ctypes as C
    from gslnls_amd import _lib
    L = _lib.lib()
    rng = np.random.Generator(np.random.PCG64(99 + p))
    J = rng.standard_normal((p + 40, p))
    A = np.ascontiguousarray(J.T @ J)
    diag = np.sqrt(np.diag(A)).copy()
    rhs = rng.standard_normal(p)
    mu = 1e-3
    sols = {}
    for mode in ("one", "blocks"):
        if mode == "blocks":
            monkeypatch.setenv("GSLNLS_LARGE_BACK_BLOCKS", "1")
        else:
            monkeypatch.delenv("GSLNLS_LARGE_BACK_BLOCKS", raising=False)
        sol = np.zeros(p)
        assert L.gslnls_debug_mchol_solve(p, A.ctypes.data_as(_lib.DP), diag.ctypes.data_as(_lib.DP), mu,
                                          rhs.ctypes.data_as(_lib.DP), sol.ctypes.data_as(_lib.DP)) == 0
        sols[mode] = sol
    monkeypatch.delenv("GSLNLS_LARGE_BACK_BLOCKS", raising=False)
    assert np.array_equal(sols["one"], sols["blocks"])
    dA = C.c_void_p()
    assert L.gslnls_debug_device_alloc(C.byref(dA), A.nbytes) == 0
    try:
        assert L.gslnls_debug_device_copy(dA, A.ctypes.data_as(C.c_void_p), A.nbytes, 1) == 0
        solr = np.zeros(p)
        assert L.gslnls_debug_mchol_solve_resident(p, dA, diag.ctypes.data_as(_lib.DP), mu, rhs.ctypes.data_as(_lib.DP),
                                                   solr.ctypes.data_as(_lib.DP)) == 0
        assert np.array_equal(solr, sols["one"])
        back = np.zeros_like(A)
        assert L.gslnls_debug_device_copy(back.ctypes.data_as(C.c_void_p), dA, A.nbytes, 0) == 0
        assert np.array_equal(back, A)  # (left as it was)
    finally:
        L.gslnls_debug_device_free(dA)
    M = A + mu * np.diag(diag * diag)
    res = float(np.linalg.norm(M @ solr - rhs) / np.linalg.norm(rhs))
    from conftest import record_parity
    record_parity("blocked Cholesky, relative residual, p = %d" % p, res)
    assert res < 1e-11


def test_large_lm_takes_the_device_factorisation_from_the_threshold_on(amd, gslref):
    """gsl_nls_large(algorithm = "lm") on the GLM family at p = 64 with the threshold lowered to 1 (every step's solve on
    the device) gives the fit of the host factorisation: same iterations, coefficients to round-off (the threshold is
    read once per process: two child processes)"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fits = {}
    for mode, thr in (("host", "0"), ("device", "1")):
        code = (
            "import os, sys, json, numpy as np\n"
            "os.environ['GSLNLS_LARGE_CHOL_DEVICE_MIN'] = %r\n"
            "sys.path.insert(0, %r)\n"
            "sys.path.insert(0, os.path.join(%r, 'tests'))\n"
            "import gslnls_amd as A\n"
            "from test_gpu_large import glm_data\n"
            "X, y, th = glm_data(20003, 64)\n"
            "fit = A.gsl_nls_large('glmexp', A=X, y=y, start=np.zeros(64), algorithm='lm')\n"
            "print(json.dumps(dict(par=list(map(float, fit['par'])), niter=int(fit['niter']), conv=int(fit['conv']),"
            " ssr=float(fit['ssr']))))\n"
        ) % (thr, root, root)
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        fits[mode] = json.loads(out.stdout.strip().splitlines()[-1])
    assert fits["host"]["conv"] == 0 and fits["device"]["conv"] == 0
    assert fits["host"]["niter"] == fits["device"]["niter"]
    assert np.allclose(fits["host"]["par"], fits["device"]["par"], rtol=1e-8, atol=1e-10)
    assert abs(fits["host"]["ssr"] - fits["device"]["ssr"]) <= 1e-10 * fits["host"]["ssr"]


@pytest.mark.parametrize("alg", ["cgst", "lm"])
def test_glm_fit_with_weights_matches_oracle(amd, gslref, alg):
    """observation weights on the large path: they scale f only -- the reference's gsl_df_large never weights J
    (src/nls_large.c:629-633; GSL multilarge eval_f applies sqrt(w) to f) -- against the oracle, which keeps the quirk"""
    p, n = 16, 3000
    A, y, th = glm_data(n, p)
    rng = np.random.Generator(np.random.PCG64(404))
    w = rng.uniform(0.5, 2.0, n)
    fit = amd.gsl_nls_large("glmexp", A=A, y=y, start=np.zeros(p), algorithm=alg, weights=w)
    o = gslref.nls_large(n, p, np.zeros(p), rowdata=dict(model=gslref.MODEL_GLMEXP, x=A, y=y), algorithm=alg, weights=w)
    assert fit["conv"] == 0 and o["conv"] == 0
    assert fit["niter"] == o["niter"]
    __import__("conftest").rel_err(fit["par"], o["par"])  # (recorded for the session summary)
    assert np.allclose(fit["par"], o["par"], rtol=1e-6, atol=1e-9)
    assert abs(fit["ssr"] - o["ssr"]) <= 1e-9 * o["ssr"]
    assert np.allclose(fit["covar"], o["covar"], rtol=1e-5, atol=1e-12)


@pytest.mark.parametrize("p", [65, 100, 128, 129, 500, 1100])
def test_round5_forms_of_the_damped_solve_give_the_same_bits(amd, monkeypatch, p):
    """csrc/mchol_device.hip, round 5: the one-launch solve for 64 < p <= 128 (cholb_small_kernel), one launch per panel step
    (cholb_step_kernel), the right-hand side a column behind, the pipelined back substitution, rhs | diag in the kernel
    arguments or read in place -- each has a developer switch that selects the form before it.  With J^T J resident (the lm
    step's and the matrix path's call) every form returns the same solution, bit for bit, and it solves the system."""
    import ctypes as C
    from gslnls_amd import _lib
    L = _lib.lib()
    rng = np.random.Generator(np.random.PCG64(7 + p))
    J = rng.standard_normal((p + 40, p))
    A = np.ascontiguousarray(J.T @ J)
    diag = np.sqrt(np.diag(A)).copy()
    rhs = rng.standard_normal(p)
    mu = 1e-3
    dA = C.c_void_p()
    assert L.gslnls_debug_device_alloc(C.byref(dA), A.nbytes) == 0
    switches = ["GSLNLS_LARGE_SMALL_OFF", "GSLNLS_LARGE_STEP_V1", "GSLNLS_LARGE_BACK_STEPWISE", "GSLNLS_LARGE_UPLOAD_COPY",
                "GSLNLS_LARGE_PANEL_V1", "GSLNLS_LARGE_BACK_V1"]
    try:
        assert L.gslnls_debug_device_copy(dA, A.ctypes.data_as(C.c_void_p), A.nbytes, 1) == 0
        sols = {}
        for mode in ["default"] + switches:
            for s in switches:
                monkeypatch.delenv(s, raising=False)
            if mode != "default":
                monkeypatch.setenv(mode, "1")
            sol = np.zeros(p)
            assert L.gslnls_debug_mchol_solve_resident(p, dA, diag.ctypes.data_as(_lib.DP), mu, rhs.ctypes.data_as(_lib.DP),
                                                       sol.ctypes.data_as(_lib.DP)) == 0
            sols[mode] = sol
        for s in switches:
            monkeypatch.delenv(s, raising=False)
    finally:
        L.gslnls_debug_device_free(dA)
    for mode in switches:
        assert np.array_equal(sols["default"], sols[mode]), mode
    M = A + mu * np.diag(diag ** 2)
    assert np.linalg.norm(M @ sols["default"] - rhs) <= 1e-12 * np.linalg.norm(rhs) * np.linalg.cond(M)


@pytest.mark.parametrize("p", [70, 128, 300])
def test_resident_solve_of_a_rank_deficient_matrix_reaches_the_pivoted_routine(amd, p):
    """a matrix the natural-order factorisation must refuse (rank p - 5): the one-launch solve for p <= 128 and the panel steps
    beyond raise the flag, the pivoted modified Cholesky (gsl_linalg_mcholesky's) produces the solution -- the same as for
    the matrix brought from the host, bit for bit, and finite"""
    import ctypes as C
    from gslnls_amd import _lib
    L = _lib.lib()
    rng = np.random.Generator(np.random.PCG64(1234 + p))
    J = rng.standard_normal((p - 5, p))
    A = np.ascontiguousarray(J.T @ J)
    diag = np.sqrt(np.diag(A)).copy()
    rhs = rng.standard_normal(p)
    mu = 0.0
    M = np.ascontiguousarray(A + mu * np.diag(diag ** 2))
    ref = np.zeros(p)
    assert L.gslnls_debug_mchol_solve(p, M.ctypes.data_as(_lib.DP), diag.ctypes.data_as(_lib.DP), mu, rhs.ctypes.data_as(_lib.DP),
                                      ref.ctypes.data_as(_lib.DP)) == 0
    dA = C.c_void_p()
    assert L.gslnls_debug_device_alloc(C.byref(dA), A.nbytes) == 0
    try:
        assert L.gslnls_debug_device_copy(dA, A.ctypes.data_as(C.c_void_p), A.nbytes, 1) == 0
        sol = np.zeros(p)
        assert L.gslnls_debug_mchol_solve_resident(p, dA, diag.ctypes.data_as(_lib.DP), mu, rhs.ctypes.data_as(_lib.DP),
                                                   sol.ctypes.data_as(_lib.DP)) == 0
    finally:
        L.gslnls_debug_device_free(dA)
    assert np.all(np.isfinite(sol)) and np.array_equal(sol, ref)
