"""The reference-side shim, EXECUTED: integration/r_shim/gslnls_hip_shim.c compiled against tests/r_mini/rmini.c -- a small
functional implementation of the slice of R's C API the shim uses on its `function`-model route (this image has no R,
SURVEY.md 0.4; until round 5 the shim was only type-checked) -- and called the way .Call(C_nls, ...) calls it: twelve SEXP
arguments in, the list of src/nls.c:632-812 out.  The closures are Python functions behind ctypes callbacks.

What it pins: the marshalling of the callback route for p > 64 (where a p-sized scratch of the shim was once left
uninitialised), the returned slots, names and dimnames, the trace text through Rprintf, start ranges
(gslnls_nls_fn_mstart) and that nothing falls through to C_nls.  The numbers are compared with the Python mirror's call of
the same core: bit for bit."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from test_gpu_function import gaussians

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def amd():
    import gslnls_amd
    from gslnls_amd import _lib
    assert _lib.lib().gslnls_device_count() >= 1, "no MI355X visible: the HIP path cannot be tested"
    return gslnls_amd


@pytest.fixture(scope="module")
def rshim(amd):
    out = os.path.join(ROOT, "tests", "r_mini", "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "rshim_test.so")
    cmd = ["gcc", "-std=gnu11", "-Wall", "-O1", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "tests", "r_stub"),
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "integration", "r_shim"),
           os.path.join(ROOT, "tests", "r_mini", "rmini.c"), os.path.join(ROOT, "integration", "r_shim", "gslnls_hip_shim.c"),
           os.path.join(ROOT, "integration", "r_shim", "gslnls_hip_large_shim.c"),
           "-o", so, "-L" + os.path.join(ROOT, "gslnls_amd"), "-lgslnls_hip", "-Wl,-rpath," + os.path.join(ROOT, "gslnls_amd")]
    subprocess.run(cmd, check=True, capture_output=True)
    L = C.CDLL(so)
    P = C.c_void_p
    for name, res, args in [("rm_nil", P, []), ("rm_real", P, [C.c_int, C.c_void_p]), ("rm_int", P, [C.c_int, C.c_void_p, C.c_int]),
                            ("rm_strings", P, [C.c_int, C.POINTER(C.c_char_p)]), ("rm_set_names", None, [P, P]),
                            ("rm_set_dim", None, [P, C.c_int, C.c_int, P, P]), ("rm_list", P, [C.c_int]),
                            ("rm_list_set", None, [P, C.c_int, P]), ("rm_list_get", P, [P, C.c_int]), ("rm_env", P, []),
                            ("rm_closure", P, [C.c_void_p, C.c_void_p, P]), ("rm_type", C.c_int, [P]), ("rm_length", C.c_int, [P]),
                            ("rm_real_ptr", C.POINTER(C.c_double), [P]), ("rm_int_ptr", C.POINTER(C.c_int), [P]),
                            ("rm_string", C.c_char_p, [P, C.c_int]), ("rm_names", P, [P]), ("rm_dimnames", P, [P]),
                            ("rm_nrow", C.c_int, [P]), ("rm_ncol", C.c_int, [P]), ("rm_warnings", C.c_char_p, []),
                            ("rm_printed", C.c_char_p, []), ("rm_fell_through", C.c_int, []), ("rm_reset", None, []),
                            ("rm_s4", P, [C.c_char_p, P]), ("rm_expr", P, [C.c_char_p, P]), ("rm_formula", P, [P, P]),
                            ("rm_env_set", None, [P, C.c_char_p, P]), ("C_nls_hip", P, [P] * 12), ("C_nls_large_hip", P, [P] * 9)]:
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    return L


CB = C.CFUNCTYPE(C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_void_p)


def _real(L, v):
    v = np.ascontiguousarray(v, dtype=np.float64)
    return L.rm_real(len(v), v.ctypes.data_as(C.c_void_p))


def _strs(L, names):
    arr = (C.c_char_p * len(names))(*[s.encode() for s in names])
    return L.rm_strings(len(names), arr)


def _vec(L, s):
    n = L.rm_length(s)
    return np.ctypeslib.as_array(L.rm_real_ptr(s), shape=(n,)).copy()


def _call(L, amd, fn, jac, y, start, names, trace=False, ranges=None, has=None, startisnum=True):
    """build the twelve arguments of .Call(C_nls, ...) (R/nls.R:716-720) and call the shim"""
    from gslnls_amd.control import gsl_nls_control, pack_control
    p, n = len(names), len(y)
    seen = {}

    def wrap(f, matrix):
        def cb(args, nargs, user):
            if L.rm_type(args[0]) == 19:  # a named list of scalars: start was a list (control_int[13] == 0, src/nls.c:163-171)
                seen["par_is_list"] = True
                th = np.array([L.rm_real_ptr(L.rm_list_get(args[0], k))[0] for k in range(p)])
            else:
                th = _vec(L, args[0])
            seen.setdefault("names", [L.rm_string(L.rm_names(args[0]), k).decode() for k in (0, p - 1)])
            v = np.asarray(f(th), dtype=np.float64)
            if matrix:
                s = _real(L, np.asfortranarray(v).reshape(-1, order="F"))
                L.rm_set_dim(s, v.shape[0], v.shape[1], L.rm_nil(), L.rm_nil())
                return s
            return _real(L, v)
        return CB(cb)
    keep = [wrap(fn, False), wrap(jac, True) if jac is not None else None]
    env = L.rm_env()
    fn_s = L.rm_closure(C.cast(keep[0], C.c_void_p), None, env)
    jac_s = L.rm_closure(C.cast(keep[1], C.c_void_p), None, env) if jac is not None else L.rm_nil()
    ci, cd = pack_control(gsl_nls_control(solver="cholesky"), "lm", trace, startisnum, False)
    nm = _strs(L, names)
    if ranges is None:
        st = _real(L, start)
        L.rm_set_names(st, nm)
        hs = L.rm_int(p, np.ones(p, dtype=np.int32).ctypes.data_as(C.c_void_p), 1)
    else:
        st = _real(L, np.asarray(ranges, dtype=np.float64).reshape(-1))  # 2 x p column-major: (lower, upper) per parameter
        L.rm_set_dim(st, 2, p, L.rm_nil(), nm)
        hsv = np.ascontiguousarray(has, dtype=np.int32)
        hs = L.rm_int(2 * p, hsv.ctypes.data_as(C.c_void_p), 1)
        L.rm_set_dim(hs, 2, p, L.rm_nil(), L.rm_nil())
    loss = L.rm_list(2)
    L.rm_list_set(loss, 0, L.rm_int(1, np.zeros(1, dtype=np.int32).ctypes.data_as(C.c_void_p), 0))
    L.rm_list_set(loss, 1, _real(L, [0.0]))
    L.rm_reset()
    ans = L.C_nls_hip(fn_s, _real(L, y), jac_s, L.rm_nil(), env, st, L.rm_nil(), L.rm_nil(),
                      L.rm_int(len(ci), ci.ctypes.data_as(C.c_void_p), 0), _real(L, cd), hs, loss)
    return ans, seen, keep


def _slots(L, ans):
    nm = L.rm_names(ans)
    return [L.rm_string(nm, k).decode() for k in range(L.rm_length(ans))]


def test_function_model_with_100_parameters_through_the_shim(amd, rshim):
    L = rshim
    x, y, model, jac, start, truth = gaussians(33, 3000, 4233)
    p = len(start)
    names = ["th%d" % (k + 1) for k in range(p)]
    ans, seen, keep = _call(L, amd, model, jac, y, start, names)
    assert L.rm_fell_through() == 0 and L.rm_warnings() == b""
    assert _slots(L, ans) == ["par", "covar", "resid", "grad", "niter", "status", "conv", "ssr", "ssrtol", "algorithm", "neval", "irls"]
    assert seen["names"] == ["th1", "th%d" % p]  # `par` reaches the closures as a named vector, every name in place
    par = L.rm_list_get(ans, 0)
    assert [L.rm_string(L.rm_names(par), k).decode() for k in (0, 64, 65, p - 1)] == ["th1", "th65", "th66", "th%d" % p]
    ref = amd.gsl_nls(model, y=y, start=start, jac=jac, control=dict(solver="cholesky"))
    assert np.array_equal(_vec(L, par), np.asarray(ref["par"]))
    assert L.rm_int_ptr(L.rm_list_get(ans, 4))[0] == ref["niter"] and L.rm_int_ptr(L.rm_list_get(ans, 6))[0] == 0
    assert L.rm_real_ptr(L.rm_list_get(ans, 7))[0] == ref["ssr"]
    assert L.rm_string(L.rm_list_get(ans, 5), 0).decode() == "success"
    cov, grad = L.rm_list_get(ans, 1), L.rm_list_get(ans, 3)
    assert (L.rm_nrow(cov), L.rm_ncol(cov), L.rm_nrow(grad), L.rm_ncol(grad)) == (p, p, len(y), p)
    assert np.array_equal(_vec(L, cov).reshape(p, p, order="F"), np.asarray(ref["covar"]))
    assert np.array_equal(_vec(L, L.rm_list_get(ans, 2)), np.asarray(ref["resid"]))
    dn = L.rm_dimnames(grad)
    assert L.rm_string(L.rm_list_get(dn, 1), p - 1).decode() == "th%d" % p
    ne = L.rm_list_get(ans, 10)
    assert [L.rm_int_ptr(ne)[k] for k in range(3)] == [ref["neval"]["f"], ref["neval"]["J"], ref["neval"]["fvv"]]


def test_trace_text_reaches_rprintf_and_wrong_results_become_warnings(amd, rshim):
    L = rshim
    x, y, model, jac, start, truth = gaussians(22, 2000, 4222)
    p = len(start)
    names = ["b%d" % k for k in range(p)]
    ans, seen, keep = _call(L, amd, model, jac, y, start, names, trace=True)
    assert _slots(L, ans)[-2:] == ["partrace", "ssrtrace"]
    out = L.rm_printed().decode()
    assert out.startswith("iter   1: ssr = ") and "summary from method 'multifit/levenberg-marquardt'" in out
    niter = L.rm_int_ptr(L.rm_list_get(ans, 4))[0]
    assert out.count("\niter ") + 1 == niter  # (one line per iteration, as the reference's callback prints them)
    # a closure that returns the wrong length: EBADFUNC, NA-filled result, the reference's warning -- raised after the core returned
    ans, seen, keep = _call(L, amd, lambda th: model(th)[:-1], None, y, start, names)
    assert b"does not return numeric vector of expected length n" in L.rm_warnings()
    assert L.rm_int_ptr(L.rm_list_get(ans, 6))[0] != 0
    assert np.all(np.isnan(_vec(L, L.rm_list_get(ans, 2)))) and L.rm_fell_through() == 0


def test_start_ranges_of_a_function_model_through_the_shim(amd, rshim):
    L = rshim
    from test_gpu_function_mstart import madsen  # noqa: F401  (the reference's unit test 4.2.x model)
    fn, jac = madsen()
    y = np.zeros(3)
    names = ["x1", "x2"]
    ranges = np.array([[-1.0, 1.0], [-1.0, 1.0]])  # (lower, upper) per parameter
    ans, seen, keep = _call(L, amd, fn, jac, y, None, names, ranges=ranges.reshape(-1), has=np.ones(4, dtype=np.int32))
    assert L.rm_fell_through() == 0 and L.rm_int_ptr(L.rm_list_get(ans, 6))[0] == 0
    ref = amd.gsl_nls(fn, y=y, start={"x1": [-1.0, 1.0], "x2": [-1.0, 1.0]}, jac=jac, control=dict(solver="cholesky"))
    assert np.array_equal(_vec(L, L.rm_list_get(ans, 0)), np.asarray(ref["par"]))


def test_par_as_a_list_of_scalars_when_start_was_a_list(amd, rshim):
    """start given as a list: the reference hands the closures a named LIST of scalars (src/nls.c:163-171, control_int[13] == 0)"""
    L = rshim
    x, y, model, jac, start, truth = gaussians(2, 300, 4202)
    names = ["p%d" % k for k in range(len(start))]
    ans, seen, keep = _call(L, amd, model, jac, y, start, names, startisnum=False)
    assert seen.get("par_is_list") and seen["names"] == ["p0", "p%d" % (len(start) - 1)]
    ref = amd.gsl_nls(model, y=y, start=start, jac=jac, control=dict(solver="cholesky"))
    assert L.rm_int_ptr(L.rm_list_get(ans, 6))[0] == 0 and np.array_equal(_vec(L, L.rm_list_get(ans, 0)), np.asarray(ref["par"]))


def _int(L, v, logical=0):
    v = np.ascontiguousarray(v, dtype=np.int32)
    return L.rm_int(len(v), v.ctypes.data_as(C.c_void_p), logical)


@pytest.mark.parametrize("kind,alg", [("dgCMatrix", "cgst"), ("dgCMatrix", "lm"), ("dgRMatrix", "cgst"), ("matrix", "cgst")])
def test_gsl_nls_large_through_the_large_shim(amd, rshim, kind, alg):
    """.Call(C_nls_large, ...) (src/nls_large.c:66-75) on the reference's own sparse example (README.md:1040-1146, penalty function I,
    p = 500): the Jacobian closure returns a Matrix-package object -- its slots read in place by the shim -- or a base matrix;
    the answer is the list of src/nls_large.c:275-416, `grad` = as.matrix() of the last Jacobian, numbers bit for bit the
    mirror's call of the same core."""
    import scipy.sparse as sp
    from gslnls_amd.control import gsl_nls_control
    from gslnls_amd.nls_large import pack_control_large
    L = rshim
    p = 500
    a = np.sqrt(1e-5)
    J0 = sp.vstack([sp.identity(p, format="csr") * a, sp.csr_matrix(np.ones((1, p)))])
    J0 = (J0.tocsr() if kind == "dgRMatrix" else J0.tocsc())
    J0.sort_indices()
    last = np.flatnonzero(J0.indices == p) if kind != "dgRMatrix" else np.arange(J0.indptr[p], J0.indptr[p + 1])
    names = ["x%d" % (k + 1) for k in range(p)]
    nm = _strs(L, names)
    seen = {}

    def fn(th):
        return np.concatenate([a * (th - 1.0), [np.sum(th ** 2) - 0.25]])

    def jac_py(th):
        J0.data[last] = 2.0 * th
        return J0

    def f_cb(args, nargs, user):
        seen.setdefault("names", [L.rm_string(L.rm_names(args[0]), k).decode() for k in (0, p - 1)])
        return _real(L, fn(_vec(L, args[0])))

    def j_cb(args, nargs, user):
        J = jac_py(_vec(L, args[0]))
        if kind == "matrix":
            D = np.asfortranarray(J.toarray())
            s = _real(L, D.reshape(-1, order="F"))
            L.rm_set_dim(s, p + 1, p, L.rm_nil(), L.rm_nil())
            return s
        slots = L.rm_list(4)
        L.rm_list_set(slots, 0, _int(L, J.indices))
        L.rm_list_set(slots, 1, _int(L, J.indptr))
        L.rm_list_set(slots, 2, _real(L, J.data))
        L.rm_list_set(slots, 3, _int(L, [p + 1, p]))
        L.rm_set_names(slots, _strs(L, ["j" if kind == "dgRMatrix" else "i", "p", "x", "Dim"]))
        return L.rm_s4(kind.encode(), slots)
    keep = [CB(f_cb), CB(j_cb)]
    env = L.rm_env()
    fn_s = L.rm_closure(C.cast(keep[0], C.c_void_p), None, env)
    jac_s = L.rm_closure(C.cast(keep[1], C.c_void_p), None, env)
    ci, cd = pack_control_large(gsl_nls_control(maxiter=500), alg, False)
    st = _real(L, np.arange(1.0, p + 1))
    L.rm_set_names(st, nm)
    L.rm_reset()
    ans = L.C_nls_large_hip(fn_s, _real(L, np.zeros(p + 1)), jac_s, L.rm_nil(), env, st, L.rm_nil(), _int(L, ci), _real(L, cd))
    assert L.rm_fell_through() == 0 and seen["names"] == ["x1", "x%d" % p]
    assert _slots(L, ans) == ["par", "covar", "resid", "grad", "niter", "status", "conv", "ssr", "ssrtol", "algorithm", "neval"]
    ref = amd.gsl_nls_large(fn, y=np.zeros(p + 1), start=np.arange(1.0, p + 1), algorithm=alg,
                            jac=(lambda th: np.asfortranarray(jac_py(th).toarray())) if kind == "matrix" else jac_py,
                            control=dict(maxiter=500))
    par = _vec(L, L.rm_list_get(ans, 0))
    assert L.rm_int_ptr(L.rm_list_get(ans, 6))[0] == 0 and np.array_equal(par, np.asarray(ref["par"]))
    assert L.rm_int_ptr(L.rm_list_get(ans, 4))[0] == ref["niter"] and abs(L.rm_real_ptr(L.rm_list_get(ans, 7))[0] - 0.004778845) < 5e-10
    assert L.rm_string(L.rm_list_get(ans, 9), 0).decode() == ref["algorithm"]
    grad = L.rm_list_get(ans, 3)
    assert (L.rm_nrow(grad), L.rm_ncol(grad)) == (p + 1, p)
    assert np.array_equal(_vec(L, grad).reshape(p + 1, p, order="F"), jac_py(par).toarray())
    assert L.rm_string(L.rm_list_get(L.rm_dimnames(grad), 1), p - 1).decode() == "x%d" % p
    ne = L.rm_list_get(ans, 10)
    assert [L.rm_string(L.rm_names(ne), k).decode() for k in range(4)] == ["f", "dfu", "df2", "fvv"]
    assert [L.rm_int_ptr(ne)[k] for k in range(3)] == [ref["neval"]["f"], ref["neval"]["dfu"], ref["neval"]["df2"]]
    cov = _vec(L, L.rm_list_get(ans, 1)).reshape(p, p, order="F")
    assert np.array_equal(cov, np.asarray(ref["covar"]))


def _formula_env(L, rhs, data):
    """the frame gsl_nls.formula evaluates .fn in (R/nls.R:565): it binds `formula` and the model frame `mf`"""
    from gslnls_amd import formula as F
    env = L.rm_env()
    vars_ = F.symbols(F.parse_expr(rhs))
    L.rm_env_set(env, b"formula", L.rm_formula(L.rm_nil(), L.rm_expr(rhs.encode(), _strs(L, vars_))))
    mf = L.rm_list(len(data))
    for k, (name, col) in enumerate(data.items()):
        L.rm_list_set(mf, k, _real(L, col))
    L.rm_set_names(mf, _strs(L, list(data)))
    L.rm_env_set(env, b"mf", mf)
    return env


@pytest.mark.parametrize("rhs,start", [
    ("A * exp(-lam * x) + b", {"b": 0.0, "A": 1.0, "lam": 1.0}),         # a hand-written device model, parameters in another order
    ("a * exp(-b * x) + c * sin(d * x)", {"a": 4.0, "b": 1.0, "c": 0.4, "d": 2.9}),  # the expression itself (GSLNLS_MODEL_EXPR)
])
def test_formula_route_of_the_shim(amd, rshim, rhs, start):
    """gsl_nls(y ~ f(x, theta), data, start): the shim finds `formula` and `mf` in the closure's frame, asks R for the text and the
    symbols of the right-hand side (here: rmini's stand-ins for deparse1 / all.vars), lowers the formula, takes the data
    columns out of the model frame, permutes start into the device model's parameter order and the result back -- no closure is
    evaluated.  Bit for bit the mirror's gsl_nls(formula)."""
    from gslnls_amd.control import gsl_nls_control, pack_control
    L = rshim
    n = 400
    rng = np.random.default_rng(5)
    x = np.linspace(0.0, 3.0, n)
    if "sin" in rhs:
        y = 5.0 * np.exp(-1.5 * x) + 0.5 * np.sin(3.0 * x) + 0.01 * rng.standard_normal(n)
    else:
        y = 5.0 * np.exp(-1.5 * x) + 1.0 + 0.01 * rng.standard_normal(n)
    names = list(start)
    p = len(names)
    env = _formula_env(L, rhs, {"y": y, "x": x})
    called = []
    cb = CB(lambda args, nargs, user: called.append(1) or None)
    fn_s = L.rm_closure(C.cast(cb, C.c_void_p), None, env)
    ci, cd = pack_control(gsl_nls_control(), "lm", False, True, False)
    st = _real(L, list(start.values()))
    L.rm_set_names(st, _strs(L, names))
    loss = L.rm_list(2)
    L.rm_list_set(loss, 0, _int(L, [0]))
    L.rm_list_set(loss, 1, _real(L, [0.0]))
    L.rm_reset()
    ans = L.C_nls_hip(fn_s, _real(L, y), L.rm_nil(), L.rm_nil(), env, st, L.rm_nil(), L.rm_nil(), _int(L, ci), _real(L, cd),
                      _int(L, np.ones(p), 1), loss)
    assert L.rm_fell_through() == 0 and not called and L.rm_warnings() == b""
    ref = amd.gsl_nls("y ~ " + rhs, data=dict(x=x, y=y), start=start)
    par = L.rm_list_get(ans, 0)
    assert [L.rm_string(L.rm_names(par), k).decode() for k in range(p)] == names
    assert L.rm_int_ptr(L.rm_list_get(ans, 6))[0] == 0 and L.rm_int_ptr(L.rm_list_get(ans, 4))[0] == ref["niter"]
    assert np.array_equal(_vec(L, par), np.asarray(ref["par"]))
    assert np.array_equal(_vec(L, L.rm_list_get(ans, 1)).reshape(p, p, order="F"), np.asarray(ref["covar"]))
    assert np.array_equal(_vec(L, L.rm_list_get(ans, 3)).reshape(n, p, order="F"), np.asarray(ref["grad"]))
    assert np.array_equal(_vec(L, L.rm_list_get(ans, 2)), np.asarray(ref["resid"]))
