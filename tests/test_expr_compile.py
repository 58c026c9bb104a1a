"""Expression lowering (csrc/expr_compile.hpp + vm_program.hpp), run on the host through hostsim:
every formula of the reference's NIST test list (inst/unit_tests/unit_tests_gslnls.R via
tests/golden/nist_formula_problems.json) must compile, evaluate like the numpy evaluator of the same
formula and carry a gradient that agrees with a Richardson-extrapolated numeric derivative."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import pytest

from gslnls_amd import formula as F
import hostsim_py as hs

HERE = os.path.dirname(os.path.abspath(__file__))
PROBLEMS = json.load(open(os.path.join(HERE, "golden", "nist_formula_problems.json")))


def _split(pb):
    lhs, rhs = F.parse_formula(pb["formula"])
    names = list(pb["start"].keys())
    xnames = [v for v in F.symbols(rhs) if v not in names]
    rhs_text = pb["formula"].split("~", 1)[1].strip()
    return rhs, rhs_text, names, xnames


@pytest.mark.parametrize("pb", PROBLEMS, ids=[p["name"] for p in PROBLEMS])
def test_program_value_and_gradient(pb):
    rhs, rhs_text, names, xnames = _split(pb)
    data = {k: np.asarray(v, dtype=np.float64) for k, v in pb["data"].items()}
    X = np.stack([data[c] for c in xnames], axis=1)
    th = np.array([pb["target"][k] for k in names]) * (1.0 + 0.01 * np.arange(1, len(names) + 1))
    val, grad, st = hs.expr_eval(rhs_text, names, xnames, th, X)
    assert 0 < st["nvalue"] <= st["nops"] <= 256

    def ev(t):
        env = dict(data)
        env.update({k: t[i] for i, k in enumerate(names)})
        return np.asarray(F.evaluate(rhs, env), dtype=np.float64) * np.ones(len(X))

    ref = ev(th)
    np.testing.assert_allclose(val, ref, rtol=1e-13, atol=1e-13 * np.max(np.abs(ref)))
    if pb["name"] == "Lubricant":
        # values reach 1e10 (x2^3 terms), so differences of the value lose the small partials: use the
        # hand-derived gradient instead
        b = th
        x1, x2 = data["x1"], data["x2"]
        q = b[7] + b[8] * x2 ** 2
        E = np.exp(-x1 / q)
        A = b[5] * x2 + b[6] * x2 ** 3
        hand = np.stack([1 / (b[1] + x1), -b[0] / (b[1] + x1) ** 2, x2, x2 ** 2, x2 ** 3, x2 * E, x2 ** 3 * E,
                         A * E * x1 / q ** 2, A * E * x1 * x2 ** 2 / q ** 2], axis=1)
        np.testing.assert_allclose(grad, hand, rtol=1e-12, atol=0)
        return
    # Richardson-extrapolated central differences: O(h^4) truncation, ~1e-9 relative accuracy
    for k in range(len(names)):
        h = 1e-4 * max(abs(th[k]), 1e-8)
        e = np.zeros(len(names))
        e[k] = h
        d1 = (ev(th + e) - ev(th - e)) / (2 * h)
        d2 = (ev(th + e / 2) - ev(th - e / 2)) / h
        num = (4 * d2 - d1) / 3
        scale = np.max(np.abs(num)) + 1e-300
        assert np.max(np.abs(grad[:, k] - num)) / scale < 1e-6, (pb["name"], names[k])


def test_constant_folding_and_sharing():
    # exp(-b*x) appears in value and both partials: must be computed once
    _, _, st = hs.expr_eval("a*exp(-b*x)", ["a", "b"], ["x"], [1.0, 2.0], np.linspace(0, 1, 5))
    assert st["nops"] <= 7
    v, g, _ = hs.expr_eval("2^3 + 0*a + x", ["a"], ["x"], [5.0], np.array([1.0, 2.0]))
    np.testing.assert_allclose(v, [9.0, 10.0])
    np.testing.assert_allclose(g[:, 0], 0.0)


def test_rejects_unknown_symbols_and_functions():
    with pytest.raises(ValueError):
        hs.expr_eval("a*foo(x)", ["a"], ["x"], [1.0], np.array([1.0]))
    with pytest.raises(ValueError):
        hs.expr_eval("a*z", ["a"], ["x"], [1.0], np.array([1.0]))


def test_native_lowering_builds_without_a_device(tmp_path, monkeypatch):
    """gslnls_expr_build: expression -> generated C++ row model -> hiprtc IN PROCESS (the kernel templates travel inside
    the library as text) -> cached code object; a second call is a cache hit.  No GPU, no hipcc, no source tree involved
    (the in-process compiler targets gfx950 whatever the host)."""
    import ctypes as C
    import time
    from gslnls_amd import _lib
    monkeypatch.setenv("GSLNLS_JIT_CACHE", str(tmp_path))
    monkeypatch.setenv("GSLNLS_HIPCC", "/nonexistent")          # nothing may look for a compiler driver
    monkeypatch.setenv("PATH", "/nonexistent")
    m = _lib.Model(_lib.MODEL_EXPR, 3, 1, None, 0)
    keep = _lib.set_expr(m, "b1*(1 - 1/exp(b2*x)) + b3^2", ["b1", "b2", "b3"], ["x"], "jit")  # noqa: F841
    assert _lib.lib().gslnls_expr_native_state(C.byref(m), 1) == 0      # nobody asked yet, cache empty
    buf = C.create_string_buffer(512)
    t0 = time.time()
    assert _lib.lib().gslnls_expr_build(C.byref(m), buf, 512) == 0
    t1 = time.time()
    assert t1 - t0 < 30.0                                              # two units (analytic + forward), ~1-2 s each
    path = buf.value.decode()
    assert path.startswith(str(tmp_path)) and os.path.exists(path)
    blob = open(path, "rb").read()
    assert blob.startswith(b"GSLRTC1\n") and b"lm_step_kernel" in blob and b"\x7fELF" in blob   # names + a code object
    assert _lib.lib().gslnls_expr_native_state(C.byref(m), 1) == 2 and _lib.lib().gslnls_expr_native_state(C.byref(m), 0) == 2
    assert len([f for f in os.listdir(tmp_path) if f.endswith(".bin")]) == 2
    # a different formula is a different unit
    m2 = _lib.Model(_lib.MODEL_EXPR, 3, 1, None, 0)
    keep2 = _lib.set_expr(m2, "b1*(1 - 1/exp(b2*x)) + b3^3", ["b1", "b2", "b3"], ["x"], "jit")  # noqa: F841
    assert _lib.lib().gslnls_expr_native_state(C.byref(m2), 1) == 0
    # an expression that does not parse is refused, not built
    bad = _lib.Model(_lib.MODEL_EXPR, 1, 1, None, 0)
    keep3 = _lib.set_expr(bad, "a*foo(x)", ["a"], ["x"], "jit")  # noqa: F841
    assert _lib.lib().gslnls_expr_build(C.byref(bad), buf, 512) == _lib.E_UNSUPPORTED


def test_without_the_in_process_compiler_native_lowering_is_refused_not_faked(tmp_path):
    """a host whose HIP runtime lacks hiprtc (simulated: GSLNLS_HIPRTC=none): building native code fails with
    GSLNLS_E_UNSUPPORTED and a message, nothing is cached; a background request reports failure instead of hanging"""
    import subprocess
    import sys
    code = r"""
import ctypes as C, sys
sys.path.insert(0, %r)
from gslnls_amd import _lib
L = _lib.lib()
m = _lib.Model(_lib.MODEL_EXPR, 3, 1, None, 0)
keep = _lib.set_expr(m, "a*exp(-b*x) + c", ["a", "b", "c"], ["x"], "jit")
buf = C.create_string_buffer(256)
print("build", L.gslnls_expr_build(C.byref(m), buf, 256), "prefetch", L.gslnls_expr_prefetch(C.byref(m), 0))
names = ["t%%d" %% k for k in range(12)]
w = _lib.Model(_lib.MODEL_EXPR, 12, 1, None, 0)
keep2 = _lib.set_expr(w, " + ".join("%%s*x^%%d" %% (nm, k) for k, nm in enumerate(names)), names, ["x"], "jit")
print("wide", L.gslnls_expr_build(C.byref(w), buf, 256))
""" % (ROOT,)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120,
                         env=dict(os.environ, GSLNLS_JIT_CACHE=str(tmp_path), GSLNLS_HIPRTC="none"))
    assert out.returncode == 0, out.stderr[-1500:]
    assert "build -101 prefetch -1" in out.stdout and "wide -101" in out.stdout
    assert "hiprtc disabled" in out.stderr
    assert not [f for f in os.listdir(tmp_path) if f.endswith(".bin")]


def test_process_exit_with_queued_background_builds_is_clean(tmp_path):
    """GSLNLS_LOWER_AUTO starts builds on a background thread; a process that exits while dozens are queued and one is
    inside the compiler must neither crash nor hang (seen before gslnls_shutdown existed: "LLVM ERROR", heap
    corruption, or a hang, as LLVM's static objects were destroyed under the compiling thread).  The Python mirror
    registers gslnls_shutdown with atexit; queued builds become no-ops, the one in flight is waited for."""
    import subprocess
    import sys
    code = r"""
import ctypes as C, sys, os, time
sys.path.insert(0, %r)
from gslnls_amd import _lib
L = _lib.lib()
keep = []
for k in range(40):
    m = _lib.Model(_lib.MODEL_EXPR, 3, 1, None, 0)
    keep.append((m, _lib.set_expr(m, "a*exp(-b*x) + c*%%d" %% (k + 2), ["a", "b", "c"], ["x"], "auto")))
    st = L.gslnls_expr_prefetch(C.byref(m), 1)
assert st in (1, 2), st       # queued -- or found in the cache: the newest request is built first, the first run left it there
time.sleep(float(sys.argv[1]))
print("leaving", flush=True)
""" % (ROOT,)
    for delay in ("0.05", "0.6"):
        out = subprocess.run([sys.executable, "-c", code, delay], capture_output=True, text=True, timeout=120,
                             env=dict(os.environ, GSLNLS_JIT_CACHE=str(tmp_path)))
        assert out.returncode == 0, (delay, out.returncode, out.stderr[-1500:])
        assert "leaving" in out.stdout and "LLVM ERROR" not in out.stderr
    # at most the builds that got through before the exit are in the cache -- and every file there is whole
    from gslnls_amd import _lib  # noqa: F401
    for f in os.listdir(tmp_path):
        if f.endswith(".bin"):
            assert open(os.path.join(tmp_path, f), "rb").read(8) == b"GSLRTC1\n"


def test_a_waiting_build_takes_a_queued_unit_over_from_the_background_compiler(tmp_path):
    """ONE background compiler thread serves GSLNLS_LOWER_AUTO through a queue; a caller that needs a unit NOW
    (gslnls_expr_build, lowering "jit") does not wait behind the queue: it compiles the unit itself and the background
    thread skips it."""
    import subprocess
    import sys
    code = r"""
import ctypes as C, sys, os, time
sys.path.insert(0, %r)
from gslnls_amd import _lib
L = _lib.lib()
keep = []
for k in range(24):
    m = _lib.Model(_lib.MODEL_EXPR, 3, 1, None, 0)
    keep.append((m, _lib.set_expr(m, "a*exp(-b*x) + c*%%d" %% (k + 2), ["a", "b", "c"], ["x"], "auto")))
    assert L.gslnls_expr_prefetch(C.byref(m), 1) == 1
    assert L.gslnls_expr_prefetch(C.byref(m), 1) == 1       # asking again queues nothing new
last = keep[-1][0]
buf = C.create_string_buffer(512)
t0 = time.time()
assert L.gslnls_expr_build(C.byref(last), buf, 512) == 0
dt = time.time() - t0
assert L.gslnls_expr_native_state(C.byref(last), 1) == 2 and L.gslnls_expr_native_state(C.byref(last), 0) == 2
nthreads = len(os.listdir("/proc/self/task"))
print("took %%.2f threads %%d" %% (dt, nthreads), flush=True)
""" % (ROOT,)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=240,
                         env=dict(os.environ, GSLNLS_JIT_CACHE=str(tmp_path)))
    assert out.returncode == 0, (out.returncode, out.stderr[-1500:])
    took, nthreads = out.stdout.split()[1], int(out.stdout.split()[3])
    assert float(took) < 20.0          # two units (~1-2 s each) plus at most the one the background thread was inside
    assert nthreads < 12               # not a thread per request (hiprtc/comgr keep a few of their own)


@pytest.mark.parametrize("pb", PROBLEMS, ids=[p["name"] for p in PROBLEMS])
def test_second_directional_derivative(pb):
    """fvv = TRUE on a formula: D^2 f[v, v] of the compiled program (third closure) against a central second
    difference of the numpy evaluation along v (the reference: stats::deriv(hessian = TRUE), R/nls.R:600-640)"""
    rhs, rhs_text, names, xnames = _split(pb)
    data = {k: np.asarray(v, dtype=np.float64) for k, v in pb["data"].items()}
    X = np.stack([data[c] for c in xnames], axis=1)
    p = len(names)
    th = np.array([pb["target"][k] for k in names]) * (1.0 + 0.01 * np.arange(1, p + 1))
    v = th * 0.01 * np.cos(np.arange(1, p + 1))
    val, grad, st = hs.expr_eval(rhs_text, names, xnames, th, X, direction=v)
    if st["nfvv"] == 0:
        pytest.skip("second derivatives of this formula exceed the program size: fvv falls back to finite differences")
    assert st["nops"] <= st["nfvv"] <= 256

    def ev(t):
        env = dict(data)
        env.update({k: t[i] for i, k in enumerate(names)})
        return np.asarray(F.evaluate(rhs, env), dtype=np.float64) * np.ones(len(X))
    if pb["name"] == "Lubricant":
        pytest.skip("values up to 1e10 swamp second differences (see the gradient test)")
    # Richardson-extrapolated central second difference along v
    d2 = lambda h: (ev(th + h * v) - 2 * ev(th) + ev(th - h * v)) / (h * h)  # noqa: E731
    num = (4 * d2(0.5e-2) - d2(1e-2)) / 3
    scale = np.max(np.abs(num)) + 1e-300
    assert np.max(np.abs(st["fvv"] - num)) / scale < 2e-5, pb["name"]


VOCABULARY = [
    ("a * sinh(b * x) + c", dict(a=1.3, b=0.7, c=0.2), (-2.0, 2.0)),
    ("a * cosh(b * x) - c * x", dict(a=0.8, b=1.1, c=0.3), (-2.0, 2.0)),
    ("a * asin(b * x) + c", dict(a=2.0, b=0.45, c=0.1), (-1.9, 1.9)),
    ("a * acos(b * x) + c * x", dict(a=1.5, b=0.4, c=0.2), (-1.9, 1.9)),
    ("a * log1p(b * x * x) + c", dict(a=1.2, b=0.9, c=-0.4), (-2.0, 2.0)),
    ("a * expm1(-b * x) + c", dict(a=2.5, b=0.6, c=0.3), (0.0, 3.0)),
    ("a * log2(b + x) + c * log10(b + x * x)", dict(a=1.1, b=2.5, c=0.7), (0.0, 4.0)),
    ("a * pnorm((x - m) / s) + c", dict(a=3.0, m=0.4, s=1.3, c=0.5), (-3.0, 3.0)),
    ("a * dnorm((x - m) / s) / s", dict(a=3.0, m=-0.3, s=0.8), (-3.0, 3.0)),
    ("a * sinpi(b * x) + c * cospi(x)", dict(a=1.0, b=0.35, c=0.5), (-1.0, 1.0)),
]


@pytest.mark.parametrize("rhs_text,pars,xr", VOCABULARY, ids=[v[0].split("(")[0].split()[-1] for v in VOCABULARY])
def test_the_rest_of_stats_derivs_table(rhs_text, pars, xr):
    """The functions stats::deriv differentiates beyond the round-3 list (R/nls.R:588-599 builds the Jacobian with it):
    sinh cosh asin acos log1p expm1 log2 log10 pnorm dnorm sinpi cospi tanpi.  Program value against numpy (1e-13),
    symbolic gradient against Richardson-extrapolated differences, second directional derivative against differences of
    the gradient."""
    names = list(pars)
    x = np.linspace(xr[0], xr[1], 41)
    th = np.array([pars[k] for k in names])
    val, grad, st = hs.expr_eval(rhs_text, names, ["x"], th, x)
    rhs = F.parse_expr(rhs_text)

    def ev(t):
        env = {"x": x}
        env.update({k: t[i] for i, k in enumerate(names)})
        return np.asarray(F.evaluate(rhs, env), dtype=np.float64) * np.ones(len(x))

    ref = ev(th)
    np.testing.assert_allclose(val, ref, rtol=1e-13, atol=1e-13 * np.max(np.abs(ref)))
    for k in range(len(names)):
        h = 1e-4 * max(abs(th[k]), 1e-8)
        e = np.zeros(len(names))
        e[k] = h
        d1 = (ev(th + e) - ev(th - e)) / (2 * h)
        d2 = (ev(th + e / 2) - ev(th - e / 2)) / h
        num = (4 * d2 - d1) / 3
        scale = np.max(np.abs(num)) + 1e-300
        assert np.max(np.abs(grad[:, k] - num)) / scale < 1e-6, (rhs_text, names[k])


SELFSTART = [
    ("SSasymp(x, Asym, R0, lrc)", dict(Asym=1.2, R0=5.8, lrc=0.1), (0.0, 3.0)),
    ("SSasympOff(x, Asym, lrc, c0)", dict(Asym=4.0, lrc=-0.3, c0=0.4), (0.5, 4.0)),
    ("SSasympOrig(x, Asym, lrc)", dict(Asym=4.0, lrc=-0.3), (0.0, 4.0)),
    ("SSbiexp(x, A1, lrc1, A2, lrc2)", dict(A1=3.0, lrc1=0.8, A2=1.0, lrc2=-1.2), (0.0, 4.0)),
    ("SSfol(4.0, x, lKe, lKa, lCl)", dict(lKe=-2.5, lKa=0.4, lCl=-3.0), (0.2, 10.0)),
    ("SSfpl(x, A, B, xmid, scal)", dict(A=0.5, B=4.0, xmid=2.0, scal=0.6), (0.0, 4.0)),
    ("SSgompertz(x, Asym, b2, b3)", dict(Asym=5.0, b2=2.2, b3=0.6), (0.0, 4.0)),
    ("SSlogis(x, Asym, xmid, scal)", dict(Asym=4.0, xmid=2.0, scal=0.5), (0.0, 4.0)),
    ("SSmicmen(x, Vm, K)", dict(Vm=200.0, K=0.06), (0.02, 1.1)),
    ("SSweibull(x, Asym, Drop, lrc, pwr)", dict(Asym=160.0, Drop=110.0, lrc=-1.5, pwr=2.2), (0.5, 4.0)),
]


@pytest.mark.parametrize("rhs_text,pars,xr", SELFSTART, ids=[v[0].split("(")[0] for v in SELFSTART])
def test_selfstart_models_by_their_closed_forms(rhs_text, pars, xr):
    """y ~ SSasymp(x, Asym, R0, lrc) and the other standard selfStart models (the reference's unit tests 6.x,
    inst/unit_tests/unit_tests_gslnls.R:267-293) are lowered by their closed forms: value against the documented formula
    evaluated by numpy, symbolic gradient against Richardson-extrapolated differences"""
    names = list(pars)
    x = np.linspace(xr[0], xr[1], 33)
    th = np.array([pars[k] for k in names])
    val, grad, st = hs.expr_eval(rhs_text, names, ["x"], th, x)
    rhs = F.parse_expr(rhs_text)

    def ev(t):
        env = {"x": x}
        env.update({k: t[i] for i, k in enumerate(names)})
        return np.asarray(F.evaluate(rhs, env), dtype=np.float64) * np.ones(len(x))

    ref = ev(th)
    np.testing.assert_allclose(val, ref, rtol=1e-12, atol=1e-13 * np.max(np.abs(ref)))
    for k in range(len(names)):
        h = 1e-4 * max(abs(th[k]), 1e-8)
        e = np.zeros(len(names))
        e[k] = h
        num = (4 * (ev(th + e / 2) - ev(th - e / 2)) / h - (ev(th + e) - ev(th - e)) / (2 * h)) / 3
        scale = np.max(np.abs(num)) + 1e-300
        assert np.max(np.abs(grad[:, k] - num)) / scale < 1e-6, (rhs_text, names[k])


# ---- the gamma family of stats::deriv's table (round 5): gamma, lgamma, digamma, trigamma, psigamma, factorial, lfactorial ----
GAMMA_CASES = [
    ("a * gamma(b * x)", ["a", "b"], [1.3, 0.7]),
    ("a * lgamma(x + b)", ["a", "b"], [2.0, 0.4]),
    ("a * digamma(b + x) + trigamma(x * b)", ["a", "b"], [0.5, 1.1]),
    ("psigamma(a * x, 2) + b * psigamma(x, 1)", ["a", "b"], [0.9, 2.0]),
    ("a * factorial(x / b) - lfactorial(x * b)", ["a", "b"], [1.0, 1.7]),
    ("exp(lgamma(a + x) - lgamma(a) - lgamma(x + 1)) * b^x", ["a", "b"], [2.5, 0.3]),  # negative-binomial kernel
]


@pytest.mark.parametrize("rhs_text,names,theta", GAMMA_CASES, ids=[c[0] for c in GAMMA_CASES])
def test_gamma_family_value_gradient_and_second_directional_derivative(rhs_text, names, theta):
    """device arithmetic of the family (devmath.hpp: gpsigamma by recurrence + asymptotic series) against scipy.special through
    the mirror's evaluator; the symbolic gradient (gamma' = gamma digamma, lgamma' = digamma, psigamma(., n)' = psigamma(., n + 1))
    and D^2 f[v, v] against Richardson-extrapolated differences of the scipy evaluation"""
    rhs = F.parse_expr(rhs_text)
    x = np.linspace(0.6, 6.0, 23)
    th = np.array(theta)
    v = np.array([0.7, -0.4])
    val, grad, st = hs.expr_eval(rhs_text, names, ["x"], th, x[:, None], direction=v)

    def ev(t):
        env = {"x": x}
        env.update(dict(zip(names, t)))
        return np.asarray(F.evaluate(rhs, env), dtype=np.float64) * np.ones(len(x))
    ref = ev(th)
    np.testing.assert_allclose(val, ref, rtol=2e-13, atol=1e-14)
    for k in range(len(names)):
        h = 1e-3 * max(abs(th[k]), 1e-8)
        e = np.zeros(len(names))
        e[k] = h
        num = (4 * (ev(th + e / 2) - ev(th - e / 2)) / h - (ev(th + e) - ev(th - e)) / (2 * h)) / 3
        assert np.max(np.abs(grad[:, k] - num)) / (np.max(np.abs(num)) + 1e-300) < 1e-7, (rhs_text, names[k])
    if st.get("fvv") is not None:
        h = 1e-3
        num2 = (-ev(th + 2 * h * v) + 16 * ev(th + h * v) - 30 * ref + 16 * ev(th - h * v) - ev(th - 2 * h * v)) / (12 * h * h)
        assert np.max(np.abs(st["fvv"] - num2)) / (np.max(np.abs(num2)) + 1e-300) < 1e-5, rhs_text


def test_psigamma_reflection_and_limits_of_the_vocabulary():
    # negative non-integer arguments go through the reflection formula
    x = np.array([-0.5, -1.3, -2.7, 0.2, 0.45])
    for n in range(5):
        val, _, _ = hs.expr_eval("psigamma(x + a, %d)" % n, ["a"], ["x"], [0.0], x[:, None])
        from scipy.special import digamma, polygamma
        m = 4
        ref = (digamma(x + m) if n == 0 else polygamma(n, x + m)) - sum(
            (-1.0) ** n * [1, 1, 2, 6, 24][n] / (x + k) ** (n + 1) for k in range(m))
        np.testing.assert_allclose(val, ref, rtol=1e-12)
    # the order has to be a literal 0..4; the fifth derivative is outside the table
    with pytest.raises(ValueError):
        hs.expr_eval("psigamma(x, a)", ["a"], ["x"], [1.0], np.array([[1.0]]))
    with pytest.raises(ValueError):
        hs.expr_eval("psigamma(x * a, 5)", ["a"], ["x"], [1.0], np.array([[1.0]]))
    # ... and a function outside the table is refused (the bindings then evaluate the formula's own closure instead)
    with pytest.raises(ValueError):
        hs.expr_eval("ifelse(x, a, 1)", ["a"], ["x"], [1.0], np.array([[1.0]]))


def test_mirror_evaluator_knows_comparisons_ifelse_pmax():
    rhs = F.parse_expr("ifelse(x < c0, a + b * x, a + b * c0) + pmax(0, x - 2) * d")
    env = dict(x=np.array([0.0, 1.0, 2.0, 3.0]), c0=1.5, a=1.0, b=2.0, d=10.0)
    np.testing.assert_allclose(F.evaluate(rhs, env), [1.0, 3.0, 4.0, 14.0])
    assert F.symbols(rhs) == ["x", "c0", "a", "b", "d"]
