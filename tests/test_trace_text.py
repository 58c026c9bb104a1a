"""trace = TRUE: the console text of a verbose call (round 5).  The reference prints one line per iteration and a summary
block (callback, src/nls.c:980-995; :610-630); the core formats exactly that text (csrc/trace_log.hpp, capi.hip) and the
bindings print it once the call is back.

CPU part (no device): gslnls_format_trace is pure host code -- it is fed the trajectory of the ORACLE on README example 2
and its output is compared line by line with the console output printed in the README (tests/golden/readme_traces.json,
"printed": README.md:568-605, :636-659, :772-795).
GPU part: the Python mirror's own stdout for the same calls, and the multi-start / IRLS / large-path lines."""
import ctypes as C
import re

import numpy as np
import pytest

FORMULA = "y ~ a * exp(-(x - b)^2 / (2 * c^2))"
NUM = re.compile(r"[-+]?(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?|inf|nan")


def same_line(got, exp, rtol):
    """identical text, or -- where a looser bound is stated -- the same text around numbers that agree to rtol"""
    if got == exp:
        return True
    if rtol <= 0:
        return False
    if NUM.sub("#", got) != NUM.sub("#", exp):
        return False
    a, b = [float(t) for t in NUM.findall(got)], [float(t) for t in NUM.findall(exp)]
    return len(a) == len(b) and all(abs(u - v) <= rtol * max(abs(v), 1e-300) for u, v in zip(a, b))


def compare(text, printed, loose_rows=(), loose_tol=5e-4, free=()):
    """line by line against the README's output.  loose_rows: iteration numbers whose numbers may differ in the 4th digit
    (README example 2 under LM crosses c ~ 0: tests/test_oracle_golden.py::_check_trace has the argument); free: prefixes of
    summary lines whose value is round-off itself (ssr achieved tolerance ~1e-15, the evaluation count of the last,
    round-off dominated iteration)"""
    got = text.split("\n")
    assert got[-1] == "" and len(got) - 1 == len(printed), (len(got) - 1, len(printed), text)
    exact = 0
    for g, e in zip(got, printed):
        if any(e.startswith(f) for f in free):
            assert g.startswith(next(f for f in free if e.startswith(f))), (g, e)
            continue
        m = re.match(r"iter +(\d+):", e)
        rtol = loose_tol if (m and int(m.group(1)) in loose_rows) else (1.2e-5 if m else 0.0)
        assert same_line(g, e, rtol), (g, e)
        exact += g == e
    return exact


def result_from_oracle(ref, maxiter, p):
    from gslnls_amd import _lib
    res = _lib.Result()
    keep = dict(partrace=np.full((maxiter + 1, p), np.nan, order="F"), ssrtrace=np.full(maxiter + 1, np.nan))
    keep["partrace"][:ref["niter"] + 1] = ref["partrace"]
    keep["ssrtrace"][:ref["niter"] + 1] = ref["ssrtrace"]
    res.partrace, res.ssrtrace = keep["partrace"].ctypes.data_as(_lib.DP), keep["ssrtrace"].ctypes.data_as(_lib.DP)
    res.niter, res.conv, res.ssr, res.ssrtol, res.chisq_init = ref["niter"], ref["conv"], ref["ssr"], ref["ssrtol"], ref["chisq_init"]
    res.neval[0], res.neval[1], res.neval[2] = ref["neval"]["f"], ref["neval"]["J"], ref["neval"]["fvv"]
    return res, keep


@pytest.mark.parametrize("variant", ["lm", "lmaccel", "lmaccel_fvv"])
def test_formatter_reproduces_the_readme_console_output_from_the_oracle_trajectory(gslref, readme, variant):
    from gslnls_amd import _lib
    from gslnls_amd.control import gsl_nls_control, pack_control
    ex = readme["ex2"]
    x, y = np.array(ex["x"]), np.array(ex["y"])
    fn = lambda th: th[0] * np.exp(-(x - th[1]) ** 2 / (2 * th[2] ** 2)) - y  # noqa: E731
    fvv = None
    if variant == "lmaccel_fvv":
        def fvv(th, v):
            a, b, c = th
            z = x - b
            e = np.exp(-z * z / (2 * c * c))
            fb, fc = a * e * z / c ** 2, a * e * z * z / c ** 3
            haa, hab, hac = 0.0, e * z / c ** 2, e * z * z / c ** 3
            hbb = a * e * (z * z / c ** 4 - 1 / c ** 2)
            hbc = a * e * (z ** 3 / c ** 5 - 2 * z / c ** 3)
            hcc = a * e * (z ** 4 / c ** 6 - 3 * z * z / c ** 4)
            del fb, fc
            return (haa * v[0] * v[0] + hbb * v[1] * v[1] + hcc * v[2] * v[2]
                    + 2 * (hab * v[0] * v[1] + hac * v[0] * v[2] + hbc * v[1] * v[2]))
    alg = "lm" if variant == "lm" else "lmaccel"
    ref = gslref.nls(50, 3, ex["start"], fn=fn, fvv=fvv, algorithm=alg, trace=True)
    assert ref["conv"] == 0 and ref["niter"] == ex[variant]["niter"]
    ctrl = gsl_nls_control()
    ci, _ = pack_control(ctrl, alg, True)
    res, keep = result_from_oracle(ref, ctrl["maxiter"], 3)
    L = _lib.lib()
    need = L.gslnls_format_trace(C.byref(res), 50, 3, ci.ctypes.data_as(_lib.IP), 0, None, 0)
    buf = C.create_string_buffer(need + 1)
    assert L.gslnls_format_trace(C.byref(res), 50, 3, ci.ctypes.data_as(_lib.IP), 0, buf, need + 1) == need
    text = buf.value.decode()
    printed = ex[variant]["printed"]
    free = ("ssr achieved tolerance:",) + (("function evaluations:",) if variant == "lm" else ())
    exact = compare(text, printed, loose_rows=range(7, 21) if variant == "lm" else (), free=free)
    # the formatter itself is exact: every line outside the stated round-off cases is the README's text
    assert exact >= len(printed) - (15 if variant == "lm" else 1) - len(free), (exact, len(printed))
    del keep


def test_trace_text_is_empty_without_a_verbose_call():
    from gslnls_amd import _lib
    L = _lib.lib()
    assert isinstance(_lib.trace_text(), str)
    order = np.array([2, 0, 1], dtype=np.int32)
    assert L.gslnls_trace_set_order(order.ctypes.data_as(_lib.IP), 3) == 0
    bad = np.array([0, 0, 1], dtype=np.int32)
    assert L.gslnls_trace_set_order(bad.ctypes.data_as(_lib.IP), 3) == 4  # not a permutation: EINVAL
    assert L.gslnls_trace_set_order(None, 0) == 0


# ---- the mirror's console output on the device -----------------------------------------------------------------------
@pytest.fixture(scope="module")
def amd():
    import gslnls_amd
    from gslnls_amd import _lib
    assert _lib.lib().gslnls_device_count() >= 1, "no MI355X visible: the HIP path cannot be tested"
    return gslnls_amd


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["lm", "lmaccel", "lmaccel_fvv"])
def test_mirror_prints_readme_example_2_line_by_line(amd, readme, capsys, variant):
    ex = readme["ex2"]
    kw = dict(algorithm="lm" if variant == "lm" else "lmaccel")
    if variant == "lmaccel_fvv":
        kw["fvv"] = True
    fit = amd.gsl_nls(FORMULA, data=dict(x=ex["x"], y=ex["y"]), start=dict(a=1, b=0, c=1), trace=True, **kw)
    text = capsys.readouterr().out
    assert fit["conv"] == 0 and fit["niter"] == ex[variant]["niter"]
    free = ("ssr achieved tolerance:",) + (("function evaluations:",) if variant == "lm" else ())
    exact = compare(text, ex[variant]["printed"], loose_rows=range(7, 21) if variant == "lm" else (), free=free)
    assert exact >= len(ex[variant]["printed"]) - (15 if variant == "lm" else 1) - len(free)
    quiet = amd.gsl_nls(FORMULA, data=dict(x=ex["x"], y=ex["y"]), start=dict(a=1, b=0, c=1), **kw)
    assert capsys.readouterr().out == "" and quiet["niter"] == fit["niter"]


@pytest.mark.gpu
def test_trace_prints_parameters_in_the_callers_order(amd, readme, capsys):
    """a formula matched against a hand-written device model with its parameters in another order: the printed vectors are
    the caller's (gslnls_trace_set_order), as partrace is"""
    e1 = readme["ex1"]
    fit = amd.gsl_nls("y ~ b + A * exp(-lam * x)", data=dict(x=e1["x"], y=e1["y"]), start=dict(b=0, lam=0, A=0), trace=True)
    lines = capsys.readouterr().out.split("\n")
    assert fit["niter"] == 9
    for it in range(1, 10):
        nums = [float(t) for t in NUM.findall(lines[it - 1].split("par = ")[1])]
        assert np.allclose(nums, fit["partrace"][it], rtol=1e-5, atol=1e-12), (lines[it - 1], fit["partrace"][it])
    assert lines[9] == "*" * 19 and lines[10] == "summary from method 'multifit/levenberg-marquardt'"


@pytest.mark.gpu
def test_multistart_irls_and_large_lines(amd, gslref, nist, readme, capsys):
    # multi-start (src/nls_mstart.c:331-337, src/nls.c:510-517): one line per accepted stationary point, the closing lines
    q = nist["BoxBOD"]
    fit = amd.gsl_nls(q["formula"], data=q["data"], start=dict(b1=[200, 250], b2=[0, 1]), trace=True,
                      control=dict(mstart_n=5, mstart_q=1, mstart_r=1.1))
    out = capsys.readouterr().out.split("\n")
    ms = [ln for ln in out if ln.startswith("mstart ssr* = ")]
    assert len(ms) >= fit["mstart"]["nsp"] >= 1 and ", det(JTJ) = " in ms[0] and ", NSP = 1, NWSP = 0, par = (" in ms[0]
    fin = [i for i, ln in enumerate(out) if ln.startswith("multi-start algorithm finished successfully (NSP = %d, NWSP = %d, # iterations = %d)"
                                                           % (fit["mstart"]["nsp"], fit["mstart"]["nwsp"], fit["mstart"]["iters"]))]
    assert len(fin) == 1 and out[fin[0] + 1] == "*" * 19 and out[fin[0] + 2].startswith("iter   1: ssr = ")
    assert out[-2] == "*" * 19 and out[-3] == "status: success"
    # robust loss (src/nls_irls.c:466-472): one line per IRLS iteration, no iteration lines, IRLS rows in the summary
    e1 = readme["ex1"]
    fit = amd.gsl_nls("y ~ A * exp(-lam * x) + b", data=dict(x=e1["x"], y=e1["y"]), start=dict(A=0, lam=0, b=0), loss="huber", trace=True)
    out = capsys.readouterr().out.split("\n")
    irls = [ln for ln in out if ln.startswith("IRLS iter: ")]
    assert len(irls) == fit["irls"]["irls_niter"] == e1["huber"]["irls_niter"]
    assert irls[0].startswith("IRLS iter:   1, weighted ssr: ") and not any(ln.startswith("iter ") for ln in out)
    assert "IRLS number of iterations: %d" % fit["irls"]["irls_niter"] in out and "IRLS convergence status: success" in out
    tol_line = [ln for ln in out if ln.startswith("IRLS achieved tolerance: ")][0]
    assert tol_line == "IRLS achieved tolerance: %g" % fit["irls"]["irls_tol"]
    assert abs(float(tol_line.split(": ")[1]) - e1["huber"]["irls_tol"]) < 5e-8  # (README.md:501-505 prints four digits of it)
    # large path (src/nls_large.c:259-273, :715-739)
    rng = np.random.Generator(np.random.PCG64(11))
    n, p = 4000, 16
    A = rng.uniform(-1, 1, (n, p)) / np.sqrt(p)
    th = rng.normal(0, 0.25, p)
    yv = np.exp(A @ th) * (1 + 0.01 * rng.standard_normal(n))
    for alg, name in (("cgst", "steihaug-toint"), ("lm", "levenberg-marquardt")):
        fit = amd.gsl_nls_large("glmexp", A=A, y=yv, start=np.zeros(p), algorithm=alg, trace=True)
        out = capsys.readouterr().out.split("\n")
        its = [ln for ln in out if ln.startswith("iter ")]
        assert len(its) == fit["niter"] and re.match(r"iter   1: ssr = \S+, \|x\|\^2 = \S+, cond\(J\) = \S+$", its[0])
        assert its[0].endswith("cond(J) = inf") == (alg == "cgst")
        assert "summary from method 'multilarge/%s'" % name in out and "status = success" in out
        assert "reason for stopping: %s" % {1: "input domain error", 2: "output range error"}[fit["info"]] in out
