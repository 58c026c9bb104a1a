#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the reference tree.

Run HERE only (needs /root/reference, which does not exist on the GPU box):

    python tests/golden/make_fixtures.py

What it harvests (data only: inputs and expected outputs the reference's own
tests/docs hold; no reference source text is stored):

  nist_formula_problems.json   33 formula problems: data, model formula string,
                               start, certified target   (R/nls_test.R:169-979)
  mgh_function_problems.json   26 function problems: start, f(start), J(start),
                               known solution, evaluated from the reference's
                               Fortran catalogue src/test_nls.f90 compiled with
                               flang into oracle/_ref/libtest_nls.so
  readme_traces.json           README.md example inputs (regenerated with R's
                               set.seed(1)/rnorm recipe, checked against the
                               literal y of unit_tests_gslnls.R:259-265) and the
                               printed iteration traces / counts / estimates
  unit_test_pins.json          scalars pinned in inst/unit_tests/unit_tests_gslnls.R
  sobol_d2.json                first 64 points of the 2-d Sobol sequence in GSL
                               order (= SciPy unscrambled Sobol without point 0)
"""
import ctypes
import json
import os
import re
import subprocess
import sys

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))


# --------------------------------------------------------------------------
# 1. NIST / formula problems
# --------------------------------------------------------------------------
def _matching_paren(s, i):
    depth = 0
    for j in range(i, len(s)):
        if s[j] == "(":
            depth += 1
        elif s[j] == ")":
            depth -= 1
            if depth == 0:
                return j
    raise ValueError("unbalanced")


def _split_top(s):
    out, depth, cur, inq = [], 0, "", False
    for ch in s:
        if ch == '"':
            inq = not inq
        elif inq:
            pass
        elif ch == "(":
            depth += 1
        elif ch == ")":
            depth -= 1
        if ch == "," and depth == 0 and not inq:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def _num(tok):
    tok = tok.strip()
    return float(tok)


def _parse_c(s):
    """parse 'c(a = 1, b = 2)' or 'c(1, 2, 3)' -> (names or None, values)"""
    s = s.strip()
    assert s.startswith("c("), s[:40]
    inner = s[2:_matching_paren(s, 1)]
    names, vals = [], []
    for tok in _split_top(inner):
        if "=" in tok:
            k, v = tok.split("=")
            names.append(k.strip())
            vals.append(_num(v))
        else:
            vals.append(_num(tok))
    return (names or None), vals


def parse_nist():
    src = open(os.path.join(REF, "R/nls_test.R")).read()
    lines = src.split("\n")
    # block starts
    starts = [(m.start(), m.group(1)) for m in re.finditer(r'identical\(name, "([^"]+)"\)\) \{\n\s+\.data', src)]
    problems = []
    for bi, (pos, name) in enumerate(starts):
        end = starts[bi + 1][0] if bi + 1 < len(starts) else src.index("return(", pos)
        blk = src[pos:end]
        line_no = src[:pos].count("\n") + 1
        # data.frame
        i = blk.index("data.frame(")
        j = _matching_paren(blk, i + len("data.frame"))
        df_inner = blk[i + len("data.frame("):j]
        data = {}
        for tok in _split_top(df_inner):
            k, v = tok.split("=", 1)
            _, vals = _parse_c(v)
            data[k.strip()] = vals
        fm = re.search(r'as\.formula\("([^"]+)"', blk).group(1)
        st = blk.index(".start <- ") + len(".start <- ")
        sn, sv = _parse_c(blk[st:])
        tg = blk.index(".target <- ") + len(".target <- ")
        tn, tv = _parse_c(blk[tg:])
        assert sn == tn, (name, sn, tn)
        ncol = {len(v) for v in data.values()}
        assert len(ncol) == 1, name
        problems.append(dict(name=name, formula=fm, data=data, start=dict(zip(sn, sv)),
                             target=dict(zip(tn, tv)), n=ncol.pop(), p=len(sv),
                             cite="R/nls_test.R:%d" % line_no))
    return problems


# --------------------------------------------------------------------------
# 2. MGH function problems via the compiled Fortran catalogue
# --------------------------------------------------------------------------
def parse_problem_table():
    src = open(os.path.join(REF, "R/nls_test.R")).read()
    blk = src[src.index("properties <- data.frame("):src.index("return(properties[, fields])")]

    def grab(key, conv):
        i = blk.index(key + " = c(") + len(key) + 3
        j = _matching_paren(blk, i + 1)
        toks = _split_top(blk[i + 2:j])
        return [conv(t.strip()) for t in toks]

    names = grab("name", lambda t: t.strip('"'))
    klass = grab("class", lambda t: t.strip('"'))
    p = grab("p", lambda t: int(t.rstrip("L")))
    n = grab("n", lambda t: int(t.rstrip("L")))
    check = grab("check", lambda t: t.strip('"'))
    return names, klass, p, n, check


def build_fortran():
    out = os.path.join(REPO, "oracle/_ref/libtest_nls.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if not os.path.exists(out):
        subprocess.check_call(["/opt/rocm/lib/llvm/bin/flang", "-shared", "-fPIC", "-O1", "-o", out,
                               os.path.join(REF, "src/test_nls.f90")])
    return ctypes.CDLL(out)


def mgh_problems():
    lib = build_fortran()
    names, klass, ps, ns, check = parse_problem_table()
    R_src = open(os.path.join(REF, "R/nls_test.R")).read()
    out = []
    ci = ctypes.c_int
    dp = ctypes.POINTER(ctypes.c_double)
    for fid0, (nm, kl, p, n, ck) in enumerate(zip(names, klass, ps, ns, check)):
        if kl != "function":
            continue
        nprob = (fid0 + 1) - 33  # src/nls_test.c:8
        x = np.zeros(p)
        lib.p00_start_(ctypes.byref(ci(nprob)), ctypes.byref(ci(p)), x.ctypes.data_as(dp))
        f = np.zeros(n)
        lib.p00_f_(ctypes.byref(ci(nprob)), ctypes.byref(ci(n)), ctypes.byref(ci(p)), x.ctypes.data_as(dp),
                   f.ctypes.data_as(dp))
        J = np.zeros((p, n))  # column-major m x n in Fortran == (p, n) C-order transposed
        lib.p00_j_(ctypes.byref(ci(nprob)), ctypes.byref(ci(n)), ctypes.byref(ci(p)), x.ctypes.data_as(dp),
                   J.ctypes.data_as(dp))
        known = ci(0)
        sol = np.zeros(p)
        lib.p00_sol_(ctypes.byref(ci(nprob)), ctypes.byref(ci(n)), ctypes.byref(ci(p)), ctypes.byref(known),
                     sol.ctypes.data_as(dp))
        target = sol.tolist() if known.value else None
        # completed targets listed in R/nls_test.R:1012-1047
        m = re.search(r'identical\(name, "%s"\)\) \{\s+\.start_sol\[\["target"\]\] <- (c\()' % re.escape(nm), R_src)
        if m:
            _, tv = _parse_c(R_src[m.start(1):])
            target = tv
        out.append(dict(name=nm, nprob=nprob, n=n, p=p, check=ck, start=x.tolist(), f_start=f.tolist(),
                        J_start_rowmajor=J.T.reshape(-1).tolist(), target=target,
                        cite="src/test_nls.f90 via src/nls_test.c:8 (nprob = id - 33)"))
    return out


# --------------------------------------------------------------------------
# 3. R's RNG: set.seed(seed) + rnorm (Mersenne-Twister + inversion)
# --------------------------------------------------------------------------
class RRng:
    """R's default RNG: MT19937 seeded through the LCG scrambler, rnorm by inversion
    (R sources: src/main/RNG.c, src/nmath/snorm.c)."""

    def __init__(self, seed):
        s = int(seed) & 0xFFFFFFFF
        for _ in range(50):
            s = (69069 * s + 1) & 0xFFFFFFFF
        dummy = np.zeros(625, dtype=np.uint32)
        for j in range(625):
            s = (69069 * s + 1) & 0xFFFFFFFF
            dummy[j] = s
        dummy[0] = 624
        self.mt = dummy[1:].copy()
        self.mti = 624

    def _genrand(self):
        N, M = 624, 397
        mt = self.mt
        if self.mti >= N:
            for kk in range(N):
                y = (int(mt[kk]) & 0x80000000) | (int(mt[(kk + 1) % N]) & 0x7FFFFFFF)
                v = int(mt[(kk + M) % N]) ^ (y >> 1) ^ (0x9908B0DF if (y & 1) else 0)
                mt[kk] = np.uint32(v)
            self.mti = 0
        y = int(mt[self.mti])
        self.mti += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y & 0xFFFFFFFF

    def unif(self):
        v = self._genrand() * 2.3283064365386963e-10
        if v <= 0.0:
            return 0.5 * 2.328306437080797e-10
        if 1.0 - v <= 0.0:
            return 1.0 - 0.5 * 2.328306437080797e-10
        return v

    def rnorm(self, n, mean=0.0, sd=1.0):
        from scipy.special import ndtri
        BIG = 134217728.0
        out = np.empty(n)
        for i in range(n):
            u = self.unif()
            u = int(BIG * u) + self.unif()
            out[i] = ndtri(u / BIG)
        return mean + sd * out


def _parse_trace(lines):
    tr = []
    for ln in lines:
        m = re.match(r"#> iter\s+(\d+): ssr = ([^,]+), par = \(([^)]*)\)", ln)
        if m:
            tr.append(dict(iter=int(m.group(1)), ssr=float(m.group(2)),
                           par=[float(t) for t in m.group(3).split(",")]))
    return tr


def readme_traces():
    readme = open(os.path.join(REF, "README.md")).read().split("\n")
    ut = open(os.path.join(REF, "inst/unit_tests/unit_tests_gslnls.R")).read()
    # literal Ex.1 data from the unit tests (:254-265)
    i = ut.index("x <- c(0, 0.125")
    _, x_lit = _parse_c(ut[i + 5:])
    j = ut.index("y <- c(5.84338654731442")
    _, y_lit = _parse_c(ut[j + 5:])

    # Ex.1 regenerated: README.md:157-163
    rng = RRng(1)
    n = 25
    x1 = (np.arange(1, n + 1) - 1) * 3 / (n - 1)
    y1 = 5 * np.exp(-1.5 * x1) + 1 + rng.rnorm(n, sd=0.25)
    err = float(np.max(np.abs(y1 - np.array(y_lit))))
    assert err < 1e-13, err
    assert np.max(np.abs(x1 - np.array(x_lit))) == 0.0

    # Ex.2: README.md:540-546
    rng = RRng(1)
    n2 = 50
    x2 = np.arange(1, n2 + 1) / n2
    y2 = 5 * np.exp(-(x2 - 0.4) ** 2 / (2 * 0.15 ** 2)) * rng.rnorm(n2, mean=1.0, sd=0.1)

    def block(first, last):
        return readme[first - 1:last]

    ex2_lm = _parse_trace(block(568, 593))
    ex2_accel = _parse_trace(block(636, 647))
    ex2_accel_fvv = _parse_trace(block(772, 783))
    assert len(ex2_lm) == 26 and len(ex2_accel) == 12 and len(ex2_accel_fvv) == 12, (
        len(ex2_lm), len(ex2_accel), len(ex2_accel_fvv))

    def printed(first, last):
        """the console output of a verbose call as the README shows it (iteration lines + summary block), without the
        "#> " prefix: expected OUTPUT of the path, the text the trace = TRUE tests compare with line by line"""
        out = [ln[3:] for ln in block(first, last)]
        assert all(ln.startswith("#> ") for ln in block(first, last)) and out[-1] == "*" * 19, out[-1]
        return out
    return dict(
        rng_check_max_abs_err=err,
        ex1=dict(cite="README.md:157-195, :246-268, :405-409; data literal unit_tests_gslnls.R:254-265",
                 model="A*exp(-lam*x)+b", x=x_lit, y=y_lit, start=[0.0, 0.0, 0.0],
                 niter=9, coef=[4.893019, 1.416863, 1.009742], ssr=1.316,
                 se=[0.1811, 0.1304, 0.1092], sigma=0.2446, df=22,
                 huber=dict(cite="README.md:493-505", coef=[4.796, 1.463, 1.092], wssr=0.8127,
                            irls_niter=8, irls_tol=0.0001023, nls_niter=9)),
        ex2=dict(cite="README.md:540-604, :636-658, :772-794",
                 model="a*exp(-(x-b)^2/(2*c^2))", x=x2.tolist(), y=y2.tolist(), start=[1.0, 0.0, 1.0],
                 lm=dict(trace=ex2_lm, niter=26, initial_ssr=210.146, final_ssr=2.7583,
                         neval_f=124, neval_J=0, ssrtol=1.33227e-15, printed=printed(568, 605)),
                 lmaccel=dict(trace=ex2_accel, niter=12, final_ssr=2.7583, neval_f=76, neval_J=0, neval_fvv=0,
                              printed=printed(636, 659)),
                 lmaccel_fvv=dict(trace=ex2_accel_fvv, niter=12, neval_f=58, neval_fvv=18, printed=printed(772, 795))),
    )


def unit_test_pins():
    return dict(
        cite="inst/unit_tests/unit_tests_gslnls.R",
        tol_abs=float(np.finfo(float).eps ** 0.25),
        misra1a=dict(deviance=0.1245514, sigma=0.1018788, cite=":353-356"),
        misra1a_huber_sigma=dict(value=0.1342963, cite=":379-380"),
        madsen=dict(deviance=0.7731991, sigma=0.8793174, cite=":398-399"),
        boxbod_wrong_basin=dict(start=[1.0, 1.0], coef=[172.5, 114.8], ssr=9771.5, cite="SURVEY.md App. B.4"),
        penalty_p500_ssr=dict(value=0.004778845, cite="README.md:1078-1101"),
        ratkowsky2=dict(niter=10, ssr=8.057, cite="README.md:1264-1274"),
        madsen_lm=dict(niter=42, start=[3.0, 1.0], coef=[-0.155489, 0.69456], cite="README.md:1286-1295"),
    )


def sobol_d2():
    from scipy.stats import qmc
    s = qmc.Sobol(d=2, scramble=False)
    pts = s.random(65)[1:]  # GSL's first returned point is (0.5, 0.5) == SciPy's 2nd
    return dict(cite="GSL qrng/sobol.c order; dims 1-2 (Bratley-Fox == Joe-Kuo there)", points=pts.tolist())


def main():
    out = {
        "nist_formula_problems.json": parse_nist(),
        "mgh_function_problems.json": mgh_problems(),
        "readme_traces.json": readme_traces(),
        "unit_test_pins.json": unit_test_pins(),
        "sobol_d2.json": sobol_d2(),
    }
    for fn, obj in out.items():
        with open(os.path.join(HERE, fn), "w") as fh:
            json.dump(obj, fh, indent=None, separators=(",", ":"))
        print(fn, os.path.getsize(os.path.join(HERE, fn)), "bytes")
    print("nist:", len(out["nist_formula_problems.json"]), "mgh:", len(out["mgh_function_problems.json"]))


if __name__ == "__main__":
    sys.exit(main())
