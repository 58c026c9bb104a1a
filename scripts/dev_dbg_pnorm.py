"""developer aid: interpreter vs native code on small formulas around pnorm (first LM iteration: ssr after one step)"""
import sys, os, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/oracle")
import gslnls_amd as A
from gslnls_amd import formula as F
CASES = [("a * pnorm(x) + c", dict(a=3.0, c=0.5)),
         ("pnorm((x - m) / s)", dict(m=0.4, s=1.3)),
         ("a * pnorm((x - m) / s) + c", dict(a=3.0, m=0.4, s=1.3, c=0.5)),
         ("a * dnorm((x - m) / s) + c * pnorm(x)", dict(a=3.0, m=0.4, s=1.3, c=0.5)),
         ("a * exp(-0.5 * ((x - m) / s)^2) + c", dict(a=3.0, m=0.4, s=1.3, c=0.5))]
for rhs_text, pars in CASES:
    names = list(pars); truth = np.array([pars[k] for k in names])
    rng = np.random.Generator(np.random.PCG64(5))
    x = np.linspace(-3, 3, 200)
    rhs = F.parse_expr(rhs_text)
    def model(t):
        env = {"x": x}; env.update({k: t[i] for i, k in enumerate(names)})
        return np.asarray(F.evaluate(rhs, env), dtype=np.float64) * np.ones(len(x))
    y = model(truth) + 0.01 * rng.standard_normal(len(x))
    start = truth * (1.0 + 0.03 * np.where(np.arange(len(names)) % 2 == 0, 1.0, -1.0))
    print(rhs_text)
    for low in ("vm", "jit"):
        for jac in (True, False):
            fit = A.gsl_nls("y ~ " + rhs_text, data=dict(x=x, y=y), start=dict(zip(names, start)), jac=jac, control=dict(solver="cholesky", maxiter=1), lowering=low, trace=True)
            print("  ", low, jac, "conv", fit["conv"], "code", fit["code_path"], "ssr0 %.17g" % fit["ssrtrace"][0], "ssr1 %.17g" % (fit["ssrtrace"][1] if len(fit["ssrtrace"]) > 1 else float("nan")), fit["neval"])
