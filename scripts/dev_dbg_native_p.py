"""developer aid: interpreter vs native code for p = 2 .. 9 (first two LM iterations of a sum of exponentials)"""
import sys, numpy as np
sys.path.insert(0, "/root/repo")
import gslnls_amd as A
x = np.linspace(0.0, 4.0, 300)
rng = np.random.Generator(np.random.PCG64(9))
for p in range(2, 10):
    names, terms, truth = [], [], []
    k = 0
    while len(names) + 2 <= p:
        k += 1
        names += ["a%d" % k, "b%d" % k]; terms.append("a%d*exp(-b%d*x)" % (k, k)); truth += [2.0 + k, 0.4 * k]
    if len(names) < p:
        names.append("c"); terms.append("c"); truth.append(0.5)
    truth = np.array(truth)
    rhs = " + ".join(terms)
    y = sum(truth[2 * j] * np.exp(-truth[2 * j + 1] * x) for j in range(k)) + (truth[-1] if len(truth) % 2 else 0.0) + 0.01 * rng.standard_normal(len(x))
    start = truth * (1.0 + 0.05 * np.where(np.arange(p) % 2 == 0, 1.0, -1.0))
    out = []
    for low in ("vm", "jit"):
        for jac in (True, False):
            fit = A.gsl_nls("y ~ " + rhs, data=dict(x=x, y=y), start=dict(zip(names, start)), jac=jac, control=dict(solver="cholesky", maxiter=2), lowering=low, trace=True)
            out.append((low, jac, fit["conv"], fit["code_path"], ["%.15g" % v for v in fit["ssrtrace"]], fit["neval"]["f"]))
    same = out[0][4] == out[2][4] and out[1][4] == out[3][4]
    print("p =", p, "native == interpreter:", same)
    if not same:
        for o in out:
            print("   ", o)
