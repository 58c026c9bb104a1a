"""developer aid: first iterations of one NIST problem, FD Jacobian, expression path (interpreter and native)"""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import gslnls_amd as amd
name = sys.argv[1] if len(sys.argv) > 1 else "Roszman1"
q = {p["name"]: p for p in json.load(open(os.path.join(ROOT, "tests", "golden", "nist_formula_problems.json")))}[name]
data = {k: np.asarray(v, dtype=np.float64) for k, v in q["data"].items()}
for low in ("vm", "jit"):
    for scale in ("more", "levenberg"):
        for jac in (False, True):
            fit = amd.gsl_nls(q["formula"], data=data, start=q["start"], jac=jac, trace=True, lowering=low, control=dict(maxiter=1, scale=scale))
            print(name, low, scale, "jac", jac, "niter", fit["niter"], "neval", fit["neval"], "par", fit["par"].tolist(), "ssr", fit["ssr"], flush=True)
