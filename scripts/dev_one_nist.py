"""developer aid: one NIST formula problem through the expression path, trace printed (run on the GPU box)"""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import gslnls_amd as amd
name = sys.argv[1] if len(sys.argv) > 1 else "Roszman1"
q = {p["name"]: p for p in json.load(open(os.path.join(ROOT, "tests", "golden", "nist_formula_problems.json")))}[name]
data = {k: np.asarray(v, dtype=np.float64) for k, v in q["data"].items()}
for jac in (False, True):
    fit = amd.gsl_nls(q["formula"], data=data, start=q["start"], jac=jac, trace=True)
    tr = np.asarray(fit["ssrtrace"])
    print(name, "jac", jac, "conv", fit["conv"], "niter", fit["niter"], "neval", fit["neval"], "par", fit["par"], "ssr", fit["ssr"])
    print("  trace", tr[:min(len(tr), fit["niter"] + 1)][:12], "...", tr[max(0, fit["niter"] - 3):fit["niter"] + 1])
    pt = np.asarray(fit["partrace"])
    print("  partrace[0..2]", pt[:3].tolist() if pt.ndim == 2 else pt[:12])
