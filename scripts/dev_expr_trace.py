"""developer aid: GPU expression-model fit vs oracle, iteration by iteration (ssr trace)"""
import sys, json, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gslnls_amd as amd
import gslref
from test_oracle_golden import nist_callbacks
pbs = {q["name"]: q for q in json.load(open(os.path.join(ROOT, "tests/golden/nist_formula_problems.json")))}
for name in sys.argv[1:]:
    q = pbs[name]
    data = {k: np.asarray(v, dtype=np.float64) for k, v in q["data"].items()}
    fit = amd.gsl_nls(q["formula"], data=data, start=q["start"], jac=False, trace=True)
    fn, names = nist_callbacks(q)
    ref = gslref.nls(q["n"], q["p"], list(q["start"].values()), fn=fn, trace=True)
    print(name, "niter", fit["niter"], ref["niter"])
    a, b = np.asarray(fit["ssrtrace"]), np.asarray(ref["ssrtrace"])
    for i in range(max(fit["niter"], ref["niter"]) + 1):
        print(i, "%.15e %.15e" % (a[i] if i < len(a) else np.nan, b[i] if i < len(b) else np.nan))
