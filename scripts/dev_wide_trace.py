"""developer script: trace of a wide fit (p = 32) beside the oracle's"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import gslref
import gslnls_amd as amd
from test_gpu_wide import gaussians_problem
ng, extra, n = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (10, 2, 100000)))
q = gaussians_problem(ng, extra, n, seed=7 + ng)
p = len(q["truth"])
fit = amd.gsl_nls(q["formula"], data=dict(x=q["x"], y=q["y"]), start=dict(zip(q["names"], q["start"])), jac=True,
                  control=dict(solver="cholesky"), trace=True)
ref = gslref.nls(n, p, q["start"], fn=lambda th: q["model"](th) - q["y"], jac=q["jac"], ctrl=gslref.control(solver="cholesky"), trace=True)
print("gpu niter", fit["niter"], "conv", fit["conv"], "neval", fit["neval"], "loop_ms", fit["loop_ms"], "launches", fit["n_launches"])
print("ref niter", ref["niter"], "conv", ref["conv"], "neval", ref["neval"])
k = max(fit["niter"], ref["niter"]) + 1
for i in range(min(k, 30)):
    a = fit["ssrtrace"][i] if i < len(fit["ssrtrace"]) else np.nan
    b = ref["ssrtrace"][i] if i < len(ref["ssrtrace"]) else np.nan
    print(i, "%.15g %.15g" % (a, b))
print("par diff", np.max(np.abs(fit["par"] - ref["par"]) / np.abs(ref["par"])))
