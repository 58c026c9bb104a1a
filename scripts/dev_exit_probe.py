"""developer probe: does the interpreter exit cleanly with (a) a background build in flight, (b) a finished one,
(c) native code only?  usage: dev_exit_probe.py a|b|c"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["GSLNLS_JIT_CACHE"] = "/tmp/gslnls_exit_probe_%s_%d" % (sys.argv[1], os.getpid())
import gslnls_amd as amd
mode = sys.argv[1]
x = np.linspace(0, 3, 1000)
y = 5 * np.exp(-1.5 * x) + 1 + 0.01 * np.cos(37 * x)
low = "jit" if mode == "c" else "auto"
for k in range(3 if mode != "c" else 1):
    fit = amd.gsl_nls("y ~ A/exp(lam*x) + b*%d" % (k + 1), data=dict(x=x, y=y), start=dict(A=1.0, lam=1.0, b=0.0), jac=True, lowering=low)
    print(mode, k, "code_path", fit["code_path"], fit["par"], flush=True)
if mode == "b":
    time.sleep(8.0)
print("leaving", flush=True)
