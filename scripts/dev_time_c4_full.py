"""C4 end to end: gsl_nls multi-start on BoxBOD with 8192 points per major iteration (not only the batch kernel)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import gslnls_amd as A
d = dict(x=np.array([1., 2., 3., 5., 7., 10.]), y=np.array([109., 149., 149., 191., 213., 224.]))
for n in (5, 512, 8192, 65536):
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        fit = A.gsl_nls("y ~ b1*(1-exp(-b2*x))", data=d, start=dict(b1=[1.0, 500.0], b2=[0.01, 5.0]), jac=True,
                        control=dict(mstart_n=n, mstart_q=max(1, n // 10), solver="cholesky"))
        ts.append(time.perf_counter() - t0)
    print("mstart_n=%6d: %.2f ms wall (best of 3); par=%s ssr=%.7f conv=%d" % (n, 1e3 * min(ts), np.asarray(fit["par"]), fit["ssr"], fit["conv"]))
