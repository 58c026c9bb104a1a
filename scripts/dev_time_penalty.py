"""README Example 4 (penalty function, p = 500): wall time of gsl_nls_large through the sparse callback path"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, scipy.sparse as sp
import gslnls_amd as A
p = 500
a = np.sqrt(1e-5)
eye = sp.identity(p, format="csr") * a
fn = lambda th: np.concatenate([a * (th - 1.0), [np.sum(th ** 2) - 0.25]])
jac = lambda th: sp.vstack([eye, sp.csr_matrix(2.0 * th.reshape(1, -1))]).tocsc()
for alg in ("cgst", "lm"):
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        fit = A.gsl_nls_large(fn, y=np.zeros(p + 1), start=np.arange(1.0, p + 1), algorithm=alg, jac=jac, control=dict(maxiter=500))
        ts.append(time.perf_counter() - t0)
    print("%-4s: %.1f ms wall (best of 3), niter %d, ssr %.9f, device passes %d" % (alg, 1e3 * min(ts), fit["niter"], fit["ssr"], fit["n_passes"]))
