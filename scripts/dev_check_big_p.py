"""developer aid: a function model with p = 1500 and p = 4096 parameters (linear in its parameters: y = X theta) through gsl_nls() --
the matrix path's device epilogue ((J^T J)^-1 by bd_trinv_kernel + X^T X, condition number by power iterations) at sizes the
tests do not reach: against the host routines at p = 1500, against X^T X cov = I at p = 4096."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gslnls_amd as amd
for p, n, host in ((1500, 1700, True), (4096, 4300, False)):
    rng = np.random.default_rng(p)
    X = rng.standard_normal((n, p)) + 0.1
    truth = rng.standard_normal(p)
    y = X @ truth + 1e-3 * rng.standard_normal(n)
    Xf = np.asfortranarray(X)
    fn = lambda th: X @ th
    jac = lambda th: Xf
    t0 = time.perf_counter()
    fit = amd.gsl_nls(fn, y=y, start=np.zeros(p), jac=jac, control=dict(solver="cholesky"))
    el = time.perf_counter() - t0
    cov = np.asarray(fit["covar"])
    chk = np.max(np.abs((X.T @ (X @ cov[:, :8])) - np.eye(p)[:, :8]))
    print("p = %d n = %d: conv %d niter %d  %.1f ms  |par - truth| %.2e  cond %.3e  max |X'X cov - I| (8 columns) %.2e  symmetric %s" % (
        p, n, fit["conv"], fit["niter"], el * 1e3, np.max(np.abs(fit["par"] - truth)), fit["jtj_cond"], chk, np.array_equal(cov, cov.T)), flush=True)
    if host:
        os.environ["GSLNLS_BD_HOST_EPILOGUE"] = "1"
        t0 = time.perf_counter()
        fh = amd.gsl_nls(fn, y=y, start=np.zeros(p), jac=jac, control=dict(solver="cholesky"))
        eh = time.perf_counter() - t0
        os.environ.pop("GSLNLS_BD_HOST_EPILOGUE")
        ch = np.asarray(fh["covar"])
        sc = np.sqrt(np.outer(np.diag(ch), np.diag(ch)))
        print("          host routines: %.1f ms; covariance device vs host %.2e of its scale, cond %.6e vs %.6e" % (
            eh * 1e3, np.max(np.abs(cov - ch) / sc), fit["jtj_cond"], fh["jtj_cond"]), flush=True)
