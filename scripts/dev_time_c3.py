"""C3: gsl_nls_large cgst, synthetic GLM n = 1e7, p = 64 (5.12 GB): pass GB/s and a whole fit"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import gslnls_amd as A_

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
p = 64
rng = np.random.Generator(np.random.PCG64(20250928))
t0 = time.time()
A = rng.uniform(-1.0, 1.0, size=(n, p))
A /= np.sqrt(p)
th = 0.25 * rng.standard_normal(p)
y = np.exp(A @ th) * (1.0 + 0.01 * rng.standard_normal(n))
print("generated in %.1f s" % (time.time() - t0), flush=True)
t0 = time.time()
prob = A_.LargeProblem(5, p, A, y)
print("uploaded in %.1f s" % (time.time() - t0), flush=True)
x = np.zeros(p)
u = rng.standard_normal(p)
byt = 8.0 * n * p + 16.0 * n
for mode, nm in ((0, "EVAL (ssr, J^T f, diag J^T J; writes m, f)"), (1, "fused J^T J u")):
    ms = prob.time_pass(mode, x, u, reps=10)
    print("%-45s %.3f ms/pass -> %.0f GB/s (algorithmic %.2f GB)" % (nm, ms, byt / ms / 1e6, byt / 1e9))
ms = prob.time_pass(2, x, u, reps=5)
print("%-45s %.3f ms -> %.2f TFLOP/s fp64 (2 n p^2 = %.1f GFLOP), %.0f GB/s" % ("full J^T J (MFMA f64 16x16x4)", ms, 2.0 * n * p * p / ms / 1e9, 2.0 * n * p * p / 1e9, (8.0 * n * p + 8.0 * n) / ms / 1e6))
t0 = time.time()
fit = prob.solve(x, "cgst", want_resid=False)
el = time.time() - t0
print("cgst fit: niter=%d conv=%d ssr=%.6g passes=%d neval=%s wall=%.3f s -> %.2f outer it/s, max|par-th|=%.3g" % (
    fit["niter"], fit["conv"], fit["ssr"], fit["n_passes"], fit["neval"], el, fit["niter"] / el, np.max(np.abs(fit["par"] - th))))
prob.close()
