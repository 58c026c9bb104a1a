import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import c2_data
import gslnls_amd as A
x, y = c2_data(1_000_000)
ctrl = A.gsl_nls_control(solver="cholesky", xtol=1.49e-8, gtol=1.49e-8)
prob = A.DenseProblem(1, 3, x, y)
for _ in range(6):
    fit = prob.solve([1.0, 1.0, 0.0], jac=True, control=ctrl, want_vectors=False, chunk=16)
print(fit["niter"], fit["loop_ms"])
prob.close()
