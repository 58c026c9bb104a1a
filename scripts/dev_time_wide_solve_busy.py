"""developer probe: is the one-wavefront solve kernel slow because the chip idles at a low clock?  Time it alone and
while a side stream keeps the other CUs busy with matrix products."""
import os, sys, time
os.environ["GSLNLS_WIDE_SOLVE_REPS"] = "300"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gslnls_amd import _lib
L = _lib.lib()
rng = np.random.default_rng(1)
p = 32
J = rng.standard_normal((4 * p, p)); A = J.T @ J
Ap = np.ascontiguousarray(np.concatenate([A[i, :i + 1] for i in range(p)]))
diag = np.sqrt(np.diag(A)); rhs = rng.standard_normal(p); sol = np.zeros(p)
def solve():
    L.gslnls_debug_wide_solve(p, Ap.ctypes.data_as(_lib.DP), diag.ctypes.data_as(_lib.DP), 1e-3, rhs.ctypes.data_as(_lib.DP), sol.ctypes.data_as(_lib.DP))
print("idle chip:", flush=True); solve()
side = torch.cuda.Stream()
a = torch.randn(8192, 8192, device="cuda", dtype=torch.float64); b = torch.randn(8192, 8192, device="cuda", dtype=torch.float64)
with torch.cuda.stream(side):
    for _ in range(40):
        c = a @ b
time.sleep(0.05)
print("busy chip (fp64 matmuls on a side stream):", flush=True); solve()
torch.cuda.synchronize()
print("idle again:", flush=True); solve()
