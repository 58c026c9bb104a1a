"""developer timing: the sparse CG product J^T (J u) of the large path on a big banded-random Jacobian"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, scipy.sparse as sp
from gslnls_amd.nls_large import SparseLargeProblem

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
k = int(sys.argv[3]) if len(sys.argv) > 3 else 16
rng = np.random.Generator(np.random.PCG64(3))
rows = np.repeat(np.arange(n, dtype=np.int64), k)
# k entries per row clustered around the row's "home" column (locality like a discretised operator)
home = (np.arange(n, dtype=np.int64) * p) // n
cols = (home[:, None] + rng.integers(-64, 65, size=(n, k))) % p
A = sp.csr_matrix((rng.standard_normal(n * k), (rows, cols.reshape(-1))), shape=(n, p))
A.sum_duplicates()
nnz = A.nnz
y = np.zeros(n)
t0 = time.time()
prob = SparseLargeProblem(lambda th: A @ th, lambda th: A, y, p)
x = rng.standard_normal(p)
ms = prob.time_pass(1, x, x, reps=20)
# one product pair: CSR pass (value 8 + column 4 per entry, w written 8n) + CSC pass (value 8 + row 4 per entry,
# w gathered 8n) + pointers; the p-vectors are cache resident.  The time includes the p-vector PCIe hops.
bytes_pair = nnz * 24 + n * 16 + (n + p) * 4
print("n=%d p=%d nnz=%d: J^T(J u) %.3f ms -> %.0f GB/s (%.1f MB algorithmic per product pair)" % (
    n, p, nnz, ms, bytes_pair / (ms * 1e-3) / 1e9, bytes_pair / 1e6))
prob.close()
