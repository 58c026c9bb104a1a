"""developer aid: records of the first 8192 multi-start points, small batch (2 or 4 lanes per fit) vs a batch of 40000 (one lane)"""
import sys, os, ctypes as C
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, json
import gslnls_amd as amd
from gslnls_amd import _lib
from gslnls_amd.control import gsl_nls_control, pack_control
x = np.array([1., 2., 3., 5., 7., 10.]); y = np.array([109., 149., 149., 191., 213., 224.])
ci, cd = pack_control(gsl_nls_control(solver="cholesky"), "lm")
prob = amd.DenseProblem(2, 2, x, y)
ranges = np.array([1.0, 500.0, 0.01, 5.0]); kd = np.array([0.75, 0.75])
K = _lib.lib().gslnls_mstart_record_size(2)
import sys
JAC = int(sys.argv[1]) if len(sys.argv) > 1 else 1
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
out = {}
for N in (NB, 40000):
    rec = np.zeros((N, K)); ms = C.c_float(0)
    _lib.lib().gslnls_mstart_batch(prob._h, JAC, ranges.ctypes.data_as(_lib.DP), kd.ctypes.data_as(_lib.DP), 0, N, 0, N, 5, 1e-6, ci.ctypes.data_as(_lib.IP), cd.ctypes.data_as(_lib.DP), None, rec.ctypes.data_as(C.c_void_p), 0, C.byref(ms))
    out[N] = rec[:8192].copy()
a, b = out[NB], out[40000]
sel = (b[:, 8] > 1e-6) & np.isfinite(b[:, 7])
relp = np.max(np.abs(a[sel, 0:2] - b[sel, 0:2]) / np.maximum(np.abs(b[sel, 0:2]), 1e-300), axis=1)
rels = np.abs(a[sel, 7] - b[sel, 7]) / np.abs(b[sel, 7])
print("par rel: max %.3g, >1e-6: %d, >1e-8: %d of %d" % (relp.max(), (relp > 1e-6).sum(), (relp > 1e-8).sum(), sel.sum()))
print("ssr rel: max %.3g, >1e-8: %d" % (rels.max(), (rels > 1e-8).sum()))
w = np.argsort(-relp)[:5]
for i in w: print(a[sel][i, :2], b[sel][i, :2], a[sel][i, 7], b[sel][i, 7], a[sel][i, 11], a[sel][i,12])
