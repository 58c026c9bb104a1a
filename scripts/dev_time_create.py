"""developer aid: cost of creating / destroying a resident problem (allocation share of a one-shot gsl_nls call)"""
import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gslnls_amd import _lib
import gslnls_amd as A
L = _lib.lib()
X = np.asfortranarray(np.array([1., 2., 3., 5., 7., 10.]).reshape(6, 1)); y = np.array([109., 149., 149., 191., 213., 224.])
model = _lib.Model(2, 2, 1, X.ctypes.data_as(C.c_void_p), 0)
err = C.c_int(0)
for rep in range(4):
    t0 = time.perf_counter()
    h = L.gslnls_dense_create(C.byref(model), y.ctypes.data_as(C.c_void_p), 6, None, C.byref(err))
    t1 = time.perf_counter()
    L.gslnls_dense_destroy(h)
    t2 = time.perf_counter()
    print("create %.3f ms destroy %.3f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t1)))
d = dict(x=X[:, 0].copy(), y=y)
for kw in (dict(start=dict(b1=100.0, b2=0.75)), dict(start=dict(b1=[1.0, 500.0], b2=[0.01, 5.0]), control=dict(mstart_n=8192, mstart_q=819, solver="cholesky"))):
    for rep in range(3):
        t0 = time.perf_counter()
        fit = A.gsl_nls("y ~ b1*(1-exp(-b2*x))", data=d, jac=True, **kw)
        print("gsl_nls %s: %.3f ms conv %d" % ("single" if "control" not in kw else "mstart 8192", 1e3 * (time.perf_counter() - t0), fit["conv"]))
# an expression (interpreted) model: the same one-shot call
xx = np.linspace(0.0, 3.0, 200); yy = 5.0 * np.exp(-1.5 * xx) + 1.0 + 0.01 * np.sin(37.0 * xx)
for rep in range(4):
    t0 = time.perf_counter()
    fit = A.gsl_nls("y ~ A * exp(-lam * x) + b + 0 * x", data=dict(x=xx, y=yy), start=dict(A=1.0, lam=1.0, b=0.0), lowering="vm")
    print("gsl_nls interpreted expression: %.3f ms conv %d" % (1e3 * (time.perf_counter() - t0), fit["conv"]))
# robust (IRLS) one-shot call on a registered model
for rep in range(4):
    t0 = time.perf_counter()
    fit = A.gsl_nls("y ~ A * exp(-lam * x) + b", data=dict(x=xx, y=yy), start=dict(A=1.0, lam=1.0, b=0.0), loss="huber", jac=True)
    print("gsl_nls huber IRLS: %.3f ms conv %d irls_niter %s" % (1e3 * (time.perf_counter() - t0), fit["conv"], fit.get("irls", {}).get("irls_niter")))
