"""ctypes wrapper of tests/hostsim/_hostsim.so: the device headers compiled for the host (test-only)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
DP = C.POINTER(C.c_double)
IP = C.POINTER(C.c_int)


def lib():
    global _LIB
    if _LIB is None:
        subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "hostsim")])
        _LIB = C.CDLL(os.path.join(_HERE, "hostsim", "_hostsim.so"))
    return _LIB


def _dp(a):
    return None if a is None else a.ctypes.data_as(DP)


def fit(model, p, x, y, start, ci, cd, jac=1, fvv=0, sw=None, lupars=None):
    n = len(y)
    x = np.asfortranarray(np.asarray(x, dtype=np.float64).reshape(n, -1))
    y = np.ascontiguousarray(y, dtype=np.float64)
    st = np.ascontiguousarray(start, dtype=np.float64)
    par = np.zeros(p)
    ints = np.zeros(8, dtype=np.int32)
    dbls = np.zeros(8)
    cov = np.zeros((p, p), order="F")
    mi = int(ci[0])
    sst = np.full(mi + 1, np.nan)
    pt = np.full((mi + 1, p), np.nan, order="F")
    ci = np.ascontiguousarray(ci, dtype=np.int32)
    cd = np.ascontiguousarray(cd, dtype=np.float64)
    lib().hostsim_fit(int(model), n, _dp(x), _dp(y), _dp(sw), _dp(st), _dp(lupars), ci.ctypes.data_as(IP), _dp(cd),
                      int(jac), int(fvv), _dp(par), ints.ctypes.data_as(IP), _dp(dbls), _dp(cov), _dp(sst), _dp(pt))
    k = int(ints[0])
    return dict(par=par, niter=k, conv=int(ints[1]), info=int(ints[2]),
                neval=dict(f=int(ints[3]), J=int(ints[4]), fvv=int(ints[5])), launches=int(ints[6]), ssr=dbls[0],
                ssrtol=dbls[1], chisq_init=dbls[2], mu=dbls[3], delta=dbls[4], det=dbls[5], covar=cov,
                ssrtrace=sst[:k + 1], partrace=pt[:k + 1])


def lm_solve(p, A_packed, diag, mu, rhs):
    sol = np.zeros(p)
    f = {3: lib().hostsim_lm_solve3, 8: lib().hostsim_lm_solve8}[p]
    f.argtypes = [DP, DP, C.c_double, DP, DP]
    f(_dp(np.ascontiguousarray(A_packed, dtype=np.float64)), _dp(np.ascontiguousarray(diag, dtype=np.float64)),
      float(mu), _dp(np.ascontiguousarray(rhs, dtype=np.float64)), _dp(sol))
    return sol


ALLGATHER_T = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int)


def mstart(model, p, x, y, start2p, ci, cd, has_start=None, jac=1, fvv=0, sw=None, lupars=None, rank=0, world=1,
           allgather=None, shard_buf=None, all_buf=None, cap_points=0):
    """product multi-start driver (mstart_driver.hpp) around the CPU evaluator; start2p is (2, p) [lower; upper]"""
    n = len(y)
    x = np.asfortranarray(np.asarray(x, dtype=np.float64).reshape(n, -1))
    y = np.ascontiguousarray(y, dtype=np.float64)
    st = np.ascontiguousarray(np.asarray(start2p, dtype=np.float64).T.reshape(-1))
    hs = np.ones(2 * p, dtype=np.int32) if has_start is None else np.ascontiguousarray(
        np.asarray(has_start, dtype=np.int32).T.reshape(-1))
    ci = np.ascontiguousarray(ci, dtype=np.int32)
    cd = np.ascontiguousarray(cd, dtype=np.float64)
    mpopt = np.zeros(p)
    ints = np.zeros(8, dtype=np.int32)
    dbls = np.zeros(4)
    cb = ALLGATHER_T(allgather) if allgather is not None else C.cast(None, ALLGATHER_T)
    f = lib().hostsim_mstart
    f.argtypes = [C.c_int, C.c_int, DP, DP, DP, DP, DP, IP, DP, IP, C.c_int, C.c_int, C.c_int, C.c_int, ALLGATHER_T,
                  DP, DP, C.c_longlong, DP, IP, DP]
    rc = f(int(model), n, _dp(x), _dp(y), _dp(sw), _dp(st), _dp(lupars), ci.ctypes.data_as(IP), _dp(cd),
           hs.ctypes.data_as(IP), int(jac), int(fvv), int(rank), int(world), cb, _dp(shard_buf), _dp(all_buf),
           int(cap_points), _dp(mpopt), ints.ctypes.data_as(IP), _dp(dbls))
    return dict(rc=rc, mpopt=mpopt, nsp=int(ints[0]), nwsp=int(ints[1]), iters=int(ints[2]), stop=int(ints[3]),
                total_fits=int(ints[4]), draws=int(ints[5]), ssropt=dbls[0], ssrconv=dbls[1])


def sobol(dim, count, first=0):
    out = np.zeros((count, dim))
    f = lib().hostsim_sobol
    f.argtypes = [C.c_int, C.c_int, C.c_int, DP]
    f(dim, first, count, _dp(out))
    return out


def mstart_batch_misra(x, y, ranges, kd, first_draw, count, maxiter, dtol, ci, cd, jac=1, sw=None):
    """records (count x 14) of one concentration batch of the Misra1a/BoxBOD model on the CPU evaluator"""
    n = len(y)
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    rec = np.zeros((count, 14))
    f = lib().hostsim_mstart_batch
    f.argtypes = [C.c_int, DP, DP, DP, DP, DP, C.c_longlong, C.c_int, C.c_int, C.c_double, IP, DP, C.c_int, DP]
    ci = np.ascontiguousarray(ci, dtype=np.int32)
    cd = np.ascontiguousarray(cd, dtype=np.float64)
    f(n, _dp(x), _dp(y), _dp(sw), _dp(np.ascontiguousarray(ranges, dtype=np.float64)),
      _dp(np.ascontiguousarray(kd, dtype=np.float64)), int(first_draw), int(count), int(maxiter), float(dtol),
      ci.ctypes.data_as(IP), _dp(cd), int(jac), _dp(rec))
    return rec


def run_batch_comm(x, y, count, rank, world, ci, cd, want_records=True, failing_peer=False):
    """one concentration batch through ms_run_batch's callback branch (world ranks simulated in this process: the
    all-gather callback leaves the other ranks' blocks zero, or marks the peer's shard failed).  Returns (rc, records)"""
    n = len(y)
    x = np.asfortranarray(np.asarray(x, dtype=np.float64).reshape(n, -1))
    y = np.ascontiguousarray(y, dtype=np.float64)
    K = 3 * 2 + 8
    per = (count + world - 1) // world
    shard = np.zeros(per * K)
    allb = np.zeros(world * per * K)
    calls = []
    CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int)

    def _ag(_ctx, per_points, k):
        calls.append((per_points, k))
        allb[:] = 0.0
        allb[rank * per_points * k:(rank + 1) * per_points * k] = shard[:per_points * k]
        if failing_peer:
            peer = (rank + 1) % world
            allb[peer * per_points * k + 3 * 2 + 6] = -424242.0   # MS_SHARD_FAILED in the peer's first record
        return 0
    cb = CB(_ag)
    rec = np.zeros((count, K))
    rc = lib().hostsim_run_batch_comm(n, _dp(x), _dp(y), count, rank, world, cb, _dp(shard), _dp(allb),
                                      C.c_longlong(world * per), int(want_records), ci.ctypes.data_as(IP), _dp(cd),
                                      _dp(rec) if want_records else None)
    return rc, rec, calls


def psi(rho, cc, x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    c3 = np.zeros(3)
    c3[:len(cc)] = cc
    a, b = np.zeros_like(x), np.zeros_like(x)
    f = lib().hostsim_psi
    f.argtypes = [C.c_int, DP, C.c_int, DP, DP, DP]
    f(int(rho), _dp(c3), len(x), _dp(x), _dp(a), _dp(b))
    return a, b


def expr_eval(rhs, parnames, varnames, theta, X, direction=None):
    """compile the formula RHS to the device program and interpret it on the host: (value[n], grad[n, p], stats);
    with `direction` the stats also carry fvv[n] = D^2 f[v, v]"""
    X = np.asfortranarray(np.asarray(X, dtype=np.float64).reshape(-1, max(1, len(varnames))))
    n, p = X.shape[0], len(parnames)
    pn = (C.c_char_p * p)(*[s.encode() for s in parnames])
    vn = (C.c_char_p * len(varnames))(*[s.encode() for s in varnames])
    th = np.ascontiguousarray(theta, dtype=np.float64)
    val = np.zeros(n)
    grad = np.zeros((n, p), order="F")
    stats = np.zeros(4, dtype=np.int32)
    f = lib().hostsim_expr_eval
    f.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_char_p), C.c_int, C.POINTER(C.c_char_p), DP, C.c_int, DP, DP, DP, IP,
                  DP, DP]
    dv = None if direction is None else np.ascontiguousarray(direction, dtype=np.float64)
    fvv = np.zeros(n)
    rc = f(rhs.encode(), p, pn, len(varnames), vn, _dp(th), n, _dp(X), _dp(val), _dp(grad), stats.ctypes.data_as(IP),
           _dp(dv), _dp(fvv))
    if rc != 0:
        raise ValueError("expression did not compile: %s" % rhs)
    return val, grad, dict(nops=int(stats[0]), nvalue=int(stats[1]), nconst=int(stats[2]), nfvv=int(stats[3]), fvv=fvv)
