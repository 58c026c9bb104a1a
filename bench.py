#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native NLS trust-region path.

Workload (BASELINE.json configs[1], "C2"): exponential model y = A exp(-lam x) + b,
n = 1e6 observations, p = 3, fp64 Levenberg-Marquardt on the normal equations
(solver = cholesky, More' scaling, analytic Jacobian), x_i = 3(i-1)/(n-1), truth (5, 1.5, 1),
noise 0.25 N(0,1) from numpy PCG64(20250927 + rank), start (1, 1, 0), xtol = gtol = 1.49e-8
(SURVEY.md 8(d)).  A "step" is ONE complete fit: device-built start state -> init pass ->
trial steps until the device's own convergence test ends the fit.  Data are resident in HBM
before the timed region; nothing n-sized crosses PCIe inside it.

    value = LM iterations (niter summed over all steps and ranks) / wall seconds

N > 1: one process per GPU -- `python bench.py --gpus N` starts the N rank processes itself when no launcher
did (torch.distributed, backend nccl == RCCL, carries the barriers and the max-over-ranks timing); a single
large fit does not shard at BASELINE sizes, so the ranks run independent replicas ("weak"),
bracketed by barriers, time = max over ranks.  The sharded multi-start path (C4: one ncclAllGather per batch,
issued by libgslnls_hip.so itself) and batched robust fits (C5: one final all-gather) are reported, summed over
the ranks, in the "multistart" and "batched_irls" objects of the same JSON line.

Extra objects: "roofline" (dominant kernel lm_step_kernel: algorithmic bytes 16 n per launch,
duration from HIP events on the library's stream), "cpu_baseline" (the oracle, single core); at N = 1 also
"large_cgst" (C3) and "large_sparse_readme" (the reference's own sparse example, README Example 4).
"""
import argparse
import ctypes as C
import gc
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_OBS = 1_000_000
START = np.array([1.0, 1.0, 0.0])
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def c2_data(n, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    x = 3.0 * np.arange(n, dtype=np.float64) / (n - 1)
    y = 5.0 * np.exp(-1.5 * x) + 1.0 + 0.25 * rng.standard_normal(n)
    return x, y


def cpu_baseline(x, y, budget_s=20.0, max_fits=64):
    """The oracle (oracle/, plain C, J materialised row-major like GSL, solver=cholesky) on the same
    workload, one host core.  Test infrastructure used here only as the reported baseline."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gslref
    ctrl = gslref.control(solver="cholesky", xtol=1.49e-8, gtol=1.49e-8)
    rd = dict(model=gslref.MODEL_EXPDECAY, x=x, y=y)
    fits, iters, t0 = 0, 0, time.perf_counter()
    while True:
        out = gslref.nls(len(y), 3, START, rowdata=rd, use_jac=True, ctrl=ctrl)
        fits += 1
        iters += out["niter"]
        el = time.perf_counter() - t0
        if el > budget_s or fits >= max_fits:
            break
    return dict(value=iters / el, unit="LM iterations/s", cores=1, kind="port",
                sample="%d complete fits of the same n=1e6,p=3 problem (%d iterations) in %.1f s, "
                       "oracle/libgslref.so single thread" % (fits, iters, el),
                niter_per_fit=out["niter"], par=[float(v) for v in out["par"]])


def _cpu_worker(args):
    n, seed, budget_s = args
    x, y = c2_data(n, seed)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gslref
    ctrl = gslref.control(solver="cholesky", xtol=1.49e-8, gtol=1.49e-8)
    rd = dict(model=gslref.MODEL_EXPDECAY, x=x, y=y)
    iters, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        iters += gslref.nls(n, 3, START, rowdata=rd, use_jac=True, ctrl=ctrl)["niter"]
    return iters, time.perf_counter() - t0


def cpu_baseline_allcores(n, seed, budget_s=8.0):
    """All host cores: the reference is single-threaded (src/Makevars.in links no OpenMP), so the all-core figure is
    `cores` independent fits of the same problem side by side -- aggregate iterations/s, cores stated."""
    import multiprocessing as mp
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    with mp.get_context("spawn").Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(n, seed, budget_s)] * cores)
    return dict(value=sum(r[0] / r[1] for r in res), unit="LM iterations/s", cores=cores, kind="port",
                sample="%d processes x complete fits of the n=1e6,p=3 problem for %.0f s each (aggregate)" % (cores, budget_s))


BOXBOD_X = [1.0, 2.0, 3.0, 5.0, 7.0, 10.0]          # NIST BoxBOD (R/nls_test.R:790-791)
BOXBOD_Y = [109.0, 149.0, 149.0, 191.0, 213.0, 224.0]


def bind_library_comm(L, torch, dist, rank, world, backend):
    """Bootstrap of the in-library RCCL communicator (include/gslnls_core.h): rank 0 makes the 128-byte id, the job's
    existing process group carries it to the others (bootstrap only -- the all-gathers of the data path are issued by
    libgslnls_hip.so itself on its own stream).  Returns a label of what the data path uses."""
    idbuf = C.create_string_buffer(128)
    ok = torch.zeros(1, dtype=torch.int32)
    # EVERY rank makes an id (only rank 0's is used): that binds RCCL (dlopen) on each of them, so that a rank without
    # a usable RCCL is known before any rank enters ncclCommInitRank, which is itself a collective
    ok[0] = 1 if L.gslnls_comm_get_unique_id(idbuf) == 0 else 0
    t = torch.frombuffer(bytearray(idbuf.raw), dtype=torch.uint8).clone()
    dev = "cuda" if backend == "nccl" else "cpu"
    t, okd = t.to(dev), ok.to(dev)
    dist.broadcast(t, 0)
    dist.all_reduce(okd, op=dist.ReduceOp.MIN)
    if int(okd.item()) != 1:
        return None, "RCCL unavailable on some rank: %s" % L.gslnls_comm_last_error().decode()
    rc = L.gslnls_comm_init_rank(bytes(t.cpu().numpy().tobytes()), rank, world)
    flag = torch.tensor([1 if rc == 0 else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) != 1:
        L.gslnls_comm_destroy()
        return None, "ncclCommInitRank failed on some rank: %s" % L.gslnls_comm_last_error().decode()
    return True, "ncclAllGather issued by libgslnls_hip.so on its own stream (RCCL bound with dlopen)"


def multistart_bench(L, _lib, job, steps, warmup, lib_comm):
    """C4: concentration fits/sec of multi-start (src/nls_mstart.c:42-128) on NIST BoxBOD, Sobol starts in
    b1 in [1,500], b2 in [0.01,5], mstart_p = 5 LM iterations each, analytic Jacobian; the points of a batch are
    sharded over the ranks in contiguous blocks and completed with ONE all-gather of the records (RCCL)."""
    from gslnls_amd.control import gsl_nls_control, pack_control
    torch, dist, rank, world = job.torch, job.dist, job.rank, job.world
    x = np.asfortranarray(np.array(BOXBOD_X).reshape(6, 1))
    y = np.array(BOXBOD_Y)
    model = _lib.Model(2, 2, 1, x.ctypes.data_as(C.c_void_p), 0)
    err = C.c_int(0)
    h = L.gslnls_dense_create(C.byref(model), y.ctypes.data_as(C.c_void_p), 6, None, C.byref(err))
    job.barrier(bool(h), "multistart: gslnls_dense_create")
    ci, cd = pack_control(gsl_nls_control(solver="cholesky"), "lm")
    ranges = np.array([1.0, 500.0, 0.01, 5.0])
    kd = np.array([0.75, 0.75])
    K = L.gslnls_mstart_record_size(2)
    out = {}
    for label, total in (("strong_8192_total", 8192), ("weak_65536_per_gpu", 65536 * world)):
        per = (total + world - 1) // world
        lo, hi = min(total, rank * per), min(total, rank * per + per)
        use_lib = world == 1 or lib_comm
        shard = allb = None
        if not use_lib:
            # fallback: the records are gathered by torch.distributed on torch's stream
            shard = torch.zeros(per * K, dtype=torch.float64, device="cuda")
            allb = torch.zeros(world * per * K, dtype=torch.float64, device="cuda")
        ms = C.c_float(0)
        kms = []
        rec_host = np.zeros((total, K))

        # argument pointers built once (numpy's .ctypes.data_as costs ~1 us per call: harness, not the path)
        rg_p, kd_p, ci_p, cd_p = (ranges.ctypes.data_as(_lib.DP), kd.ctypes.data_as(_lib.DP), ci.ctypes.data_as(_lib.IP),
                                  cd.ctypes.data_as(_lib.DP))
        out_p = C.c_void_p(shard.data_ptr()) if shard is not None else None
        host_p, ms_p = rec_host.ctypes.data_as(C.c_void_p), C.byref(ms)

        state = {"rc": 0}

        def step(to_host=False):
            # a failed batch is carried to the next sync point, never raised between two collectives.  (The library
            # fails together: a rank whose shard failed still enters the all-gather, every rank gets the error.)
            if state["rc"]:
                return
            if use_lib:
                # lo = -1: this rank's block + the library's own all-gather; records stay in HBM unless asked for
                rc = L.gslnls_mstart_batch(h, 1, rg_p, kd_p, 0, total, -1, 0, 5, 1e-6, ci_p, cd_p, None,
                                           host_p if to_host else None, 0, ms_p)
            else:
                rc = L.gslnls_mstart_batch(h, 1, rg_p, kd_p, 0, total, lo, hi, 5, 1e-6, ci_p, cd_p, None, out_p, 1, ms_p)
            if rc != 0:
                state["rc"] = rc
                return
            kms.append(ms.value)
            if not use_lib:
                dist.all_gather_into_tensor(allb, shard)

        def timed(to_host):
            for _ in range(warmup):
                step(to_host)
            job.barrier(state["rc"] == 0, "multistart %s warmup (rc %d)" % (label, state["rc"]))
            kms.clear()
            t0 = time.perf_counter()
            for _ in range(steps):
                step(to_host)
            ok = job.sync(state["rc"] == 0)
            el = time.perf_counter() - t0
            if not ok:
                raise LegFailed("multistart %s timed batches (rc %d on this rank)" % (label, state["rc"]))
            return job.max_over_ranks(el)
        n0 = L.gslnls_comm_allgather_count()
        el = timed(False)
        n_coll = L.gslnls_comm_allgather_count() - n0
        kernel_ms = float(np.mean(kms))
        el_host = timed(True) if use_lib else None
        coll_us = None
        if use_lib and world > 1:
            # device time of the exchange alone: HIP-event pair around the library's ncclAllGather, a few extra batches
            L.gslnls_comm_set_timing(1)
            for _ in range(max(5, steps)):
                step(False)
            job.barrier(state["rc"] == 0, "multistart %s all-gather timing" % label)
            nt = C.c_longlong(0)
            tot_ms = L.gslnls_comm_allgather_ms(C.byref(nt))
            L.gslnls_comm_set_timing(0)
            coll_us = 1e3 * tot_ms / nt.value if nt.value else None
        if use_lib:
            rec = rec_host
        else:
            rec = allb[:total * K].view(total, K).cpu().numpy()
        fitted = rec[:, 3 * 2 + 2] > 1e-6
        good = int(np.sum(np.abs(rec[fitted, 3 * 2 + 1] - 1168.0088766) < 1.0))
        # flops by SURVEY.md 8(d): (p + 2) model sweeps x n = 6 rows x (1 exp + ~12 flop) per LM iteration; an fp64 exp
        # is ~25 flop in this library (devmath.hpp): 4 x 6 x 37 = 888 flop per iteration, 5 iterations per fit
        flop_fit = (2 + 2) * 6 * (25 + 12) * 5
        out[label] = {"fits_per_s": total * steps / el, "points_per_batch": total, "ms_per_batch": el / steps * 1e3,
                      "kernel_ms_per_batch_rank0": kernel_ms, "kernel_us_per_batch_rank0": 1e3 * kernel_ms,
                      "allgather_us_per_batch_rank0": coll_us, "points_per_rank": hi - lo,
                      "points_passing_det_filter": int(fitted.sum()),
                      "points_in_global_basin_after_5_iters": good,
                      "allgathers_by_library_in_timed_region": int(n_coll),
                      "fits_per_s_records_on_host": (total * steps / el_host) if el_host else None,
                      "achieved_GFLOPs_kernel": (hi - lo) * flop_fit / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else None}
    L.gslnls_dense_destroy(h)
    # counters of the SAME kernel at the SAME batch size from the committed profile (one rocprofv3 --pmc capture per
    # batch size, scripts/profile_c4c5_pmc.sh): share of the SIMD issue slots the kernel's vector instructions take
    try:
        pfile = next(os.path.join(ROOT, "profiles", f) for f in ("r03_c4c5_pmc.json", "r02_c4c5_pmc.json")
                     if os.path.exists(os.path.join(ROOT, "profiles", f)))
        with open(pfile) as f:
            pm = json.load(f)
        for label, pts in (("strong_8192_total", 8192), ("weak_65536_per_gpu", 65536)):
            k = next((v for kk, v in pm.items() if kk.startswith("ms_fit_kernel") and kk.endswith("@ %d points per dispatch" % pts)), None)
            if k and label in out:
                valu, gui = k["SQ_INSTS_VALU"]["median_per_dispatch"], k["GRBM_GUI_ACTIVE"]["median_per_dispatch"]
                out[label]["valu_issue_share"] = valu * 4.0 / (1024.0 * gui / 8.0)
                out[label]["valu_issue_share_source"] = ("committed profile profiles/%s, %d points per dispatch (SQ_INSTS_VALU x 4 "
                                                         "cycles / (1024 SIMDs x GRBM_GUI_ACTIVE per XCD)), NOT measured in this "
                                                         "run" % (os.path.basename(pfile), pts))
    except Exception:  # noqa
        pass
    if world == 1:
        # the whole multi-start procedure of C4 (sampling, concentration, reduction, local searches, final solve)
        import gslnls_amd as A
        d = dict(x=np.array(BOXBOD_X), y=np.array(BOXBOD_Y))
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            fit = A.gsl_nls("y ~ b1*(1-exp(-b2*x))", data=d, start=dict(b1=[1.0, 500.0], b2=[0.01, 5.0]), jac=True,
                            control=dict(mstart_n=8192, mstart_q=819, solver="cholesky"))
            ts.append(time.perf_counter() - t0)
        out["full_multistart_fit_8192"] = {"wall_ms": 1e3 * min(ts), "par": [float(v) for v in fit["par"]],
                                           "ssr": float(fit["ssr"]), "conv": int(fit["conv"]),
                                           "target": [213.80940889, 0.54723748542, 1168.0088766]}
    out["metric"] = "multi-start concentration fits/s (BoxBOD n=6 p=2, 5 LM iterations each, all-gather of records)"
    out["collective"] = ("none (one rank)" if world == 1 else
                         (lib_comm if isinstance(lib_comm, str) else "torch.distributed all_gather_into_tensor (fallback)"))
    if world > 1:
        out["scaling_note"] = ("strong_8192_total is LATENCY-bound by construction: 8192 points are one 35-40 us kernel on one "
                               "GPU (its duration is the slowest lane's walk through its trials, not the number of points), so "
                               "%d ranks x %d points each take about as long per batch, plus the all-gather "
                               "(allgather_us_per_batch_rank0) and one host wait: expect flat or negative strong scaling. "
                               "weak_65536_per_gpu (65536 points per rank, one all-gather of the %d x 14 doubles) is the "
                               "throughput figure." % (world, (8192 + world - 1) // world, 65536 * world))
    out["note"] = ("working set is 96 B per fit: bound by fp64 VALU + exp latency and by launch/collective latency, "
                   "an HBM fraction is not meaningful (SURVEY.md 8(d)); fits_per_s leaves the gathered records in HBM, "
                   "fits_per_s_records_on_host adds the D2H the host-side commit of the real procedure needs")
    return out


def large_bench(L, _lib, n=10_000_000, p=64):
    """C3: gsl_nls_large(cgst), synthetic GLM f_i = exp(a_i . theta), A n x p row-major fp64 resident in HBM
    (SURVEY.md 8(d)): matrix-free pass GB/s (algorithmic 8np + 16n bytes per pass) and one whole fit."""
    from gslnls_amd.nls_large import LargeProblem
    rng = np.random.Generator(np.random.PCG64(20250928))
    A = rng.uniform(-1.0, 1.0, size=(n, p))
    A /= np.sqrt(p)
    th = 0.25 * rng.standard_normal(p)
    y = np.exp(A @ th) * (1.0 + 0.01 * rng.standard_normal(n))
    prob = LargeProblem(5, p, A, y)
    del A
    x0 = np.zeros(p)
    u = rng.standard_normal(p)
    byt = 8.0 * n * p + 16.0 * n
    ms_eval = prob.time_pass(0, x0, u, reps=10)
    ms_jtjv = prob.time_pass(1, x0, u, reps=10)
    prob.time_pass(2, x0, u, reps=1)  # first call allocates the partial sums
    ms_jtj = prob.time_pass(2, x0, u, reps=10)
    t0 = time.perf_counter()
    fit = prob.solve(x0, "cgst", want_resid=False)
    el = time.perf_counter() - t0
    prob.close()
    return {"workload": "C3: gsl_nls_large cgst, GLM exp(A theta), n=%d p=%d (%.2f GB)" % (n, p, 8.0 * n * p / 1e9),
            "bytes_per_pass": byt, "eval_pass_ms": ms_eval, "eval_pass_GBs": byt / ms_eval / 1e6,
            "jtju_pass_ms": ms_jtjv, "jtju_pass_GBs": byt / ms_jtjv / 1e6,
            "roofline_frac_jtju": byt / ms_jtjv / 1e6 / HBM_PEAK_GBS,
            "full_jtj": {"ms_host_clock": ms_jtj, "what": "J^T J for the lm variant: v_mfma_f64_16x16x4 SYRK kernel + reduction of "
                         "the workgroup partials + 32 KB read-back; the kernel alone (glm_jtj_mfma_kernel<64,256>) is in the "
                         "profiles/ kernel tables (r02: 1.06-1.17 ms = 43.6-48 TFLOP/s executed on the 10 lower-triangle "
                         "blocks = 0.61-0.67 of the 72 TFLOP/s this instruction reaches back to back on the same device, "
                         "scripts/mfma_probe/overlap.hip -- rounds 2-4 quoted 47.7 for that rate: a probe artifact, DESIGN.md "
                         "round 5; it streams A at 4.4-4.8 TB/s)",
                         "effective_TFLOPs_2np2": 2.0 * n * p * p / ms_jtj / 1e9},
            "fit": {"niter": int(fit["niter"]), "conv": int(fit["conv"]), "passes": int(fit["n_passes"]),
                    "wall_s": el, "outer_iterations_per_s": fit["niter"] / el, "ssr": float(fit["ssr"]),
                    "max_abs_err_vs_truth": float(np.max(np.abs(fit["par"] - th)))}}


def end_to_end_bench(L, _lib, x, y, n, ci_p, cd_p, reps=10):
    """What .Call(C_nls) delivers at C2 ("also report end-to-end"): ONE gslnls_nls() on host buffers --
    create / re-bind, H2D of x and y (16 MB, pageable memory as R vectors are), the fit, the finalize kernel, D2H of resid +
    grad (32 MB) + covar, destroy / park -- median wall clock of `reps` calls after warm-up, with the library's own
    breakdown (gslnls_last_call_profile), for the analytic and the forward-difference Jacobian.  Two kinds of result
    buffers: pages the process has touched before (a recycled allocation) and a fresh anonymous mapping per call (what a
    large Rf_allocVector is): a device-to-host copy into untouched pages pays the operating system's page faults.
    The PCIe floor beside it: the same bytes through hipMemcpy on pinned buffers, measured here."""
    import mmap
    import torch
    names = ("create", "h2d", "loop", "finalize", "d2h", "destroy", "total")
    start = (C.c_double * 3)(1.0, 1.0, 0.0)
    out = {"workload": "one gslnls_nls() at C2 on host buffers, resid + grad + covar returned (n = %d, p = 3)" % n,
           "bytes_in": 16 * n, "bytes_out": 32 * n + 72}
    # the link, measured: pinned host memory both ways
    hp = torch.empty(4 * n, dtype=torch.float64).pin_memory()
    dv = torch.empty(4 * n, dtype=torch.float64, device="cuda")
    link = {}
    for name, nd, src, dst in (("h2d_16MB", 2 * n, hp, dv), ("d2h_32MB", 4 * n, dv, hp)):
        ts = []
        for _ in range(7):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dst[:nd].copy_(src[:nd], non_blocking=True)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        link[name + "_ms"] = 1e3 * float(np.median(ts))
    link["sum_ms"] = link["h2d_16MB_ms"] + link["d2h_32MB_ms"]
    out["pcie_floor"] = link
    del hp, dv
    for jac in (1, 0):
        for fresh in (False, True):
            ts, profs, last = [], [], None
            for rep in range(reps + 3):
                model = _lib.Model(1, 3, 1, x.ctypes.data_as(C.c_void_p), 0)
                par, covar, res = np.empty(3), np.empty(9), _lib.Result()
                if fresh:
                    m1, m2 = mmap.mmap(-1, 8 * n), mmap.mmap(-1, 24 * n)
                    resid = np.frombuffer(m1, dtype=np.float64)
                    grad = np.frombuffer(m2, dtype=np.float64)
                else:
                    resid, grad = np.empty(n), np.empty(3 * n)
                    if rep == 0:
                        resid[:] = 0.0
                        grad[:] = 0.0
                res.par, res.covar = par.ctypes.data_as(_lib.DP), covar.ctypes.data_as(_lib.DP)
                res.resid, res.grad = resid.ctypes.data_as(_lib.DP), grad.ctypes.data_as(_lib.DP)
                t0 = time.perf_counter()
                rc = L.gslnls_nls(C.byref(model), y.ctypes.data_as(C.c_void_p), n, jac, 0, start, 0, None, 0, None, ci_p, cd_p,
                                  None, 0, None, C.byref(res))
                ts.append(1e3 * (time.perf_counter() - t0))
                if rc != 0:
                    raise RuntimeError("gslnls_nls failed with %d" % rc)
                pr = (C.c_double * 8)()
                L.gslnls_last_call_profile(pr, 8)
                profs.append(list(pr)[:7])
                last = (par.copy(), float(res.ssr), int(res.niter), float(resid[n // 2]), float(grad[2 * n + 5]))
                del resid, grad
            med = np.median(np.array(profs[3:]), axis=0)
            key = ("analytic" if jac else "forward_fd") + ("_fresh_pages" if fresh else "")
            out[key] = {"ms": float(np.median(ts[3:])), "min_ms": float(np.min(ts[3:])), "calls": reps,
                        "breakdown_ms": {k: float(v) for k, v in zip(names, med)},
                        "niter": last[2], "ssr": last[1], "par": [float(v) for v in last[0]],
                        "x_pcie_floor": float(np.median(ts[3:])) / link["sum_ms"]}
    out["note"] = ("ms = wall clock of the call as the caller sees it; breakdown from inside the library (create = allocation "
                   "or re-binding of the parked problem, loop = the whole solve incl. launch overhead).  *_fresh_pages: resid "
                   "and grad are fresh anonymous mappings (never touched), so the copy engine's writes fault every page in -- an "
                   "operating-system cost a CPU implementation filling the same vectors pays as well (memcpy into untouched "
                   "pages on this host: 6.9 GB/s = 4.6 ms for these 32 MB, scripts/pcie_probe/fresh_pages.hip)")
    return out


def wide_dense_bench(_lib, n=100_000, ng=10):
    """The wide dense path (10 <= p <= 64: the reference takes any p, src/nls.c:266): gsl_nls() on a sum of ten Gaussian
    peaks + a line, p = 32, n = 1e5 (and the step time at n = 1e6); analytic Jacobian from the formula compiled in
    process, J^T J accumulated in v_mfma_f64_16x16x4 tiles (csrc/wide_kernels.hpp), one trial step = three launches
    (pass, reduce, advance)."""
    import gslnls_amd as A
    rng = np.random.Generator(np.random.PCG64(20250930))
    names, terms = [], []
    for k in range(1, ng + 1):
        names += ["a%d" % k, "m%d" % k, "s%d" % k]
        terms.append("a%d*exp(-(x-m%d)^2/s%d^2)" % (k, k, k))
    names += ["c0", "c1"]
    rhs = " + ".join(terms) + " + c0 + c1*x"
    amp, mid, wid = rng.uniform(2.0, 6.0, ng), 10.0 * np.arange(ng) + rng.uniform(3.0, 7.0, ng), rng.uniform(1.2, 2.4, ng)
    truth = np.append(np.stack([amp, mid, wid], axis=1).reshape(-1), [0.5, 0.01])
    p = len(truth)
    out = {"workload": "gsl_nls() on a sum of %d Gaussians + line: p = %d, analytic Jacobian, LM normal equations" % (ng, p)}
    for nn in (n, 10 * n):
        x = np.linspace(0.0, 10.0 * ng, nn)
        m = sum(truth[3 * k] * np.exp(-(x - truth[3 * k + 1]) ** 2 / truth[3 * k + 2] ** 2) for k in range(ng))
        y = m + truth[-2] + truth[-1] * x + 0.05 * rng.standard_normal(nn)
        start = truth * (1.0 + 0.02 * np.where(np.arange(p) % 2 == 0, 1.0, -1.0))
        prob = A.DenseProblem(_lib.MODEL_EXPR, p, x, y, expr=rhs, parnames=names, xnames=["x"], lowering="jit")
        ctrl = A.gsl_nls_control(solver="cholesky")
        prob.solve(start, jac=True, control=ctrl, want_vectors=False)          # builds / loads the kernels
        t0 = time.perf_counter()
        fit = prob.solve(start, jac=True, control=ctrl, want_vectors=False)
        wall = time.perf_counter() - t0
        ms_step = prob.time_pass(start, jac=True, reps=200)
        prob.close()
        # flops of the J^T J tiles one pass executes: NQ = 3 lower-triangle 16x16 blocks x 2 * 16 * 16 per row
        nq = (p + 15) // 16 * ((p + 15) // 16 + 1) // 2
        out["n=%d" % nn] = {"niter": int(fit["niter"]), "conv": int(fit["conv"]), "steps": int(fit["n_steps"]),
                            "fit_wall_ms": wall * 1e3, "fit_loop_ms": float(fit["loop_ms"]), "ms_per_trial_step": ms_step,
                            "LM_iterations_per_s": fit["niter"] / wall, "code_path": int(fit["code_path"]),
                            "rows_per_s_one_step": nn / (ms_step * 1e-3),
                            "mfma_tile_GFLOPs_per_step": nq * 512.0 * nn / (ms_step * 1e-3) / 1e9,
                            "max_rel_err_vs_truth": float(np.max(np.abs(fit["par"] - truth) / np.abs(truth)))}
    out["note"] = ("a row costs ten fp64 exp + the 32 gradient entries (vector instructions) and 3 MFMA tiles per four rows -- on "
                   "this device fp64 MFMA and fp64 vector instructions share ONE pipe (scripts/mfma_probe/overlap.hip), so their issue "
                   "times add: 48 x 64 + ~520 x 4 clocks per 64-row tile is the floor of the pass, not the larger of the two; one trial "
                   "step = ONE launch: rows -> per-workgroup sums -> two-level in-launch reduction (arrival tickets) -> the LM step "
                   "by the workgroup that completes the totals, natural-order L D L^T of the damped system on one wavefront "
                   "(gsl_linalg_mcholesky's pivoted form when a pivot is not safely positive), a speculative solve for the "
                   "rejected case by workgroup 0 while the rows stream.  ms_per_trial_step: back-to-back steps held at the "
                   "converged point (mostly rejected steps); fit_loop_ms / steps: the real fit's mix")
    return out


def wide_multistart_bench(L, _lib, npts=4096):
    """Multi-start for 10 <= p <= 64 (src/nls_mstart.c:42-128 at p = 12): one concentration batch of 4096 Sobol points on a
    sum of four Gaussians, n = 200, 5 LM iterations each -- every point at once, one workgroup per point
    (wide_fit_kernel), against the round-3 form that fitted them one after the other through the launch-per-step chain."""
    import gslnls_amd as A
    from gslnls_amd.control import gsl_nls_control, pack_control
    rng = np.random.Generator(np.random.PCG64(20250931))
    ng, n = 4, 200
    names, terms = [], []
    for k in range(1, ng + 1):
        names += ["a%d" % k, "m%d" % k, "s%d" % k]
        terms.append("a%d*exp(-(x-m%d)^2/s%d^2)" % (k, k, k))
    rhs = " + ".join(terms)
    amp, mid, wid = rng.uniform(2.0, 6.0, ng), 10.0 * np.arange(ng) + rng.uniform(3.0, 7.0, ng), rng.uniform(1.2, 2.4, ng)
    truth = np.stack([amp, mid, wid], axis=1).reshape(-1)
    p = len(truth)
    x = np.linspace(0.0, 10.0 * ng, n)
    y = sum(truth[3 * k] * np.exp(-(x - truth[3 * k + 1]) ** 2 / truth[3 * k + 2] ** 2) for k in range(ng)) + 0.05 * rng.standard_normal(n)
    prob = A.DenseProblem(_lib.MODEL_EXPR, p, x, y, expr=rhs, parnames=names, xnames=["x"], lowering="jit")
    prob.solve(truth * 1.02, jac=True, control=A.gsl_nls_control(solver="cholesky"), want_vectors=False)  # builds the kernels
    ci, cd = pack_control(gsl_nls_control(solver="cholesky"), "lm")
    ranges = np.stack([truth * 0.8, truth * 1.2], axis=1).reshape(-1).copy()
    kd = np.full(p, 0.75)
    K = L.gslnls_mstart_record_size(p)
    out = {"workload": "one concentration batch of %d Sobol points, sum of %d Gaussians (p = %d), n = %d, 5 LM iterations each, "
                       "analytic Jacobian" % (npts, ng, p, n)}
    recs = {}
    for label, env, count in (("batch_kernel", "1", npts), ("one_after_the_other", "0", 256)):
        os.environ["GSLNLS_WIDE_MS_BATCH"] = env
        rec = np.zeros((count, K))
        ms = C.c_float(0)
        args = (prob._h, 1, ranges.ctypes.data_as(_lib.DP), kd.ctypes.data_as(_lib.DP), 0, count, 0, count, 5, 1e-6,
                ci.ctypes.data_as(_lib.IP), cd.ctypes.data_as(_lib.DP), None, rec.ctypes.data_as(C.c_void_p), 0, C.byref(ms))
        rc = L.gslnls_mstart_batch(*args)
        if rc != 0:
            raise RuntimeError("gslnls_mstart_batch (%s) failed with %d" % (label, rc))
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            L.gslnls_mstart_batch(*args)
        el = (time.perf_counter() - t0) / reps
        out[label] = {"points": count, "ms_per_batch": 1e3 * el, "us_per_point": 1e6 * el / count, "fits_per_s": count / el,
                      "points_fitted": int(np.sum(rec[:, 3 * p + 5] > 0))}
        recs[label] = rec
    os.environ.pop("GSLNLS_WIDE_MS_BATCH", None)
    out["speedup_per_point"] = out["one_after_the_other"]["us_per_point"] / out["batch_kernel"]["us_per_point"]
    a, b = recs["batch_kernel"][:256], recs["one_after_the_other"]
    out["max_rel_diff_of_the_first_256_records"] = float(np.max(np.abs(a[:, :p] - b[:, :p]) / np.maximum(np.abs(b[:, :p]), 1e-300)))
    prob.close()
    return out


def function_model_bench():
    """gsl_nls() on `function` models (gslnls_nls_fn: the closures stay on the host, csrc/bd_host.hpp): README example 4 exactly
    as the README calls it -- gsl_nls(fn = f, y = rep(0, p + 1), start = 1:p, control = list(maxiter = 500)), p = 500, the
    reference's own dense benchmark (36.66 s there on unstated hardware: context, not a baseline) -- and a p = 199 sum of
    Gaussians."""
    import gslnls_amd as A
    p = 500
    a = np.sqrt(1e-5)

    def f(th):
        return np.concatenate([a * (th - 1.0), [np.sum(th ** 2) - 0.25]]), np.vstack([a * np.eye(p), 2.0 * th[None, :]])
    out = {"workload": "README Example 4 through gsl_nls(): penalty function I, p = 500, n = 501, dense Jacobian from a Python "
                       "closure, LM", "reference_readme_quotes_s": 36.66}
    for _ in range(2):
        t0 = time.perf_counter()
        fit = A.gsl_nls(f, y=np.zeros(p + 1), start=np.arange(1.0, p + 1.0), control=dict(maxiter=500, solver="cholesky"))
        el = time.perf_counter() - t0
    out["readme_example_4"] = {"wall_ms": 1e3 * el, "niter": int(fit["niter"]), "conv": int(fit["conv"]), "ssr": float(fit["ssr"]),
                               "ssr_target": 0.004778845, "neval_f": int(fit["neval"]["f"]), "neval_J": int(fit["neval"]["J"]),
                               "code_path": int(fit["code_path"])}
    return out



def streamed_large_n_bench(L, _lib, jac):
    """lm_step_kernel where it is bound by HBM instead of by the latency of a launch: the same fused pass (x and y read once,
    16 n bytes, nothing n-sized written) at n = 4e6 and 6.4e7 (64 MB / 1 GB of data: beyond the 256 MiB Infinity Cache at
    the larger size), back-to-back launches timed with HIP events on the library's stream (gslnls_dense_time_pass)."""
    out = []
    th = np.array([4.0, 1.2, 0.8])
    for nn in (4_000_000, 64_000_000):
        x = 3.0 * np.arange(nn, dtype=np.float64) / (nn - 1)
        y = 5.0 * np.exp(-1.5 * x) + 1.0
        model = _lib.Model(1, 3, 1, x.ctypes.data_as(C.c_void_p), 0)
        err = C.c_int(0)
        h = L.gslnls_dense_create(C.byref(model), y.ctypes.data_as(C.c_void_p), nn, None, C.byref(err))
        if not h:
            out.append({"n": nn, "error": int(err.value)})
            continue
        ms = float(L.gslnls_dense_time_pass(h, jac, th.ctypes.data_as(_lib.DP), 200 if nn < 10_000_000 else 40))
        L.gslnls_dense_destroy(h)
        del x, y
        gc.collect()
        gbs = 16.0 * nn / (ms * 1e-3) / 1e9 if ms > 0 else None
        out.append({"n": nn, "bytes_per_launch": 16.0 * nn, "ms_per_launch": ms, "GBs": gbs, "frac": gbs / HBM_PEAK_GBS if gbs else None})
    L.gslnls_trim_cache()
    return out


def matrix_path_bench(_lib):
    """gsl_nls() on FORMULAS beyond 64 parameters (the matrix path, csrc/bd_host.hpp: no closure, rows by a kernel compiled for
    the formula, J^T J on the matrix cores, the damped solve on the device): sums of Gaussians with p = 99, 198, 501 -- wall
    time per trial step from the library's own profile of the call (gslnls_last_matrix_path_profile), split into the damped
    solve with the fused trial evaluation, the Jacobian + J^T J + J^T f of the accepted points and the rest; the device time
    of one J^T J (HIP events) against the measured MFMA f64 rate 72 TFLOP/s (scripts/mfma_probe/overlap.hip); the device time of one damped solve."""
    import gslnls_amd as A
    L = _lib.lib()
    out = {"workload": "formula models with p > 64: sums of ng Gaussians a*exp(-((x-m)/w)^2), p = 3 ng, analytic Jacobian, lm, "
                       "solver = cholesky; one host synchronisation per trial step (the solve's), one more per accepted point",
           "mfma_f64_measured_tflops": 72.0, "cases": []}
    for ng, n in ((33, 3000), (66, 5000), (167, 20000)):
        pp = 3 * ng
        rng = np.random.Generator(np.random.PCG64(ng))
        x = np.linspace(0.0, 10.0 * ng, n)
        amp, mid, wid = rng.uniform(2.0, 6.0, ng), 10.0 * np.arange(ng) + rng.uniform(3.0, 7.0, ng), rng.uniform(1.2, 2.4, ng)
        y = np.sum(amp * np.exp(-((x[:, None] - mid) / wid) ** 2), axis=1) + 0.01 * rng.standard_normal(n)
        rhs = " + ".join("a%d * exp(-((x - m%d) / w%d)^2)" % (g, g, g) for g in range(ng))
        start = {}
        for g in range(ng):
            start["a%d" % g], start["m%d" % g], start["w%d" % g] = 0.9 * amp[g], mid[g] + 0.15, 1.1 * wid[g]
        best = None
        for _ in range(4):  # (the first call compiles the formula's kernels or loads them from the cache)
            t0 = time.perf_counter()
            fit = A.gsl_nls("y ~ " + rhs, data=dict(x=x, y=y), start=start, jac=True, control=dict(solver="cholesky"))
            el = 1e3 * (time.perf_counter() - t0)
            prof = np.zeros(12)
            L.gslnls_last_matrix_path_profile(prof.ctypes.data_as(_lib.DP), 12)
            if best is None or prof[1] < best[1][1]:
                best = (el, prof.copy())
        el, prof = best
        steps, njac = max(1.0, prof[8]), prof[9]
        syrk_ms = float(L.gslnls_debug_bd_syrk_ms(n, pp, 20))
        # the damped solve alone, J^T J resident (HIP events around its kernels)
        J0 = rng.standard_normal((pp + 50, pp))
        Am = np.ascontiguousarray(J0.T @ J0)
        dA = C.c_void_p()
        L.gslnls_debug_device_alloc(C.byref(dA), Am.nbytes)
        L.gslnls_debug_device_copy(dA, Am.ctypes.data_as(C.c_void_p), Am.nbytes, 1)
        d, r, sol = np.sqrt(np.diag(Am)).copy(), rng.standard_normal(pp), np.zeros(pp)
        sm = []
        L.gslnls_debug_mchol_timing(1)  # (the solves' event pair: off in the product path)
        for _ in range(12):
            L.gslnls_debug_mchol_solve_resident(pp, dA, d.ctypes.data_as(_lib.DP), 1e-3, r.ctypes.data_as(_lib.DP), sol.ctypes.data_as(_lib.DP))
            sm.append(L.gslnls_debug_mchol_last_device_ms())
        L.gslnls_debug_mchol_timing(0)
        L.gslnls_debug_device_free(dA)
        out["cases"].append({
            "p": pp, "n": n, "niter": int(fit["niter"]), "conv": int(fit["conv"]), "trial_steps": int(prof[8]), "jacobians": int(njac),
            "host_syncs_per_trial_step": 1 if prof[10] else None,
            "us_per_trial_step": 1e3 * prof[1] / steps,
            "breakdown_us_per_trial_step": {"damped_solve_and_trial_evaluation": 1e3 * prof[2] / steps,
                                            "jacobian_jtj_jtf_of_accepted_points": 1e3 * prof[3] / steps,
                                            "residuals_outside_the_fused_step": 1e3 * prof[4] / steps,
                                            "host_vectors_and_decisions": 1e3 * (prof[1] - prof[2] - prof[3] - prof[4]) / steps},
            "per_call_ms": {"whole_call": el, "set_up": prof[0], "loop": prof[1], "covariance_and_condition_diagnostic": prof[5] + prof[7],
                            "resid_and_grad_to_host": prof[6]},
            "jtj_device_ms": syrk_ms, "jtj_tflops": 2.0 * n * pp * pp / (syrk_ms * 1e-3) / 1e12 / 2.0 if syrk_ms > 0 else None,
            "jtj_frac_of_measured_mfma_rate": (n * pp * pp / (syrk_ms * 1e-3) / 1e12) / 72.0 if syrk_ms > 0 else None,
            "damped_solve_device_ms": float(np.median(sm))})
    # the J^T J kernel alone at sizes the formula cases do not reach (function models go to p = 4096): 128-column blocks from
    # p = 384 and n = 2048 on (bd_syrk128_kernel), 64-column blocks below
    out["jtj_kernel"] = []
    for n, pp in ((20000, 501), (50000, 1000), (20000, 2000), (4300, 4096)):
        ms = float(L.gslnls_debug_bd_syrk_ms(n, pp, 10))
        out["jtj_kernel"].append({"n": n, "p": pp, "device_ms": ms, "tflops_n_p2": n * pp * pp / (ms * 1e-3) / 1e12 if ms > 0 else None,
                                  "frac_of_measured_mfma_rate": n * pp * pp / (ms * 1e-3) / 1e12 / 72.0 if ms > 0 else None})
    out["note"] = ("J^T J flops counted as n p^2 (the lower triangle is computed: n p (p + 1) flops of the 2 n p^2 of the full product); "
                   "trial steps include the first evaluation's share of the loop")
    return out


def sparse_readme_bench():
    """The reference's own sparse example (README.md:1040-1146: penalty function I, p = 500, Jacobian a dgCMatrix)
    through gsl_nls_large(fn, jac) -- host closures, every product with J on the device.  One call each for cgst and lm
    (the second of two: the first pays for the allocations that are parked afterwards)."""
    import scipy.sparse as sp
    import gslnls_amd as A
    p = 500
    a = np.sqrt(1e-5)
    J0 = sp.vstack([sp.identity(p, format="csr") * a, sp.csr_matrix(np.ones((1, p)))]).tocsc()
    J0.sort_indices()
    last_row = np.flatnonzero(J0.indices == p)

    def fn(th):
        return np.concatenate([a * (th - 1.0), [np.sum(th ** 2) - 0.25]])

    def jac(th):
        J0.data[last_row] = 2.0 * th  # values of a prebuilt pattern (what Matrix does for a dgCMatrix of fixed structure)
        return J0
    out = {"workload": "README Example 4: penalty function I, p = 500, n = 501, sparse Jacobian (dgCMatrix), Python closures",
           "reference_readme_quotes_ms": {"cgst": 158.0, "lm": 5670.0, "note": "README.md:1146, unstated hardware; not a baseline"}}
    for alg in ("cgst", "lm"):
        for _ in range(2):
            t0 = time.perf_counter()
            fit = A.gsl_nls_large(fn, y=np.zeros(p + 1), start=np.arange(1.0, p + 1), algorithm=alg, jac=jac,
                                  control=dict(maxiter=500))
            el = time.perf_counter() - t0
        out[alg] = {"wall_ms": el * 1e3, "niter": int(fit["niter"]), "conv": int(fit["conv"]), "ssr": float(fit["ssr"]),
                    "device_passes": int(fit["n_passes"]), "ssr_target": 0.004778845}
    # ... and the two DENSE rows of the same table (README.md:1143-1144: the Jacobian closure returns a plain matrix)
    Jd = np.zeros((p + 1, p), order="F")  # (column-major, as the R matrix of the README's closure is: it crosses as it is)
    Jd[np.arange(p), np.arange(p)] = a

    def jac_dense(th):
        Jd[p, :] = 2.0 * th
        return Jd
    out["reference_readme_quotes_ms"].update({"dense_cgst": 1320.0, "dense_lm": 7800.0})
    for alg in ("cgst", "lm"):
        for _ in range(2):
            t0 = time.perf_counter()
            fit = A.gsl_nls_large(fn, y=np.zeros(p + 1), start=np.arange(1.0, p + 1), algorithm=alg, jac=jac_dense,
                                  control=dict(maxiter=500))
            el = time.perf_counter() - t0
        out["dense_" + alg] = {"wall_ms": el * 1e3, "niter": int(fit["niter"]), "conv": int(fit["conv"]),
                               "ssr": float(fit["ssr"]), "device_passes": int(fit["n_passes"]), "ssr_target": 0.004778845}
    # the damped solve of the lm step alone (csrc/mchol_device.hip: blocked natural-order Cholesky, the pivoted modified
    # routine as its fallback): wall time per solve with the p x p matrix uploaded from the host, with it resident, and
    # the residual it leaves
    from gslnls_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(7)
    fac = {}
    for pp in (500, 1000, 2000):
        J = rng.standard_normal((2 * pp, pp))
        Aj = np.ascontiguousarray(J.T @ J)
        dg = np.sqrt(np.diag(Aj)).copy()
        rhs = rng.standard_normal(pp)
        sol = np.zeros(pp)
        args = (pp, Aj.ctypes.data_as(_lib.DP), dg.ctypes.data_as(_lib.DP), 1e-3, rhs.ctypes.data_as(_lib.DP),
                sol.ctypes.data_as(_lib.DP))
        rc = L.gslnls_debug_mchol_solve(*args)
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            rc = L.gslnls_debug_mchol_solve(*args) or rc
        el = (time.perf_counter() - t0) / reps
        M = Aj + 1e-3 * np.diag(dg ** 2)
        entry = {"rc": int(rc), "ms_per_solve_matrix_from_host": el * 1e3,
                 "rel_residual": float(np.max(np.abs(M @ sol - rhs)) / np.max(np.abs(rhs)))}
        # ... and as the lm step calls it: J^T J resident on the device (it is assembled there), diag and rhs go up (2 p
        # doubles), the solution comes down -- the upload of the p x p matrix of the line above is not part of a step
        dA = C.c_void_p()
        if L.gslnls_debug_device_alloc(C.byref(dA), Aj.nbytes) == 0:
            L.gslnls_debug_device_copy(dA, Aj.ctypes.data_as(C.c_void_p), Aj.nbytes, 1)
            solr = np.zeros(pp)
            rargs = (pp, dA, dg.ctypes.data_as(_lib.DP), 1e-3, rhs.ctypes.data_as(_lib.DP), solr.ctypes.data_as(_lib.DP))
            rc2 = L.gslnls_debug_mchol_solve_resident(*rargs)
            dev_ms, wall_ms = [], []
            for _ in range(4 * reps):
                t0 = time.perf_counter()
                rc2 = L.gslnls_debug_mchol_solve_resident(*rargs) or rc2
                wall_ms.append((time.perf_counter() - t0) * 1e3)
            L.gslnls_debug_mchol_timing(1)  # (the event pair around a solve's kernels: off in the product path and in the wall times above)
            for _ in range(2 * reps):
                rc2 = L.gslnls_debug_mchol_solve_resident(*rargs) or rc2
                dev_ms.append(L.gslnls_debug_mchol_last_device_ms())
            L.gslnls_debug_mchol_timing(0)
            # median, mean and max of the host clock around each call: in THIS process one call in twenty at p = 2000 takes tens
            # of milliseconds longer on the host while the events show the same device work; 300 consecutive solves in a
            # process of their own show no such call (scripts/dev_mchol_stalls.py: median = mean = 1.05 ms, max 1.1) -- the
            # pauses are this interpreter's (collector, the other legs' threads), not the solve's
            entry["ms_per_solve"] = float(np.median(wall_ms))
            entry["ms_per_solve_mean"] = float(np.mean(wall_ms))
            entry["ms_per_solve_max"] = float(np.max(wall_ms))
            # (HIP events around the kernels of each solve: what the device did, whatever else the host was busy with)
            entry["device_ms_per_solve"] = float(np.median(dev_ms))
            entry["rc_resident"] = int(rc2)
            entry["resident_equals_uploaded"] = bool(np.array_equal(solr, sol))
            L.gslnls_debug_device_free(dA)
        else:
            entry["ms_per_solve"] = el * 1e3
            entry["resident_error"] = "device allocation failed"
        fac["p=%d" % pp] = entry
    out["lm_step_factorisation_on_device"] = fac
    return out


def batch_irls_bench(_lib, job, B=4096, n=10000):
    """C5: B data sets x n rows, NIST Gauss1 family p = 8, 2 % outliers, loss = bisquare; data sets are
    independent, so rank r fits the contiguous block [r B/W, (r+1) B/W) with no traffic, and ONE all-gather of
    theta-hat, sigma-hat and the status words completes the result on every rank (gslnls_batch_irls_gather)"""
    from gslnls_amd.batch import BatchProblem
    rank, world = job.rank, job.world
    truth = np.array([98.778210871, 0.010497276517, 100.48990633, 67.481111276, 23.129773360, 71.994503004,
                      178.99805021, 18.389389025])
    start = np.array([97.0, 0.009, 100.0, 65.0, 20.0, 70.0, 178.0, 16.5])
    per = (B + world - 1) // world
    lo, hi = min(B, rank * per), min(B, rank * per + per)
    x = 250.0 * np.arange(1, n + 1) / n
    X = np.tile(x, (hi - lo, 1))
    Y = np.zeros((hi - lo, n))
    for d in range(lo, hi):
        rng = np.random.Generator(np.random.PCG64(20250929 + d))
        th = truth * (1.0 + 0.05 * rng.uniform(-1, 1, 8))
        yy = (th[0] * np.exp(-th[1] * x) + th[2] * np.exp(-(x - th[3]) ** 2 / th[4] ** 2)
              + th[5] * np.exp(-(x - th[6]) ** 2 / th[7] ** 2)) + 2.5 * rng.standard_normal(n)
        yy[rng.choice(n, n // 50, replace=False)] += 50.0
        Y[d - lo] = yy
    prob, out, state = None, None, {"err": None}

    def guarded(fn):
        # library failures are carried to the next sync point (gslnls_batch_irls_gather itself fails on every rank
        # together: a rank-local failure is taken through the collective)
        if state["err"] is None:
            try:
                return fn()
            except Exception as e:  # noqa
                state["err"] = repr(e)
        return None
    prob = guarded(lambda: BatchProblem(4, 8, X, Y))
    job.barrier(state["err"] is None, "batched IRLS: problem creation (%s)" % state["err"])
    kw = dict(loss="bisquare", jac=True, control=dict(solver="cholesky"))
    guarded(lambda: prob.irls_gathered(B, start, **kw))
    job.barrier(state["err"] is None, "batched IRLS warm-up (%s)" % state["err"])
    t0 = time.perf_counter()
    out = guarded(lambda: prob.irls_gathered(B, start, **kw))
    ok = job.sync(state["err"] is None)
    el = time.perf_counter() - t0
    if not ok:
        raise LegFailed("batched IRLS timed call (%s)" % state["err"])
    el = job.max_over_ranks(el)
    prob_passes = dict(prob.last_passes)
    prob.close()
    # Algorithmic bytes by SURVEY.md 8(d): the unit is one pass over a data set's rows (x, y, w = 24 B per row; an LM
    # iteration with one trial = 2 such passes in the reference, here every trial is one fused pass) + per re-weighting
    # n x 16 B read + n x 8 B written.  The kernel counts its own passes (gslnls_batch_last_passes): this rank's data sets.
    lm_passes, rws = prob_passes["lm"], prob_passes["reweight"]
    alg_bytes = lm_passes * n * 24.0 + rws * n * 24.0
    rw = float(out["irls_niter"].sum())
    kms = out["kernel_ms"]
    # counters of the same kernel on the same workload from the committed profile (separate rocprofv3 --pmc passes,
    # scripts/profile_c4c5_pmc.sh): L2-miss traffic (gfx950 correction applied) and fp64-VALU issue share
    pmc = {}
    try:
        pfile = next(os.path.join(ROOT, "profiles", f) for f in ("r03_c4c5_pmc.json", "r02_c4c5_pmc.json")
                     if os.path.exists(os.path.join(ROOT, "profiles", f)))
        with open(pfile) as f:
            k5 = next(v for k, v in json.load(f).items() if k.startswith("irls_batch_kernel"))
        valu, gui = k5["SQ_INSTS_VALU"]["median_per_dispatch"], k5["GRBM_GUI_ACTIVE"]["median_per_dispatch"]
        pmc = {"traffic": k5["hbm_bytes_per_dispatch_corrected"],
               "traffic_source": "committed profile profiles/" + os.path.basename(pfile) + " (4096 data sets per dispatch; bytes that "
                                 "missed L2 -- served by Infinity Cache or HBM), NOT measured in this run",
               "valu_wave_instructions_per_dispatch": valu,
               "valu_issue_share": valu * 4.0 / (1024.0 * gui / 8.0),
               "valu_issue_share_note": "SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE per XCD): the share "
                                        "of SIMD issue slots the kernel's vector instructions take if each took the 4 "
                                        "cycles of a wave64 fp64 FMA (divisions, exp and sqrt sequences take longer)"}
    except Exception:
        pmc = {"traffic": None}
    return {"workload": "C5: %d data sets x n=%d, Gauss1 family p=8, bisquare IRLS, %d rank(s)" % (B, n, world),
            "datasets_per_s": B / el, "irls_iterations_per_s": rw / el, "wall_ms": el * 1e3,
            "kernel_ms_rank0": kms, "converged": int((out["conv"] == 0).sum()),
            "irls_converged": int((out["irls_status"] == 0).sum()), "mean_irls_iterations": float(out["irls_niter"].mean()),
            "lm_passes_per_dataset_this_rank": lm_passes / max(1, hi - lo),
            "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                         "achieved": alg_bytes / (kms * 1e-3) / 1e9 if kms > 0 else None,
                         "frac": alg_bytes / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS if kms > 0 else None,
                         "accounting": "240 KB (n x 24 B) per pass over a data set's rows, LM passes and re-weightings as "
                                       "counted by the kernel, this rank's data sets / its kernel time.  The 240 KB of a data "
                                       "set stay in L2 / Infinity Cache for its whole fit and every row costs three fp64 exp "
                                       "and 46 accumulator FMAs: the kernel is bound by fp64 VALU issue (two workgroups per "
                                       "CU, 256 VGPRs each), not by HBM", **pmc}}


DEADLINE_S = float(os.environ.get("GSLNLS_BENCH_DEADLINE_S", "480"))


class LegFailed(RuntimeError):
    """A leg failed and every rank already knows (the failure was agreed at a sync point)."""


class Job:
    """The ranks of one bench run.  Every synchronisation point of a leg is `sync(ok)`: ONE all-reduce(MIN) of an ok
    flag (itself a barrier: no rank gets the result before every rank has contributed) + torch.cuda.synchronize(),
    followed by dist.barrier() when everybody is fine.  A rank that failed locally never raises between two
    collectives: it carries `ok = False` to the next sync point, where all ranks learn it and raise LegFailed
    together.  A rank that threw something unforeseen lands in run_leg's closing `sync(False)`, which pairs with
    whatever sync point the other ranks are waiting in (same collective, same shape) -- so no rank is left alone
    in a collective because a peer raised."""

    def __init__(self, rank, world, dist=None, torch=None, device="cpu"):
        self.rank, self.world, self.dist, self.torch, self.device = rank, world, dist, torch, device

    def cuda_sync(self):
        if self.torch is not None and self.device != "cpu":
            self.torch.cuda.synchronize()

    def sync(self, ok=True):
        """agreeing barrier; returns True iff every rank said ok"""
        if self.dist is None:
            self.cuda_sync()
            return bool(ok)
        self.cuda_sync()
        flag = self.torch.tensor([1 if ok else 0], dtype=self.torch.int32, device=self.device)
        self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN)
        agreed = int(flag.item()) == 1
        if agreed:
            self.dist.barrier()
        self.cuda_sync()
        return agreed

    def barrier(self, ok=True, what="leg"):
        if not self.sync(ok):
            raise LegFailed("%s: failed on %s" % (what, "this rank" if not ok else "another rank"))

    def max_over_ranks(self, v):
        if self.dist is None:
            return float(v)
        t = self.torch.tensor([v], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, v):
        if self.dist is None:
            return float(v)
        t = self.torch.tensor([v], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())


def run_leg(job, name, fn, *a, **kw):
    """Run one leg on every rank; returns (result, None) on all ranks or (None, error string) on all ranks."""
    try:
        out = fn(*a, **kw)
        err = None
    except LegFailed as e:          # agreed at a sync point inside the leg: every rank is here, nothing to exchange
        return None, "%s: %s" % (name, e)
    except Exception as e:          # noqa: unforeseen, this rank only -- tell the others at their next sync point
        out, err = None, "%s: %r" % (name, e)
    if job.sync(err is None):
        return out, None
    return None, err or ("%s: failed on another rank" % name)


def install_watchdog(rank):
    """No code path may block longer than the deadline: a daemon timer ends this rank (exit code 124) when the run
    has not finished by then -- whatever it is blocked in (a collective whose peer died, a hung kernel)."""
    import threading

    def _fire():
        sys.stderr.write("bench.py: rank %d still running after %.0f s (GSLNLS_BENCH_DEADLINE_S): giving up\n" % (rank, DEADLINE_S))
        sys.stderr.flush()
        os._exit(124)
    t = threading.Timer(DEADLINE_S, _fire)
    t.daemon = True
    t.start()
    return t


def fault(where, rank):
    """Test hook (GPU-free tests of the N > 1 failure paths): GSLNLS_BENCH_FAULT = "<kind>:<where>:<rank>" with kind in
    raise | fail | kill | hang, where in headline | side.  `fail` is returned to the caller (a library call that
    reported an error), the others happen here."""
    spec = os.environ.get("GSLNLS_BENCH_FAULT", "")
    if not spec:
        return False
    kind, w, r = spec.split(":")
    if w != where or int(r) != rank:
        return False
    if kind == "raise":
        raise RuntimeError("injected fault in %s on rank %d" % (where, rank))
    if kind == "kill":
        import signal
        os.kill(os.getpid(), signal.SIGKILL)
    if kind == "hang":
        time.sleep(1e6)
    return kind == "fail"


def spawn_ranks(n_ranks):
    """`python bench.py --gpus N` without a launcher: start N rank processes of this script (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in their environment, exactly what torch.distributed.run would set), relay rank 0's one
    JSON line, return the worst exit code.  The launcher owns an overall deadline (GSLNLS_BENCH_DEADLINE_S, default
    480 s): when it passes, or when one rank has died and the others have not followed within the grace period, it
    kills ITS OWN children (each in its own session, by process group id -- never by pattern), still relays what rank 0
    wrote, and exits non-zero."""
    import signal
    import socket
    import subprocess
    import threading
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, start_new_session=True))
    chunks = []
    reader = threading.Thread(target=lambda: chunks.extend(iter(lambda: procs[0].stdout.read(65536), b"")), daemon=True)
    reader.start()
    grace = float(os.environ.get("GSLNLS_BENCH_GRACE_S", "20"))
    t0 = time.monotonic()
    first_death = None
    verdict = 0
    while True:
        rcs = [p.poll() for p in procs]
        if all(rc is not None for rc in rcs):
            break
        now = time.monotonic()
        if first_death is None and any(rc not in (None, 0) for rc in rcs):
            first_death = now
        timed_out = now - t0 > DEADLINE_S
        orphaned = first_death is not None and now - first_death > grace
        if timed_out or orphaned:
            verdict = 124 if timed_out else 1
            sys.stderr.write("bench.py launcher: %s; stopping the remaining ranks\n" %
                             ("deadline of %.0f s passed" % DEADLINE_S if timed_out else
                              "a rank died (exit codes so far %s) and the others did not finish within %.0f s" % (rcs, grace)))
            for sig in (signal.SIGTERM, signal.SIGKILL):
                for p in procs:
                    if p.poll() is None:
                        try:
                            os.killpg(p.pid, sig)
                        except ProcessLookupError:
                            pass
                t1 = time.monotonic()
                while time.monotonic() - t1 < 5.0 and any(p.poll() is None for p in procs):
                    time.sleep(0.05)
            break
        time.sleep(0.05)
    for p in procs:
        try:
            p.wait(timeout=10)
        except Exception:  # noqa
            pass
    reader.join(timeout=5)
    sys.stdout.write(b"".join(chunks).decode(errors="replace"))
    sys.stdout.flush()
    worst = max([abs(p.returncode) if p.returncode is not None else 125 for p in procs] + [verdict])
    return min(worst, 125)


def plumbing_only(args):
    """The launch path and the failure handling without a GPU: rendezvous (gloo), a 'headline' leg and a 'side' leg with
    the same structure as the real ones (agreeing sync points, run_leg), one line from rank 0.  A failed headline leg
    ends every rank non-zero; a failed side leg becomes {"error": ...} in the line."""
    import datetime
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    install_watchdog(rank)
    job = Job(rank, world)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("GSLNLS_BENCH_BACKEND", "gloo"), rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=min(DEADLINE_S, 300.0)))
        job = Job(rank, world, dist, torch, "cpu")

    def headline():
        ok = not fault("headline", rank)            # a library call that reported failure: carried, not raised
        job.barrier(ok, "headline warmup")
        ranks, pids = [rank], [os.getpid()]
        if world > 1:
            mine = torch.tensor([rank, os.getpid()], dtype=torch.int64)
            allv = torch.zeros(2 * world, dtype=torch.int64)
            dist.all_gather_into_tensor(allv, mine)
            ranks, pids = [int(v) for v in allv[0::2]], [int(v) for v in allv[1::2]]
        job.barrier(True, "headline timed region")
        return {"ranks": ranks, "pids": pids}

    def side():
        ok = not fault("side", rank)
        job.barrier(ok, "side leg")
        return {"ok": True}
    head, err = run_leg(job, "headline", headline)
    if err:
        sys.stderr.write("bench.py: %s\n" % err)
        if dist.is_initialized():
            dist.destroy_process_group()
        return 1
    side_out, side_err = run_leg(job, "side", side)
    if world > 1:
        job.sync(True)
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"plumbing_only": True, "n_gpus": world, "gpus_flag": args.gpus, "ranks": head["ranks"],
                          "pids": head["pids"], "steps": args.steps, "warmup": args.warmup,
                          "side": side_out if side_err is None else {"error": side_err}}))
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--n", type=int, default=N_OBS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fd", action="store_true", help="forward finite-difference Jacobian instead of analytic")
    ap.add_argument("--chunk", type=int, default=0, help="step launches per host check; 0 = library default (adaptive)")
    ap.add_argument("--headline-only", action="store_true", help="skip the C3 / C5 side measurements")
    ap.add_argument("--plumbing-only", action="store_true",
                    help="GPU-free check of the N-rank launch: spawn, rendezvous, one all-gather of rank ids, one line")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # launched as plain `python bench.py --gpus N`: become a launcher.  Nothing in this process has touched the
        # GPU (torch is not even imported yet); the ranks are fresh child processes, never a re-exec of this one.
        raise SystemExit(spawn_ranks(args.gpus))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%s: refusing to report an N-GPU line from a different "
                         "number of ranks" % (args.gpus, os.environ.get("WORLD_SIZE")))
    if args.plumbing_only:
        raise SystemExit(plumbing_only(args))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    install_watchdog(rank)
    import torch
    dist = None
    # developer dry-run of the N > 1 code path on a 1-GPU box: GSLNLS_BENCH_BACKEND=gloo GSLNLS_BENCH_ONE_DEVICE=1
    backend = os.environ.get("GSLNLS_BENCH_BACKEND", "nccl")
    if os.environ.get("GSLNLS_BENCH_ONE_DEVICE"):
        local_rank = 0
    if world > 1:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=min(DEADLINE_S, 300.0)))
    job = Job(rank, world, dist, torch, "cuda" if (world > 1 and backend == "nccl") else ("cpu" if world > 1 else "cuda"))

    from gslnls_amd import _lib
    from gslnls_amd.control import gsl_nls_control, pack_control
    L = _lib.lib()
    have_dev = L.gslnls_device_count() >= 1 and L.gslnls_set_device(local_rank) == 0
    if not job.sync(have_dev):
        raise SystemExit("bench.py: no MI355X visible on some rank; the HIP path has no CPU fallback")

    # the interpreter's cyclic collector can pause for tens of ms once torch's object graph is loaded (seen: one
    # 38 ms step among 0.1 ms ones); the timed regions allocate nothing that needs it
    gc.collect()
    gc.disable()
    n = args.n
    x, y = c2_data(n, 20250927 + rank)
    X = np.asfortranarray(x.reshape(n, 1))
    model = _lib.Model(1, 3, 1, X.ctypes.data_as(C.c_void_p), 0)
    err = C.c_int(0)
    ctrl = gsl_nls_control(solver="cholesky", xtol=1.49e-8, gtol=1.49e-8)
    ci, cd = pack_control(ctrl, "lm")
    par = np.zeros(3)
    res = _lib.Result()
    res.par = par.ctypes.data_as(_lib.DP)
    jac = 0 if args.fd else 1
    st = START.copy()
    # argument pointers built once: numpy's .ctypes.data_as costs ~1 us per call, which is harness, not the path
    st_p, ci_p, cd_p, res_p = st.ctypes.data_as(_lib.DP), ci.ctypes.data_as(_lib.IP), cd.ctypes.data_as(_lib.DP), C.byref(res)
    solve = L.gslnls_dense_solve
    H = {}

    def headline():
        """The timed region of the contract: W untimed fits, barrier + synchronize, EXACTLY K fits, barrier + synchronize,
        max over ranks.  A failed library call is carried to the next sync point, where every rank stops together."""
        h = L.gslnls_dense_create(C.byref(model), y.ctypes.data_as(C.c_void_p), n, None, C.byref(err))
        H["h"] = h
        job.barrier(bool(h), "gslnls_dense_create (%s)" % _lib.strerror(err.value))
        bad = {"rc": 0}

        def one_fit():
            if bad["rc"]:
                return 0, 0, 0.0, 0
            rc = solve(h, jac, 0, st_p, None, ci_p, cd_p, args.chunk, res_p)
            if rc != 0:
                bad["rc"] = rc
                return 0, 0, 0.0, 0
            return res.niter, res.n_launches, res.loop_ms, res.neval[0] + res.neval[1]

        for _ in range(args.warmup):
            one_fit()
        job.barrier(bad["rc"] == 0, "warm-up fits (%s)" % _lib.strerror(bad["rc"]))
        L.gslnls_dense_loop_event_stats(h, None, None, 1)  # HIP-event totals of the launch loops: start from zero
        t0 = time.perf_counter()
        iters = launches = ref_passes = 0
        loop_ms = 0.0
        for _ in range(args.steps):
            a, b, c, d = one_fit()
            iters += a
            launches += b
            loop_ms += c
            ref_passes += d
        ok = job.sync(bad["rc"] == 0)
        elapsed = time.perf_counter() - t0
        if not ok:
            raise LegFailed("timed fits (%s on this rank)" % _lib.strerror(bad["rc"]))
        return dict(iters=iters, launches=launches, ref_passes=ref_passes, loop_ms=loop_ms, elapsed=elapsed,
                    tmax=job.max_over_ranks(elapsed), tot_iters=job.sum_over_ranks(float(iters)))
    hd, herr = run_leg(job, "headline", headline)
    if herr:
        # no valid headline: every rank leaves non-zero (they all know), nothing is printed on stdout
        sys.stderr.write("bench.py: %s\n" % herr)
        if dist is not None:
            dist.destroy_process_group()
        raise SystemExit(1)
    h = H["h"]
    iters, launches, ref_passes, loop_ms = hd["iters"], hd["launches"], hd["ref_passes"], hd["loop_ms"]
    tmax, tot_iters = hd["tmax"], hd["tot_iters"]
    ev_ms, ev_launches = C.c_double(0.0), C.c_longlong(0)
    L.gslnls_dense_loop_event_stats(h, C.byref(ev_ms), C.byref(ev_launches), 0)

    # Dominant kernel lm_step_kernel, HIP events on the library's own stream.
    # (1) over the timed region: every fit brackets its launch loop with an event pair on that stream (first step
    #     launch ... last launch of the last chunk; read back lazily, gslnls_dense_loop_event_stats); their sum is the
    #     device time of all step launches of the timed region, trailing ones and the gaps between launches included.
    #     Algorithmic bytes by SURVEY.md 8(d): the reference reads x and y once per f evaluation and once per Jacobian
    #     evaluation (16n B each) = 32n per LM iteration with one trial, +16n per extra rejected trial, 32n for the
    #     initial point; "a speculative single fused pass is allowed but the reported denominator stays 32n".
    #     neval.f + neval.J is exactly that count of passes.
    # (2) streamed view: back-to-back launches that each do the full prologue + one full fused pass
    #     (gslnls_dense_time_pass), 16n B actually read per launch -- what the PMC traffic is compared with.
    assert ev_launches.value == launches, (ev_launches.value, launches)
    alg_bytes_fit = 16.0 * n * ref_passes / args.steps
    ms_loop_fit = ev_ms.value / args.steps
    launches_fit = launches / args.steps
    achieved = alg_bytes_fit / (ms_loop_fit * 1e-3) / 1e9
    th = np.array([4.0, 1.2, 0.8])
    ms_launch = float(L.gslnls_dense_time_pass(h, jac, th.ctypes.data_as(_lib.DP), 2000))
    streamed = 16.0 * n / (ms_launch * 1e-3) / 1e9 if ms_launch > 0 else None
    traffic = None
    traffic_source = None
    tpath = next((os.path.join(ROOT, "profiles", f) for f in ("r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json")
                  if os.path.exists(os.path.join(ROOT, "profiles", f))), "")
    if os.path.exists(tpath):
        traffic_source = "committed profile %s (separate rocprofv3 --pmc passes), NOT measured in this run" % os.path.relpath(tpath, ROOT)
        try:
            # PMC passes (scripts/profile_round.sh): 2 x FETCH_SIZE (gfx950 correction) + WRITE_SIZE per launch
            pmc = json.load(open(tpath))
            traffic = next((v.get("bytes_per_launch_corrected") for k, v in pmc.items()
                            if k.startswith("lm_step_kernel<gslnls::ModelExpDecay")), None)
        except Exception:  # noqa
            traffic = None

    line = {
        "metric": "LM iterations/sec at n=1e6,p=3 single-GPU; multi-start fits/sec at 1/2/4/8 GPUs",
        "value": tot_iters / tmax,
        "unit": "LM iterations/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": tmax / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "C2: exponential model n=%d p=3, LM normal equations (cholesky), %s Jacobian, "
                               "one step = one complete fit from (1,1,0)" % (n, "forward-FD" if args.fd else "analytic"),
                   "niter_per_fit": iters / args.steps, "launches_per_fit": launches / args.steps,
                   "device_loop_ms_per_fit": ev_ms.value / args.steps, "host_loop_ms_per_fit": loop_ms / args.steps,
                   "parallelism": "replicas only (x%d)" % world, "par": [float(v) for v in par]},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                     "kernel": "lm_step_kernel<ModelExpDecay>",
                     "bytes_per_launch": alg_bytes_fit / launches_fit, "ms_per_launch": ms_loop_fit / launches_fit,
                     "accounting": "SURVEY 8(d): 16n B per reference pass (f or J evaluation); %.0f passes per fit "
                                   "done in %.0f launches; device time = HIP event pair around every fit's launch loop on the "
                                   "library's stream, summed over the timed region" % (ref_passes / args.steps, launches_fit),
                     "streamed": {"bytes_per_launch": 16.0 * n, "ms_per_full_launch": ms_launch, "GBs": streamed,
                                  "frac": streamed / HBM_PEAK_GBS if streamed else None,
                                  "note": "bytes actually read by one fused launch (x and y once); compare traffic"}},
    }

    def side(name, fn, *a, **kw):
        """A side measurement must not cost the line: a failure becomes {"error": ...} -- on every rank, agreed."""
        out, e = run_leg(job, name, fn, *a, **kw)
        line[name] = out if e is None else {"error": e}
        if e is not None:
            sys.stderr.write("bench.py: side leg failed -- %s\n" % e)

    if rank == 0 and world == 1:
        def other_jacobian():
            # SURVEY.md 8(d) asks for the forward-difference run beside the analytic one
            other = 1 - jac
            for _ in range(3):
                solve(h, other, 0, st_p, None, ci_p, cd_p, args.chunk, res_p)
            torch.cuda.synchronize()
            kfd = max(5, min(50, args.steps))
            t1 = time.perf_counter()
            it_o = 0
            for _ in range(kfd):
                solve(h, other, 0, st_p, None, ci_p, cd_p, args.chunk, res_p)
                it_o += res.niter
            torch.cuda.synchronize()
            el_o = time.perf_counter() - t1
            return {"jacobian": "analytic" if other else "forward-FD", "value": it_o / el_o,
                    "unit": "LM iterations/s", "fits": kfd, "niter_per_fit": it_o / kfd,
                    "launches_per_fit": res.n_launches, "neval_f": res.neval[0], "neval_J": res.neval[1]}

        def one_shot_vs_repeated():
            # a fit on a fresh handle (the launch count of the previous fit is not known yet: 16 launches + top-ups,
            # trailing ones run as no-ops) beside the repeated-fit loop timed above, whose first chunk is sized by the
            # previous fit
            h2 = L.gslnls_dense_create(C.byref(model), y.ctypes.data_as(C.c_void_p), n, None, C.byref(err))
            solve(h2, jac, 0, st_p, None, ci_p, cd_p, args.chunk, res_p)
            first = {"loop_ms": res.loop_ms, "launches": res.n_launches, "niter": res.niter}
            solve(h2, jac, 0, st_p, None, ci_p, cd_p, args.chunk, res_p)
            out = {"first_fit_on_a_fresh_handle": first,
                   "second_fit_on_it": {"loop_ms": res.loop_ms, "launches": res.n_launches},
                   "timed_loop_ms_per_fit": loop_ms / args.steps,
                   "note": "value/ms_per_step above are the repeated-fit loop (the best case)"}
            L.gslnls_dense_destroy(h2)
            return out
        def streamed_large_n():
            line["roofline"]["streamed_large_n"] = streamed_large_n_bench(L, _lib, jac)
            return {"see": "roofline.streamed_large_n"}
        if not args.headline_only:
            side("streamed_large_n", streamed_large_n)
        side("other_jacobian", other_jacobian)
        side("one_shot_vs_repeated", one_shot_vs_repeated)
        side("end_to_end", end_to_end_bench, L, _lib, x, y, n, ci_p, cd_p)
    lib_comm = None
    if world > 1:
        def bind():
            if os.environ.get("GSLNLS_BENCH_ONE_DEVICE"):
                return None, "developer dry run: the ranks share one device, RCCL refuses that"
            return bind_library_comm(L, torch, dist, rank, world, backend)
        got, e = run_leg(job, "communicator", bind)
        lib_comm, why = got if e is None else (None, e)
        if lib_comm:
            lib_comm = why
        else:
            # the records are then gathered by torch.distributed (and batched IRLS by the callback form)
            print("bench.py: in-library RCCL communicator not available (%s); torch.distributed gathers" % why, file=sys.stderr)
            from gslnls_amd import dist as gdist
            gdist.init_multistart_comm(65536 * world, 8)
    side("multistart", multistart_bench, L, _lib, job, max(5, args.steps // 4), 3, lib_comm)
    if not args.headline_only:
        side("batched_irls", batch_irls_bench, _lib, job)
        if world == 1:
            side("large_cgst", large_bench, L, _lib)
            side("large_sparse_readme", sparse_readme_bench)
            side("wide_dense", wide_dense_bench, _lib)
            side("wide_multistart", wide_multistart_bench, L, _lib)
            side("function_models", function_model_bench)
            side("matrix_path", matrix_path_bench, _lib)
    if world > 1:
        L.gslnls_comm_destroy()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # rank 0 at N = 1 only (the contract): about 20 s of one core plus the all-core figure
        line["cpu_baseline"] = cpu_baseline(x, y)
        if not args.headline_only:
            try:
                line["cpu_baseline"]["all_cores"] = cpu_baseline_allcores(n, 20250927 + rank)
            except Exception as e:  # noqa
                line["cpu_baseline"]["all_cores"] = {"error": repr(e)}
    elif rank == 0:
        line["cpu_baseline"] = None
    L.gslnls_dense_destroy(h)
    if dist is not None:
        job.sync(True)
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line))


if __name__ == "__main__":
    main()
