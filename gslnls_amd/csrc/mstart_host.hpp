// mstart_host.hpp -- the HIP BatchEvaluator behind multi-start and the numeric body of the
// multi-start branch of C_nls_internal (src/nls.c:274-532).
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
#include "batch_kernels.hpp"
#include "dense_host.hpp"
#include "mstart_driver.hpp"

namespace gslnls
{

template <class M>
struct HipMsEvaluator : MsEvaluator
{
    static constexpr int P = M::P;
    static constexpr int K = MsRecord<P>::K;
    DenseFit<M> &fit;
    LmParams prm; // tolerances / scaling / FD settings of the fit (maxiter, gtol overridden per batch)
    int jacmode;
    const double *lupars;
    SobolTable *d_sobol = nullptr;
    long long *d_draw = nullptr;
    double *d_start = nullptr, *d_rec = nullptr;
    MsBatch batch_cache;     // reused by DenseFit::mstart_batch
    double *h_rec = nullptr; // pinned staging for the records: a pageable destination costs ~7x the PCIe time
    int cap = 0;
    float last_kernel_ms = 0.f;
    hipEvent_t e0 = nullptr, e1 = nullptr;

    // settings of one call (the object itself -- Sobol table, staging buffers -- is kept by the problem)
    void configure(const int *ci, const double *cd, int jac, int fvv, const double *lu)
    {
        lupars = lu;
        prm = make_params(ci, cd, jac, fvv, lu != nullptr, fit.ctx.sw != nullptr);
        jacmode = jac ? JAC_ANALYTIC : (ci[5] ? JAC_CENTER : JAC_FORWARD);
    }

    HipMsEvaluator(DenseFit<M> &f, const int *ci, const double *cd, int jac, int fvv, const double *lu)
        : fit(f), lupars(lu)
    {
        configure(ci, cd, jac, fvv, lu);
        SobolTable t;
        sobol_build(t, P);
        hipMalloc(&d_sobol, sizeof(SobolTable));
        hipMemcpy(d_sobol, &t, sizeof(SobolTable), hipMemcpyHostToDevice);
        hipEventCreate(&e0);
        hipEventCreate(&e1);
    }
    ~HipMsEvaluator() override
    {
        hipFree(d_sobol);
        hipFree(d_draw);
        hipFree(d_start);
        hipFree(d_rec);
        if (h_rec)
            (void)hipHostFree(h_rec);
        if (e0)
            hipEventDestroy(e0);
        if (e1)
            hipEventDestroy(e1);
    }

    int ensure(int count)
    {
        if (count <= cap)
            return 0;
        hipFree(d_draw);
        hipFree(d_start);
        hipFree(d_rec);
        if (h_rec)
            (void)hipHostFree(h_rec);
        h_rec = nullptr;
        GSLNLS_HIP_OK(hipHostMalloc(&h_rec, sizeof(double) * (size_t)count * K));
        GSLNLS_HIP_OK(hipMalloc(&d_draw, sizeof(long long) * count));
        GSLNLS_HIP_OK(hipMalloc(&d_start, sizeof(double) * (size_t)count * P));
        GSLNLS_HIP_OK(hipMalloc(&d_rec, sizeof(double) * (size_t)count * K));
        cap = count;
        return 0;
    }

    bool defer_sync = false; // run_async: leave the stream running, the collective is enqueued behind the kernel
    int run_async(MsBatch &b, int lo, int hi, double *dev_out) override
    {
        defer_sync = true;
        const int rc = run(b, lo, hi, dev_out, true);
        defer_sync = false;
        return rc;
    }
    void *stream() override { return (void *)fit.stream; }
    int fetch_stream(const double *dev_src, double *dst, size_t nd) override
    {
        if (nd)
            GSLNLS_HIP_OK(hipMemcpyAsync(dst, dev_src, sizeof(double) * nd, hipMemcpyDeviceToHost, fit.stream));
        GSLNLS_HIP_OK(hipStreamSynchronize(fit.stream));
        if (e0 && e1)
            hipEventElapsedTime(&last_kernel_ms, e0, e1);
        return 0;
    }
    int poke(double *dev_dst, double value) override
    {
        GSLNLS_HIP_OK(hipMemcpyAsync(dev_dst, &value, sizeof(double), hipMemcpyHostToDevice, fit.stream));
        GSLNLS_HIP_OK(hipStreamSynchronize(fit.stream));
        return 0;
    }
    int run(MsBatch &b, int lo, int hi, double *out, bool out_on_device) override
    {
        if (b.p != P || b.K != K)
            return GSLNLS_EINVAL;
        int rc = ensure(b.count);
        if (rc)
            return rc;
        hipStream_t st = fit.stream;
        // consecutive fresh draws (the first major iteration, the benchmark) need no per-point upload
        bool consecutive = b.count > 0 && b.draw[0] >= 0;
        if (b.consecutive >= 0)
            consecutive = consecutive && b.consecutive == 1; // the caller built the batch that way (65536 compares are ~20 us)
        else
            for (int i = 1; consecutive && i < b.count; ++i)
                consecutive = b.draw[i] == b.draw[0] + i;
        if (!consecutive)
        {
            GSLNLS_HIP_OK(hipMemcpyAsync(d_draw, b.draw.data(), sizeof(long long) * b.count, hipMemcpyHostToDevice, st));
            GSLNLS_HIP_OK(hipMemcpyAsync(d_start, b.start.data(), sizeof(double) * (size_t)b.count * P,
                                         hipMemcpyHostToDevice, st));
        }
        MsKernelArgs<P> a;
        for (int c = 0; c < 4; ++c)
            a.x[c] = c < M::NX ? fit.ctx.x[c] : nullptr;
        a.y = fit.ctx.y;
        a.sw = fit.ctx.sw;
        a.draw = consecutive ? nullptr : d_draw;
        a.first_draw = consecutive ? b.draw[0] : 0;
        a.start = d_start;
        // records of point idx land at records + idx*K: shift the base so that point `lo` lands at out[0]
        double *dev_out = (out_on_device && out) ? out : d_rec; // (device output without a buffer: the evaluator's own)
        a.records = dev_out - (size_t)lo * K;
        a.sobol = d_sobol;
        a.lo = lo;
        a.hi = hi;
        for (int k = 0; k < P; ++k)
        {
            a.l0[k] = b.range[2 * k];
            a.l1[k] = b.range[2 * k + 1];
            a.kd[k] = b.kd[k];
            a.lu[2 * k] = lupars ? lupars[2 * k] : -INFINITY;
            a.lu[2 * k + 1] = lupars ? lupars[2 * k + 1] : INFINITY;
        }
        a.has_lu = lupars != nullptr;
        a.mp.prm = prm;
        a.mp.prm.maxiter = b.maxiter;
        a.mp.prm.gtol = 1e-3; // src/nls_mstart.c:91, :254
        a.mp.dtol = b.dtol;
        a.mp.n = fit.n;
        a.mp.always_fit = b.always_fit;
        // a batch that leaves most SIMDs empty anyway runs two lanes per fit when the data set sits in registers:
        // its duration is the latency of the slowest fit, and the pass over the rows is split between the lanes
        // A batch that leaves most SIMDs empty anyway runs two or four lanes per fit when the data set sits in
        // registers: its duration is the latency of the slowest fit, and the pass over the rows is split between the
        // lanes of a group (8192 BoxBOD points, 5 iterations: 43.9 us with one lane, 37.4 with two, 36.3 with four).
        // (hand-written models only: for the expression models every extra instantiation is seconds of hipcc -- in
        // front of the first fit of a natively lowered formula, and minutes over the nine interpreter units of the
        // library build)
        constexpr bool MULTI_LANE = M::ID < 100;
        int lpf = 1;
        if (MULTI_LANE && fit.n <= MS_REG_ROWS)
        {
            if (fit.n > 4 && (long long)(hi - lo) * 4 <= 65536)
                lpf = 4;
            else if (fit.n > 1 && (long long)(hi - lo) * 2 <= 65536)
                lpf = 2;
        }
        // Lane refill (ms_fit_refill_kernel: a wavefront owns a slice of the batch, a lane that finishes takes the slice's
        // next point) is built, bit-identical, and MEASURED SLOWER than letting the hardware refill at wavefront
        // granularity: 1,048,576 BoxBOD points 0.455 ms against 0.378 ms, 262,144 points 0.142 against 0.113 ms (best
        // of 1024 / 2048 / 4096 wavefronts x refill thresholds 8 / 16 / 32; gpurun_out r03d).  The refill
        // path (record, Sobol point, state reset) runs divergent, the kernel needs 256 VGPRs + scratch against 182, and
        // with two to four wavefronts per SIMD the scheduler already fills the slots a finished wavefront frees.
        // Opt-in: GSLNLS_MS_REFILL=1 (GSLNLS_MS_WAVES wavefronts, default two per SIMD; GSLNLS_MS_REFILL_AT lanes).
        int refill_waves = 0;
        if (MULTI_LANE && lpf == 1)
        {
            const char *re = getenv("GSLNLS_MS_REFILL"), *we = getenv("GSLNLS_MS_WAVES");
            const int refill_env = re ? atoi(re) : 0, waves_env = we ? atoi(we) : 0;
            const int waves = waves_env > 0 ? waves_env : 2048;
            if (refill_env && (long long)(hi - lo) >= 2LL * waves * MS_T)
                refill_waves = waves;
        }
        const int nblk = (int)(((long long)(hi - lo) * lpf + MS_T - 1) / MS_T);
        const size_t lds = (fit.n <= MS_LDS_ROWS) ? sizeof(double) * (size_t)fit.n * (M::NX + 2) : 0;
        hipEventRecord(e0, st);
#define GSLNLS_MS_LAUNCH(JACMODE)                                                                              \
    if constexpr (MULTI_LANE)                                                                                  \
    {                                                                                                          \
        if (lpf == 4)                                                                                          \
            hipLaunchKernelGGL((ms_fit_kernel<M, JACMODE, 4>), dim3(nblk), dim3(MS_T), lds, st, a);              \
        else if (lpf == 2)                                                                                     \
            hipLaunchKernelGGL((ms_fit_kernel<M, JACMODE, 2>), dim3(nblk), dim3(MS_T), lds, st, a);              \
    }                                                                                                          \
    if (lpf == 1)                                                                                              \
        hipLaunchKernelGGL((ms_fit_kernel<M, JACMODE, 1>), dim3(nblk), dim3(MS_T), lds, st, a);
        if constexpr (MULTI_LANE)
        {
            if (refill_waves > 0)
            {
                const int slice = (int)(((long long)(hi - lo) + refill_waves - 1) / refill_waves);
                const char *te = getenv("GSLNLS_MS_REFILL_AT");
                const int thresh = te ? atoi(te) : 16; // finished lanes that make a wavefront stop for the refill
                const int nw = (int)(((long long)(hi - lo) + slice - 1) / slice);
                switch (jacmode)
                {
                case JAC_ANALYTIC:
                    hipLaunchKernelGGL((ms_fit_refill_kernel<M, JAC_ANALYTIC>), dim3(nw), dim3(MS_T), lds, st, a, slice, thresh);
                    break;
                case JAC_FORWARD:
                    hipLaunchKernelGGL((ms_fit_refill_kernel<M, JAC_FORWARD>), dim3(nw), dim3(MS_T), lds, st, a, slice, thresh);
                    break;
                default:
                    hipLaunchKernelGGL((ms_fit_refill_kernel<M, JAC_CENTER>), dim3(nw), dim3(MS_T), lds, st, a, slice, thresh);
                    break;
                }
            }
        }
        if (refill_waves == 0)
        switch (jacmode)
        {
        case JAC_ANALYTIC:
            GSLNLS_MS_LAUNCH(JAC_ANALYTIC)
            break;
        case JAC_FORWARD:
            GSLNLS_MS_LAUNCH(JAC_FORWARD)
            break;
        default:
            GSLNLS_MS_LAUNCH(JAC_CENTER)
            break;
        }
#undef GSLNLS_MS_LAUNCH
        hipEventRecord(e1, st);
        if (hipGetLastError() != hipSuccess)
            return GSLNLS_E_NODEVICE;
        if (defer_sync && out_on_device)
            return 0;
        if (!out_on_device)
            GSLNLS_HIP_OK(hipMemcpyAsync(h_rec, d_rec, sizeof(double) * (size_t)(hi - lo) * K, hipMemcpyDeviceToHost, st));
        GSLNLS_HIP_OK(hipStreamSynchronize(st));
        if (!out_on_device && out)
            memcpy(out, h_rec, sizeof(double) * (size_t)(hi - lo) * K);
        hipEventElapsedTime(&last_kernel_ms, e0, e1);
        return 0;
    }
    // the records of a whole batch, consumed by the driver where the copy engine wrote them
    int run_view(MsBatch &b, const double **view) override
    {
        const int rc = run(b, 0, b.count, nullptr, false);
        *view = h_rec;
        return rc;
    }
    int fetch_stream_view(const double *dev_src, size_t nd, MsBatch &b, const double **view) override
    {
        int rc = ensure(b.count);
        if (rc)
            return rc;
        if (nd > (size_t)cap * K)
            return GSLNLS_EINVAL;
        rc = fetch_stream(dev_src, h_rec, nd);
        *view = h_rec;
        return rc;
    }

    int fetch(const double *src, bool src_on_device, double *dst, size_t nd) override
    {
        if (src_on_device)
            GSLNLS_HIP_OK(hipMemcpy(dst, src, sizeof(double) * nd, hipMemcpyDeviceToHost));
        else
            memcpy(dst, src, sizeof(double) * nd);
        return 0;
    }
};

template <class M>
int DenseFit<M>::mstart(int jac, int fvv, const double *start2p, const double *lupars, const int *ci, const double *cd,
                        const int *has_start, const MsComm &comm, int loss_rho, const double *loss_cc,
                        gslnls_result *out)
{
    if (ci[2] > 1)
        return GSLNLS_E_UNSUPPORTED;
    if (fvv && !M::HAS_FVV)
        return GSLNLS_E_UNSUPPORTED;
    MsState m;
    ms_init(m, P, ci, cd, start2p, has_start, lupars);
    if (!ms_eval)
        ms_eval = new HipMsEvaluator<M>(*this, ci, cd, jac, fvv, lupars);
    HipMsEvaluator<M> &ev = *static_cast<HipMsEvaluator<M> *>(ms_eval);
    ev.configure(ci, cd, jac, fvv, lupars);
    ev.batch_cache.consecutive = -1;
    int rc = ms_major_loop(m, ev, comm, start2p);
    if (rc)
        return rc < 0 && rc > -100 ? GSLNLS_FAILURE : rc;
    // robust second pass (src/nls.c:401-509): Cook's-distance outliers get zero weight, then the whole
    // multi-start is repeated (fresh counters and quasi-random sequence, ranges and exponents carried over)
    if (loss_rho != 0)
    {
        if (m.mssropt[1] < m.mssropt[0])
            m.mpopt = m.mpopt1;
        prepare(jac, fvv, lupars, ci, cd, false);
        const int jacmode = jac ? JAC_ANALYTIC : (ci[5] ? JAC_CENTER : JAC_FORWARD);
        double *d_sw_robust = nullptr;
        GSLNLS_HIP_OK(hipMalloc(&d_sw_robust, sizeof(double) * (size_t)n));
        const int second = robust_weights(jacmode, m.mpopt.data(), d_sw_robust);
        if (second < 0)
        {
            hipFree(d_sw_robust);
            return second;
        }
        if (second == 1)
        {
            const double *keep_sw = ctx.sw;
            ctx.sw = d_sw_robust;
            ev.prm.has_weights = 1;
            m.next_draw = 0; // gsl_qrng_init
            m.second_pass = true;
            m.mstop = ST_CONTINUE;
            m.mstarts = m.nsp = m.nwsp = 0;
            m.dtol = 1.0e-6;
            m.rejectscl = 1.25;
            m.mssropt[0] = m.mssropt[1] = INFINITY;
            m.ssrconv[0] = m.ssrconv[1] = 1.0;
            std::fill(m.ntix.begin(), m.ntix.end(), 0);
            std::fill(m.luchange.begin(), m.luchange.end(), 0);
            rc = ms_major_loop(m, ev, comm, start2p);
            ctx.sw = keep_sw; // "reset original weights" (src/nls.c:490-507)
        }
        hipFree(d_sw_robust);
        if (rc)
            return rc < 0 && rc > -100 ? GSLNLS_FAILURE : rc;
    }
    ms_trace_finished(m);
    // src/nls.c:518-531
    if (m.mssropt[1] < m.mssropt[0])
    {
        m.mssropt[0] = m.mssropt[1];
        m.ssrconv[0] = m.ssrconv[1];
        m.mpopt = m.mpopt1;
    }
    const double ftol = cd[6];
    if (m.mssropt[0] < ftol || m.ssrconv[0] < ftol)
    {
        if (lupars)
            m.mpopt[0] = fmin(m.mpopt[0] + 1.0e-4, isfinite(lupars[1]) ? lupars[1] : INFINITY);
        else
            m.mpopt[0] = m.mpopt[0] + 1.0e-4;
    }
    if (loss_rho != 0)
        rc = irls(jac, fvv, m.mpopt.data(), lupars, ci, cd, loss_rho, loss_cc, out);
    else
        rc = solve(jac, fvv, m.mpopt.data(), lupars, ci, cd, 0, out);
    out->mstart_nsp = m.nsp;
    out->mstart_nwsp = m.nwsp;
    out->mstart_iters = m.mstarts;
    out->mstart_stop = m.mstop;
    out->mstart_ssropt = m.mssropt[0];
    return rc;
}

template <class M>
int DenseFit<M>::mstart_batch(int jac, const double *ranges, const double *kd, long long first_draw, int count, int lo,
                              int hi, int maxiter, double dtol, const int *ci, const double *cd, const double *lupars,
                              double *records, int records_on_device, float *kernel_ms)
{
    // the evaluator (Sobol table, device buffers) is kept between calls with the same settings
    if (!ms_eval)
        ms_eval = new HipMsEvaluator<M>(*this, ci, cd, jac, 0, lupars);
    HipMsEvaluator<M> &ev = *static_cast<HipMsEvaluator<M> *>(ms_eval);
    ev.prm = make_params(ci, cd, jac, 0, lupars != nullptr, ctx.sw != nullptr);
    ev.jacmode = jac ? JAC_ANALYTIC : (ci[5] ? JAC_CENTER : JAC_FORWARD);
    ev.lupars = lupars;
    // the batch description is kept too: a fresh 1.5 MB of vectors per call (65536 points) costs more host time
    // in page faults than the kernel takes
    MsBatch &b = ev.batch_cache;
    if (b.count != count || b.draw.empty() || b.draw[0] != first_draw)
    {
        b.count = count;
        b.draw.resize(count);
        for (int i = 0; i < count; ++i)
            b.draw[i] = first_draw + i;
        b.start.assign((size_t)count * P, 0.0);
    }
    b.consecutive = 1;
    b.p = P;
    b.K = MsRecord<P>::K;
    b.range.assign(ranges, ranges + 2 * P);
    b.kd.resize(P);
    for (int k = 0; k < P; ++k)
        b.kd[k] = kd ? kd[k] : 0.75;
    b.maxiter = maxiter;
    b.dtol = dtol;
    b.always_fit = 0;
    int rc;
    if (lo < 0)
    {
        // whole batch, sharded over the ranks of the bound communicator and completed by its all-gather -- exactly
        // what one concentration stage of the multi-start driver does (ms_run_batch); records: host, count x K
        if (!batch_comm)
            return GSLNLS_EINVAL;
        b.host_records = (records != nullptr);
        rc = ms_run_batch(ev, *batch_comm, b);
        if (rc == 0 && records)
            memcpy(records, b.rec, sizeof(double) * (size_t)count * b.K);
        b.host_records = true;
    }
    else
        rc = ev.run(b, lo, hi, records, records_on_device != 0);
    if (kernel_ms)
        *kernel_ms = ev.last_kernel_ms;
    return rc;
}

} // namespace gslnls
