// wide_host.hpp -- host orchestration of gsl_nls() for 10 <= p <= 64 parameters (and up to WIDE_NX data columns).
//
// The reference takes any p (src/nls.c:266 allocates the n x p workspace; R/nls.R:588-599 derives the Jacobian of any
// formula).  lm_core.hpp / dense_kernels.hpp keep the whole p x p algebra in registers and stop at p = 9; from there on
// one trial step of trust_iterate_lu_LD (src/trust.c:445-546) is three stream-ordered launches:
//   wide_pass_kernel     rows -> per-workgroup sums; J^T J on the matrix cores (wide_kernels.hpp; compiled in process
//                        for the formula, rtc_host.hpp -- there is no interpreted form of the wide path);
//   wide_reduce_kernel   the G partial sets -> one (coalesced: thread = value, loop over the sets in index order);
//   wide_advance_kernel  one workgroup: rho, accept / reject, mu, D, modified Cholesky, next trial point (wide_core.hpp).
// The host enqueues chunks of steps and polls the device's completion word, exactly like dense_host.hpp.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <string>
#include <vector>
#include "dense_host.hpp"
#include "irls_kernels.hpp"
#include "large_host.hpp"
#include "robust_host.hpp"
#include "rtc_host.hpp"
#include "wide_core.hpp"
#include "wide_kernels.hpp"

namespace gslnls
{

// G partial sets -> one.  A workgroup owns 64 consecutive values; its 1024 threads are 16 groups of 64 lanes, group q adds
// the sets q, q + 16, q + 32, ... (all loads of a thread are independent and in flight together: the sets were written
// by other XCDs, a load is a ~2 us fabric round trip, and 256 of them one after the other took 57 us), then the 16
// group sums are added in group order.  Fixed order => bit-identical results run to run.
constexpr int WIDE_RED_T = 1024;
__global__ __launch_bounds__(WIDE_RED_T) void wide_reduce_kernel(const double *partials, int G, int NV, int NVP, double *totals,
                                                                 const WState *state)
{
    __shared__ double part[16][64];
    if (state->phase == PH_DONE)
        return;
    const int l = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int v = blockIdx.x * 64 + l;
    double s = 0.0;
    if (v < NV)
    {
        constexpr int PER = WIDE_MAX_G / 16; // sets per group
        double t[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k)
        {
            const int g = q + 16 * k;
            t[k] = g < G ? partials[(size_t)g * NVP + v] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < PER; ++k)
            s += t[k];
    }
    part[q][l] = s;
    __syncthreads();
    if (q == 0 && v < NV)
    {
        double r = part[0][l];
#pragma unroll
        for (int k = 1; k < 16; ++k)
            r += part[k][l];
        totals[v] = r;
    }
}

__global__ __launch_bounds__(WT_ADV) void wide_advance_kernel(WAdvanceArgs a)
{
    __shared__ WideLds L;
    wide_advance(a, L);
}

// bit patterns of |r_i|: the keys of the radix select that finds the median (irls_kernels.hpp)
__global__ __launch_bounds__(256) void wide_keys_kernel(const double *r, long long n, unsigned long long *keys)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        keys[i] = (unsigned long long)__double_as_longlong(fabs(r[i]));
}

// hat values h_i = J_i (J^T J)^-1 J_i^T and Cook's distances D_i = e_i^2 / (p s^2) h_i / (1 - h_i)^2 (hat_values, cooks_d:
// src/nls_utils.c:88-150) from the weighted residual and Jacobian the finalize kernel left in HBM (grad: n x p
// column-major, so the reads of a wavefront are contiguous for every k); (J^T J)^-1 sits in LDS
__global__ __launch_bounds__(256) void wide_cooks_kernel(const double *resid, const double *grad, long long n, int p,
                                                         const double *Cinv, double s2, double *d, unsigned long long *keys,
                                                         double *hat)
{
    __shared__ double C_s[WP * WP];
    for (int e = threadIdx.x; e < p * p; e += 256)
        C_s[e] = Cinv[e];
    __syncthreads();
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    {
        double h = 0.0;
        for (int j = 0; j < p; ++j)
        {
            double t = 0.0;
            for (int k = 0; k < p; ++k)
                t += grad[i + (size_t)n * k] * C_s[k * p + j];
            h += t * grad[i + (size_t)n * j];
        }
        const double e = resid[i];
        const double di = (e * e) / (p * s2) * (h / ((1 - h) * (1 - h)));
        if (d)
            d[i] = di;
        if (keys)
            keys[i] = (unsigned long long)__double_as_longlong(fabs(di));
        if (hat)
            hat[i] = h;
    }
}

// test hook: the damped solve alone (tests compare it with the oracle's modified Cholesky)
__global__ __launch_bounds__(64) void wide_solve_debug_kernel(int p, const double *Ap, const double *diag, double mu,
                                                              const double *rhs, double *sol, int pivoted)
{
    __shared__ WideLds L;
    const int lane = threadIdx.x;
    {
        // (all loads in flight before the first is waited for, as in wide_advance)
        constexpr int IT = (WNA + 63) / 64;
        double buf[IT];
#pragma unroll
        for (int i = 0; i < IT; ++i)
            buf[i] = lane + 64 * i < p * (p + 1) / 2 ? Ap[lane + 64 * i] : 0.0;
#pragma unroll
        for (int i = 0; i < IT; ++i)
            if (lane + 64 * i < p * (p + 1) / 2)
                L.A[lane + 64 * i] = buf[i];
    }
    if (lane < p)
    {
        L.diag[lane] = diag[lane];
        L.rhs[lane] = rhs[lane];
    }
    wide_lds_sync();
    wide_solve(L, p, mu, L.rhs, L.sol, lane, pivoted);
    if (lane < p)
        sol[lane] = L.sol[lane];
}

inline std::string rtc_wide_source(const WideProgram &pr, int nx_model)
{
    std::string s = "// generated by gslnls wide_host.hpp\n";
    if (const char *e = getenv("GSLNLS_WIDE_WAVES")) // developer switch: workgroups per CU the pass is compiled for
        s += std::string("#define GSLNLS_WIDE_WAVES(PW) ") + (atoi(e) >= 2 ? "2" : "1") + "\n";
    s += "#include \"wide_kernels.hpp\"\n";
    s += rtc_emit_model(pr, nx_model);
    return s;
}
inline std::string rtc_wide_pass_expr(int jacmode, int PW)
{
    return "&gslnls::wide_pass_kernel<gslnls::ModelJit, " + std::to_string(jacmode) + ", " + std::to_string(PW) + ">";
}
inline std::string rtc_wide_step_expr(int jacmode, int PW)
{
    return "&gslnls::wide_step_kernel<gslnls::ModelJit, " + std::to_string(jacmode) + ", " + std::to_string(PW) + ">";
}
inline std::string rtc_wide_fit_expr(int jacmode, int PW)
{
    return "&gslnls::wide_fit_kernel<gslnls::ModelJit, " + std::to_string(jacmode) + ", " + std::to_string(PW) + ">";
}
inline std::string rtc_wide_finalize_expr(int jacmode)
{
    return "&gslnls::wide_finalize_kernel<gslnls::ModelJit, " + std::to_string(jacmode) + ">";
}

// det(J^T J) = (prod L_ii)^2 by plain Cholesky of the packed lower triangle, 0 when not positive definite
// (det_cholesky_jtj, src/nls_utils.c:55-73) -- the run-time-p form of det_cholesky<P> in lm_core.hpp
inline double wide_det_cholesky(int p, const double *Ap)
{
    std::vector<double> Lm((size_t)p * p, 0.0);
    for (int i = 0; i < p; ++i)
        for (int j = 0; j <= i; ++j)
            Lm[(size_t)i * p + j] = Ap[i * (i + 1) / 2 + j];
    double det = 1.0;
    bool ok = true;
    for (int j = 0; j < p; ++j)
    {
        double ajj = Lm[(size_t)j * p + j];
        for (int k = 0; k < j; ++k)
            ajj -= Lm[(size_t)j * p + k] * Lm[(size_t)j * p + k];
        if (!(ajj > 0.0))
            ok = false;
        ajj = sqrt(ajj);
        Lm[(size_t)j * p + j] = ajj;
        det *= ajj;
        for (int i = j + 1; i < p; ++i)
        {
            double t = Lm[(size_t)i * p + j];
            for (int k = 0; k < j; ++k)
                t -= Lm[(size_t)i * p + k] * Lm[(size_t)j * p + k];
            Lm[(size_t)i * p + j] = t / ajj;
        }
    }
    return ok ? det * det : 0.0;
}

struct WideFit;
// the per-point work of gsl_multistart_driver (src/nls_mstart.c:72-95, :245-255) for p > 9: det filter at the sampled
// point, a short fit through the wide path, det where it ended -- one point after the other (a wide fit uses the whole
// chip; the lane-per-fit batch kernel of batch_kernels.hpp holds its state in registers and stops at p = 9)
struct WideMsEvaluator : MsEvaluator
{
    WideFit &fit;
    SobolTable tab;
    LmParams prm;
    int jm = 0;
    const double *lupars = nullptr;
    std::vector<double> staged; // records of a shard on their way to a device buffer (the in-library collective)
    explicit WideMsEvaluator(WideFit &f) : fit(f) {}
    int run(MsBatch &b, int lo, int hi, double *out, bool out_on_device) override;
    int run_host(MsBatch &b, int lo, int hi, double *out);
    int fetch(const double *src, bool src_on_device, double *dst, size_t nd) override;
    // the stream-ordered forms the RCCL all-gather of capi.hip uses (one process per GPU: a rank fits its block of the
    // points one after the other, the records reach the shard buffer in one copy)
    void *stream() override;
    int run_async(MsBatch &b, int lo, int hi, double *dev_out) override;
    int fetch_stream(const double *dev_src, double *dst, size_t nd) override;
    int poke(double *dev_dst, double value) override;
};

struct WideFit : DenseBase
{
    WideProgram *prog = nullptr; // (45 KB: on the heap)
    int nx = 1, PW = 16, G = 1, NV = 0, NVP = 0; // NVP: doubles from one partial set to the next (NV rounded up to even)
    bool owns_data = false, sw_owned = false;
    double *d_x = nullptr, *d_y = nullptr, *d_sw = nullptr;
    const double *cur_sw = nullptr;
    WState *d_state = nullptr, *h_state = nullptr; // h_state: pinned + mapped, [0] staging of the start, [1] final state
    volatile unsigned int *h_done = nullptr;
    double *d_partials = nullptr, *d_totals = nullptr;
    // one launch per trial step (wide_kernels.hpp): level-1 sums and hand-off words of the in-launch reduction
    double *d_gsums = nullptr;
    WFuseBuf *d_fb = nullptr;
    unsigned long long *d_stamps = nullptr; // GSLNLS_WIDE_STAMPS=1 (developer): phase stamps of the fused kernel
    int fuse = WIDE_FUSE_STEP; // GSLNLS_WIDE_FUSE = 0: pass, reduce and advance as three launches (the round-3 chain)
    int spec = 1;              // GSLNLS_WIDE_SPEC = 0: no speculative solve of the step that follows a rejection
    int spec_fault = 0;        // GSLNLS_WIDE_SPEC_FAULT = 1 / 2 / 3: test switch of the hand-off (wide_kernels.hpp, WPassArgs)
    int pivoted = 0;           // GSLNLS_WIDE_PIVOTED = 1: every damped solve by the reference's pivoted modified Cholesky
    double *d_ssrtrace = nullptr, *d_partrace = nullptr, *d_resid = nullptr, *d_grad = nullptr;
    int trace_cap = 0;
    unsigned int seq = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::string rtc_src;
    std::shared_ptr<RtcEntry> rtc[3], rtc_step[3];
    hipFunction_t fn_pass[3] = {nullptr, nullptr, nullptr}, fn_final[3] = {nullptr, nullptr, nullptr};
    // the one-launch-per-step kernel of the formula: several seconds of compiler, so it is built on the background thread
    // (GSLNLS_LOWER_JIT: now) while fits run pass | reduce | advance as three launches, and bound when it is ready --
    // same sums in the same order, same LM step: the switch changes no bit of any result
    hipFunction_t fn_step[3] = {nullptr, nullptr, nullptr};
    bool step_failed[3] = {false, false, false};
    int n_cu = 0;
    int lowering = GSLNLS_LOWER_AUTO;
    // one workgroup = one complete fit (wide_fit_kernel): the batch form of the multi-start evaluator; built when a
    // multi-start first asks for it (GSLNLS_WIDE_MS_BATCH=0: the points are fitted one after the other instead)
    std::shared_ptr<RtcEntry> rtc_fit[3];
    hipFunction_t fn_fit[3] = {nullptr, nullptr, nullptr};
    bool fit_failed[3] = {false, false, false};
    WState *d_ms_states = nullptr;
    double *d_ms_starts = nullptr, *d_ms_records = nullptr, *d_ms_lupars = nullptr;
    int ms_cap = 0;
    double ev_ms_total = 0.0;
    long long ev_launches_total = 0;
    float last_ms = 0.f;
    long long last_launches = 0;
    int pred_kind = -1, pred_steps = 0; // trial steps the previous fit of this kind needed (adaptive first chunk)
    int wf_only = 0;                    // gsl_nls_large on this handle: weights scale f only (WideLargeOps)
    double *d_cinv = nullptr;   // (J^T J)^-1 for the hat values
    void *irls_arena = nullptr; // work arrays of the robust re-weighting, kept between calls
    size_t irls_arena_bytes = 0;

    ~WideFit() override
    {
        delete prog;
        hipFree(irls_arena);
        hipFree(d_cinv);
        if (owns_data)
        {
            hipFree(d_x);
            hipFree(d_y);
        }
        if (sw_owned)
            hipFree(d_sw);
        hipFree(d_state);
        if (h_state)
            (void)hipHostFree(h_state);
        hipFree(d_partials);
        hipFree(d_totals);
        hipFree(d_gsums);
        hipFree(d_fb);
        hipFree(d_stamps);
        hipFree(d_ms_states);
        hipFree(d_ms_starts);
        hipFree(d_ms_records);
        hipFree(d_ms_lupars);
        hipFree(d_ssrtrace);
        hipFree(d_partrace);
        hipFree(d_resid);
        hipFree(d_grad);
        if (ev0)
            hipEventDestroy(ev0);
        if (ev1)
            hipEventDestroy(ev1);
        if (stream)
            hipStreamDestroy(stream);
    }

    int init(const WideProgram &pr, const gslnls_model *fn, const double *y, int n_, const double *swts)
    {
        prog = new WideProgram(pr);
        p = pr.p;
        n = n_;
        nx = fn->nx > 0 ? fn->nx : 1;
        PW = 16 * ((p + 15) / 16);
        NV = 2 + p * (p + 1) / 2 + p;
        NVP = (NV + 1) & ~1;
        const long long tiles = ((long long)n + 63) / 64;
        long long g = (tiles + (WIDE_T / 64) - 1) / (WIDE_T / 64);
        // one workgroup per CU for the wide tiles (PW >= 48: LDS), two below
        // (one less than the device holds: the grid of a step has one more workgroup, the solver / the LM state machine)
        {
            int dev = 0;
            (void)hipGetDevice(&dev);
            if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu < 1)
                n_cu = 256;
        }
        int gmax = (PW <= 32 ? 2 : 1) * n_cu - 1;
        if (gmax > WIDE_MAX_G)
            gmax = WIDE_MAX_G;
        if (gmax < 1)
            gmax = 1;
        G = (int)(g < 1 ? 1 : (g > gmax ? gmax : g));
        GSLNLS_HIP_OK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        GSLNLS_HIP_OK(hipEventCreate(&ev0));
        GSLNLS_HIP_OK(hipEventCreate(&ev1));
        const size_t nb = sizeof(double) * (size_t)n;
        if (fn->x_on_device)
        {
            if (fn->nx < 1 || !fn->x || !y)
                return GSLNLS_EINVAL; // device-resident data: every column the formula reads must exist
            d_x = const_cast<double *>(fn->x);
            d_y = const_cast<double *>(y);
            d_sw = const_cast<double *>(swts);
        }
        else
        {
            owns_data = true;
            GSLNLS_HIP_OK(hipMalloc(&d_x, nb * nx));
            GSLNLS_HIP_OK(hipMalloc(&d_y, nb));
            if (swts)
            {
                GSLNLS_HIP_OK(hipMalloc(&d_sw, nb));
                sw_owned = true;
            }
            const double t_h2d = now_s();
            if (fn->nx > 0)
                GSLNLS_HIP_OK(hipMemcpyAsync(d_x, fn->x, nb * nx, hipMemcpyHostToDevice, stream));
            else
                GSLNLS_HIP_OK(hipMemsetAsync(d_x, 0, nb, stream));
            GSLNLS_HIP_OK(hipMemcpyAsync(d_y, y, nb, hipMemcpyHostToDevice, stream));
            if (swts)
                GSLNLS_HIP_OK(hipMemcpyAsync(d_sw, swts, nb, hipMemcpyHostToDevice, stream));
            GSLNLS_HIP_OK(hipStreamSynchronize(stream));
            g_call_prof.h2d_ms = 1e3 * (now_s() - t_h2d);
        }
        cur_sw = swts ? d_sw : nullptr;
        GSLNLS_HIP_OK(hipMalloc(&d_state, sizeof(WState)));
        GSLNLS_HIP_OK(hipHostMalloc(&h_state, sizeof(WState) * 2 + 64, hipHostMallocMapped));
        memset(h_state, 0, sizeof(WState) * 2 + 64);
        h_done = reinterpret_cast<volatile unsigned int *>(reinterpret_cast<char *>(h_state) + sizeof(WState) * 2);
        GSLNLS_HIP_OK(hipMalloc(&d_partials, sizeof(double) * (size_t)G * NVP));
        GSLNLS_HIP_OK(hipMalloc(&d_totals, sizeof(double) * (size_t)NVP));
        GSLNLS_HIP_OK(hipMalloc(&d_gsums, sizeof(double) * (size_t)WIDE_NGRP * NVP));
        GSLNLS_HIP_OK(hipMalloc(&d_fb, sizeof(WFuseBuf)));
        GSLNLS_HIP_OK(hipMemset(d_fb, 0, sizeof(WFuseBuf)));
        if (const char *e = getenv("GSLNLS_WIDE_FUSE"))
            fuse = atoi(e) <= 0 ? WIDE_FUSE_NONE : WIDE_FUSE_STEP;
        if (const char *e = getenv("GSLNLS_WIDE_SPEC"))
            spec = atoi(e) != 0;
        if (const char *e = getenv("GSLNLS_WIDE_SPEC_FAULT"))
            spec_fault = atoi(e);
        if (const char *e = getenv("GSLNLS_WIDE_PIVOTED"))
            pivoted = atoi(e) != 0;
        if (getenv("GSLNLS_WIDE_STAMPS"))
        {
            GSLNLS_HIP_OK(hipMalloc(&d_stamps, sizeof(unsigned long long) * (10 * 1024 + 16)));
            GSLNLS_HIP_OK(hipMemset(d_stamps, 0, sizeof(unsigned long long) * (10 * 1024 + 16)));
        }
        lowering = fn->lowering;
        rtc_src = rtc_wide_source(*prog, nx);
        return GSLNLS_SUCCESS;
    }

    // the wide path has no interpreter: the kernels of this formula are built (or found in the cache) before the fit
    int bind(int jm)
    {
        if (fn_pass[jm])
        {
            bind_step(jm, false);
            return 0;
        }
        const std::string ep = rtc_wide_pass_expr(jm, PW), ef = rtc_wide_finalize_expr(jm);
        rtc[jm] = rtc_request(rtc_src, {ep, ef}, true);
        std::string msg;
        if (rtc[jm]->state.load() == RTC_READY)
        {
            fn_pass[jm] = rtc_function(*rtc[jm], ep, msg);
            fn_final[jm] = fn_pass[jm] ? rtc_function(*rtc[jm], ef, msg) : nullptr;
        }
        else
            msg = rtc[jm]->log;
        if (!fn_pass[jm] || !fn_final[jm])
        {
            fn_pass[jm] = fn_final[jm] = nullptr;
            fprintf(stderr, "gslnls: cannot build the kernels of a p = %d model (the wide path needs the in-process compiler): %s\n",
                    p, msg.c_str());
            return GSLNLS_E_UNSUPPORTED;
        }
        bind_step(jm, lowering == GSLNLS_LOWER_JIT);
        return 0;
    }

    // the batch kernel of the multi-start evaluator: built (or found in the cache) now; false = not available
    bool bind_fit(int jm)
    {
        // (read per call: the benchmark times both forms in one process)
        const char *e = getenv("GSLNLS_WIDE_MS_BATCH");
        if ((e && atoi(e) == 0) || fit_failed[jm])
            return false;
        if (fn_fit[jm])
            return true;
        const std::string ef = rtc_wide_fit_expr(jm, PW);
        rtc_fit[jm] = rtc_request(rtc_src, {ef}, true);
        std::string msg;
        if (rtc_fit[jm]->state.load() == RTC_READY)
            fn_fit[jm] = rtc_function(*rtc_fit[jm], ef, msg);
        if (!fn_fit[jm])
            fit_failed[jm] = true;
        return fn_fit[jm] != nullptr;
    }

    // `count` complete short fits, one workgroup each, at most MS_CHUNK per launch (a workgroup keeps a 21 KB state in HBM):
    // starts [count][p] (host) -> records [count][3 p + 8] (host)
    static constexpr int MS_CHUNK = 16384;
    int fit_batch(int jm, const LmParams &q, const double *starts, int count, const double *lupars, bool always_fit, double dtol,
                  double *records)
    {
        const int K = 3 * p + 8;
        const int cap = count < MS_CHUNK ? count : MS_CHUNK;
        if (ms_cap < cap)
        {
            hipFree(d_ms_states);
            hipFree(d_ms_starts);
            hipFree(d_ms_records);
            d_ms_states = nullptr;
            d_ms_starts = d_ms_records = nullptr;
            ms_cap = 0;
            GSLNLS_HIP_OK(hipMalloc(&d_ms_states, sizeof(WState) * (size_t)cap));
            GSLNLS_HIP_OK(hipMalloc(&d_ms_starts, sizeof(double) * (size_t)cap * p));
            GSLNLS_HIP_OK(hipMalloc(&d_ms_records, sizeof(double) * (size_t)cap * K));
            ms_cap = cap;
        }
        if (lupars && !d_ms_lupars)
            GSLNLS_HIP_OK(hipMalloc(&d_ms_lupars, sizeof(double) * 2 * WP));
        if (lupars)
            GSLNLS_HIP_OK(hipMemcpyAsync(d_ms_lupars, lupars, sizeof(double) * 2 * p, hipMemcpyHostToDevice, stream));
        for (int lo = 0; lo < count; lo += MS_CHUNK)
        {
            const int m = count - lo < MS_CHUNK ? count - lo : MS_CHUNK;
            GSLNLS_HIP_OK(hipMemcpyAsync(d_ms_starts, starts + (size_t)lo * p, sizeof(double) * (size_t)m * p, hipMemcpyHostToDevice, stream));
            WFitArgs fa;
            memset(&fa, 0, sizeof fa);
            fa.pass = pass_args(q);
            fa.prm = q;
            fa.pivoted = pivoted;
            fa.nfit = m;
            fa.always_fit = always_fit ? 1 : 0;
            fa.has_bounds = lupars != nullptr;
            fa.dtol = dtol;
            fa.starts = d_ms_starts;
            fa.lupars = lupars ? d_ms_lupars : nullptr;
            fa.states = d_ms_states;
            fa.records = d_ms_records;
            void *args[] = {(void *)&fa};
            if (hipModuleLaunchKernel(fn_fit[jm], m, 1, 1, WIDE_T, 1, 1, 0, stream, args, nullptr) != hipSuccess)
                return GSLNLS_E_NODEVICE;
            GSLNLS_HIP_OK(hipMemcpyAsync(records + (size_t)lo * K, d_ms_records, sizeof(double) * (size_t)m * K, hipMemcpyDeviceToHost, stream));
            GSLNLS_HIP_OK(hipStreamSynchronize(stream)); // (the next chunk reuses the buffers)
            if (g_interrupt_hook && g_interrupt_hook())
                return GSLNLS_E_INTERRUPTED;
        }
        return hipGetLastError() == hipSuccess ? GSLNLS_SUCCESS : GSLNLS_E_NODEVICE;
    }

    // ask for / pick up the fused kernel; a failed build leaves the three-launch chain in place (it is complete by itself)
    void bind_step(int jm, bool wait)
    {
        if (!fuse || fn_step[jm] || step_failed[jm])
            return;
        const std::string es = rtc_wide_step_expr(jm, PW);
        if (!rtc_step[jm] || wait)
            rtc_step[jm] = rtc_request(rtc_src, {es}, wait);
        const int st = rtc_step[jm]->state.load(std::memory_order_acquire);
        if (st == RTC_READY)
        {
            std::string msg;
            fn_step[jm] = rtc_function(*rtc_step[jm], es, msg);
            if (!fn_step[jm])
                step_failed[jm] = true;
        }
        else if (st == RTC_FAILED)
            step_failed[jm] = true;
    }

    WPassArgs pass_args(const LmParams &prm) const
    {
        WPassArgs a;
        a.x = d_x;
        a.y = d_y;
        a.sw = cur_sw;
        a.n = n;
        a.state = d_state;
        a.partials = d_partials;
        a.h_df = prm.h_df;
        a.h_fvv = prm.h_fvv;
        a.fvv_analytic = prm.fvv_analytic;
        a.wf_only = wf_only;
        a.fuse = WIDE_FUSE_NONE;
        a.G = G;
        a.spec = 0;
        a.spec_fault = 0;
        a.gsums = d_gsums;
        a.totals = d_totals;
        a.fb = d_fb;
        a.stamps = d_stamps;
        memset(&a.adv, 0, sizeof a.adv);
        return a;
    }

    // one pass at the state's trial point -> d_totals (sums_at, the gsl_nls_large operators): one launch, or two
    void launch_pass_totals(int jm, const LmParams &prm)
    {
        WPassArgs pa = pass_args(prm);
        const bool fused = fuse && fn_step[jm];
        pa.fuse = fused ? WIDE_FUSE_REDUCE : WIDE_FUSE_NONE;
        void *args[] = {(void *)&pa};
        (void)hipModuleLaunchKernel(fused ? fn_step[jm] : fn_pass[jm], G, 1, 1, WIDE_T, 1, 1, 0, stream, args, nullptr);
        if (!fused)
            hipLaunchKernelGGL(wide_reduce_kernel, dim3((NV + 63) / 64), dim3(WIDE_RED_T), 0, stream, d_partials, G, NV, NVP,
                               d_totals, d_state);
    }

    int launches_per_step(int jm) const { return fuse && fn_step[jm] ? 1 : 3; }

    // one trial step of trust_iterate_lu_LD (src/trust.c:445-546): rows -> sums -> LM step.  ONE launch: the workgroup
    // that completes the totals runs the step, workgroup 0 (no rows) solves ahead for the case that the step rejects.
    void launch_step(int jm, const LmParams &prm, WAdvanceArgs &adv, int index)
    {
        WPassArgs pa = pass_args(prm);
        adv.launch_idx = index;
        void *args[] = {(void *)&pa};
        if (fuse && fn_step[jm])
        {
            pa.fuse = WIDE_FUSE_STEP;
            pa.spec = spec;
            pa.spec_fault = spec_fault;
            pa.adv = adv;
            (void)hipModuleLaunchKernel(fn_step[jm], G + pa.spec, 1, 1, WIDE_T, 1, 1, 0, stream, args, nullptr);
            return;
        }
        (void)hipModuleLaunchKernel(fn_pass[jm], G, 1, 1, WIDE_T, 1, 1, 0, stream, args, nullptr);
        hipLaunchKernelGGL(wide_reduce_kernel, dim3((NV + 63) / 64), dim3(WIDE_RED_T), 0, stream, d_partials, G, NV, NVP, d_totals,
                           d_state);
        hipLaunchKernelGGL(wide_advance_kernel, dim3(1), dim3(WT_ADV), 0, stream, adv);
    }

    int run_loop(int jm, const LmParams &prm, const double *start, const double *lupars, bool trace, int chunk)
    {
        const int maxiter = prm.maxiter;
        if (trace)
        {
            if (trace_cap < maxiter + 1)
            {
                hipFree(d_ssrtrace);
                hipFree(d_partrace);
                GSLNLS_HIP_OK(hipMalloc(&d_ssrtrace, sizeof(double) * (maxiter + 1)));
                GSLNLS_HIP_OK(hipMalloc(&d_partrace, sizeof(double) * (size_t)(maxiter + 1) * p));
                trace_cap = maxiter + 1;
            }
            GSLNLS_HIP_OK(hipMemsetAsync(d_ssrtrace, 0xFF, sizeof(double) * (maxiter + 1), stream));
            GSLNLS_HIP_OK(hipMemsetAsync(d_partrace, 0xFF, sizeof(double) * (size_t)(maxiter + 1) * p, stream));
        }
        // lm_state_reset
        WState &s0 = h_state[0];
        memset(&s0, 0, sizeof(WState));
        for (int k = 0; k < p; ++k)
        {
            s0.x[k] = s0.xt[k] = start[k];
            s0.diag[k] = 1.0;
            const double lo = lupars ? lupars[2 * k] : -INFINITY, up = lupars ? lupars[2 * k + 1] : INFINITY;
            s0.lo[k] = std::isfinite(lo) ? lo : -INFINITY;
            s0.up[k] = std::isfinite(up) ? up : INFINITY;
        }
        s0.fnorm2 = INFINITY;
        s0.nu = 2.0;
        s0.chisq0 = s0.chisq1 = s0.chisq_init = INFINITY;
        s0.phase = PH_INIT;
        s0.status = ST_CONTINUE;
        s0.info = ST_CONTINUE;
        s0.p = p;
        const double t_begin = now_s();
        GSLNLS_HIP_OK(hipMemcpyAsync(d_state, &s0, sizeof(WState), hipMemcpyHostToDevice, stream));
        seq += 1;
        WAdvanceArgs adv;
        adv.pivoted = pivoted;
        adv.state = d_state;
        adv.totals = d_totals;
        adv.prm = prm;
        adv.ssrtrace = trace ? d_ssrtrace : nullptr;
        adv.partrace = trace ? d_partrace : nullptr;
        {
            void *dptr = nullptr;
            GSLNLS_HIP_OK(hipHostGetDevicePointer(&dptr, h_state, 0));
            adv.host_mirror = reinterpret_cast<WState *>(dptr) + 1;
            adv.done_seq = reinterpret_cast<unsigned int *>(reinterpret_cast<char *>(dptr) + sizeof(WState) * 2);
        }
        adv.seq = seq;
        adv.launch_idx = 0;
        adv.p = p;
        // steps are enqueued in chunks and the completion word is polled between them; what is enqueued beyond the step
        // that ends the fit still runs (three no-op launches each).  The first chunk is sized by the previous fit of the
        // same kind on this handle (IRLS re-solves, repeated fits end where the last one did), then small top-ups.
        const int kind = jm * 4 + prm.trs * 2 + (cur_sw ? 1 : 0);
        int next_chunk = 4;
        if (chunk <= 0)
            chunk = (pred_kind == kind && pred_steps > 0) ? pred_steps : 8;
        else
            next_chunk = chunk;
        const long long max_steps = ((long long)maxiter * 17 + 2) * (prm.trs ? 2 : 1) + 2 * (chunk + next_chunk);
        long long steps = 0;
        hipEventRecord(ev0, stream);
        for (;;)
        {
            for (int k = 0; k < chunk; ++k)
                launch_step(jm, prm, adv, (int)(steps + k));
            steps += chunk;
            chunk = next_chunk;
            if (hipGetLastError() != hipSuccess)
                return GSLNLS_E_NODEVICE;
            hipEventRecord(ev1, stream);
            bool done = false;
            for (;;)
            {
                if (*h_done == seq)
                {
                    done = true;
                    break;
                }
                const hipError_t q = hipEventQuery(ev1);
                if (q == hipSuccess)
                {
                    done = (*h_done == seq);
                    break;
                }
                if (q != hipErrorNotReady)
                    return GSLNLS_E_NODEVICE;
            }
            if (done)
                break;
            if (steps > max_steps)
                return GSLNLS_FAILURE;
            if (g_interrupt_hook && g_interrupt_hook())
            {
                (void)hipStreamSynchronize(stream);
                return GSLNLS_E_INTERRUPTED;
            }
        }
        __sync_synchronize();
        last_ms = (float)(1e3 * (now_s() - t_begin));
        last_launches = launches_per_step(jm) * steps;
        pred_kind = kind;
        pred_steps = h_state[1].end_launch + 1;
        if (pred_steps < 1 || pred_steps > 4096)
            pred_steps = 0;
        GSLNLS_HIP_OK(hipStreamSynchronize(stream)); // trailing steps (no-ops) + the event pair
        if (d_stamps && fuse && fn_step[jm])
        {
            const int ns = pred_steps > 0 && pred_steps < 1024 ? pred_steps : 0;
            std::vector<unsigned long long> hs((size_t)10 * 1024 + 16);
            (void)hipMemcpy(hs.data(), d_stamps, sizeof(unsigned long long) * hs.size(), hipMemcpyDeviceToHost);
            {
                const unsigned long long *r = &hs[(size_t)10 * 1024];
                fprintf(stderr, "[wide stamps] first row block, last pass: start -> rows evaluated %.2f us | contraction %.2f | the four wavefronts' sums side by side %.2f\n",
                        (double)(r[1] - r[0]) * 0.01, (double)(r[2] - r[1]) * 0.01, (double)(r[4] - r[2]) * 0.01);
                if (r[7] > 0)
                    fprintf(stderr, "[wide stamps] first row block, last pass, first wavefront: %llu tiles, row phase %.2f us per tile, contraction %.2f us per tile\n",
                            r[7], (double)r[5] * 0.01 / (double)r[7], (double)r[6] * 0.01 / (double)r[7]);
            }
            static const char *nm[8] = {"rows", "partial+arrive1", "reduce1", "arrive2", "reduce2", "advance_pre", "solve/spec", "post"};
            for (int kind = 0; kind < 2; ++kind)
            {
                double acc[8] = {0}, gap = 0;
                int cnt = 0;
                for (int k = 1; k < ns; ++k)
                {
                    const unsigned long long *o = &hs[(size_t)k * 10];
                    const bool rej = (o[9] >> 16) & 1;
                    if ((int)rej != kind || o[0] == 0)
                        continue;
                    for (int j = 0; j < 8; ++j)
                        acc[j] += (double)(o[j + 1] - o[j]) * 0.01;
                    gap += (double)(o[0] - hs[(size_t)(k - 1) * 10 + 8]) * 0.01;
                    cnt += 1;
                }
                if (cnt)
                {
                    fprintf(stderr, "[wide stamps] %s steps (%d): end of previous step -> start %.2f us |", kind ? "rejected" : "accepted", cnt, gap / cnt);
                    for (int j = 0; j < 8; ++j)
                        fprintf(stderr, " %s %.2f", nm[j], acc[j] / cnt);
                    fprintf(stderr, "\n");
                }
            }
        }
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess)
        {
            ev_ms_total += ms;
            ev_launches_total += launches_per_step(jm) * steps;
        }
        return GSLNLS_SUCCESS;
    }

    int solve(int jac, int fvv, const double *start, const double *lupars, const int *ci, const double *cd, int chunk,
              gslnls_result *out) override
    {
        if (ci[2] > 1)
            return GSLNLS_E_UNSUPPORTED; // dogleg family: not lowered
        if (fvv && prog->nfvv == 0)
            return GSLNLS_E_UNSUPPORTED; // second derivatives too large for the program: fvv = FALSE (finite differences)
        const int jm = jac ? 0 : (ci[5] ? 2 : 1);
        if (const int rc = bind(jm))
            return rc;
        const bool trace = ci[1] != 0 && out->ssrtrace && out->partrace;
        LmParams prm = make_params(ci, cd, jac, fvv, lupars != nullptr, cur_sw != nullptr);
        prm.chisq_in = NAN;
        int rc = run_loop(jm, prm, start, lupars, trace, chunk);
        if (rc)
            return rc;
        return pack(jm, prm, start, trace, out);
    }

    // resid / grad / covar / traces / scalars -> gslnls_result, like src/nls.c:648-753
    int pack(int jm, const LmParams &prm, const double *start, bool trace, gslnls_result *out)
    {
        const WState &s = h_state[1];
        const bool ok = (s.status == ST_SUCCESS || s.status == ST_EMAXITER);
        double t_fin = 0.0;
        if (ok && (out->resid || out->grad))
        {
            if (out->resid && !d_resid)
                GSLNLS_HIP_OK(hipMalloc(&d_resid, sizeof(double) * (size_t)n));
            if (out->grad && !d_grad)
                GSLNLS_HIP_OK(hipMalloc(&d_grad, sizeof(double) * (size_t)n * p));
            WPassArgs pa = pass_args(prm);
            double *r = out->resid ? d_resid : nullptr, *g = out->grad ? d_grad : nullptr;
            void *args[] = {(void *)&pa, (void *)&r, (void *)&g};
            int gf = (int)(((long long)n + 255) / 256);
            gf = gf > 2048 ? 2048 : (gf < 1 ? 1 : gf);
            if (prefault)
                prefault->before_d2h();
            t_fin = now_s();
            hipEventRecord(ev0, stream);
            (void)hipModuleLaunchKernel(fn_final[jm], gf, 1, 1, 256, 1, 1, 0, stream, args, nullptr);
            hipEventRecord(ev1, stream);
            if (out->resid)
                GSLNLS_HIP_OK(hipMemcpyAsync(out->resid, d_resid, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, stream));
            if (out->grad)
                GSLNLS_HIP_OK(hipMemcpyAsync(out->grad, d_grad, sizeof(double) * (size_t)n * p, hipMemcpyDeviceToHost, stream));
        }
        if (trace)
        {
            GSLNLS_HIP_OK(hipMemcpyAsync(out->ssrtrace, d_ssrtrace, sizeof(double) * (prm.maxiter + 1), hipMemcpyDeviceToHost, stream));
            GSLNLS_HIP_OK(hipMemcpyAsync(out->partrace, d_partrace, sizeof(double) * (size_t)(prm.maxiter + 1) * p,
                                         hipMemcpyDeviceToHost, stream));
        }
        GSLNLS_HIP_OK(hipStreamSynchronize(stream));
        if (t_fin > 0.0)
        {
            float fms = 0.f;
            (void)hipEventElapsedTime(&fms, ev0, ev1);
            g_call_prof.finalize_ms = fms;
            g_call_prof.d2h_ms = 1e3 * (now_s() - t_fin) - fms;
        }
        for (int k = 0; k < p; ++k)
            if (out->par)
                out->par[k] = ok ? s.x[k] : start[k];
        if (out->covar)
        {
            // gsl_multifit_nlinear_covar on the normal equations: (J^T J)^-1 of the final point (as gsl_nls_large's)
            bool good = ok;
            std::vector<double> A((size_t)p * p);
            if (good)
            {
                for (int i = 0; i < p; ++i)
                    for (int j = 0; j <= i; ++j)
                        A[(size_t)i * p + j] = A[(size_t)j * p + i] = s.A[i * (i + 1) / 2 + j];
                good = lg_chol(p, A);
                if (good)
                    lg_chol_invert(p, A);
            }
            for (int i = 0; i < p; ++i)
                for (int j = 0; j < p; ++j)
                    out->covar[i + (size_t)p * j] = good ? A[(size_t)i * p + j] : NAN;
        }
        if (!ok)
        {
            if (out->resid)
                for (int i = 0; i < n; ++i)
                    out->resid[i] = NAN;
            if (out->grad)
                for (size_t i = 0; i < (size_t)n * p; ++i)
                    out->grad[i] = NAN;
        }
        out->niter = s.niter;
        out->conv = s.status;
        out->info = s.info;
        out->ssr = s.chisq1;
        out->ssrtol = s.chisq0 - s.chisq1;
        out->neval[0] = s.nevalf;
        out->neval[1] = s.nevaldf;
        out->neval[2] = s.nevalfvv;
        out->chisq_init = s.chisq_init;
        out->loop_ms = last_ms;
        out->n_launches = (int)last_launches;
        out->n_steps = s.end_launch + 1;
        out->jtj_cond = ok ? scaled_jtj_cond(s.A, p) : NAN;
        out->code_path = 3;
        return s.status;
    }

    // test hook: the sums of ONE pass at theta (ssr, non-finite flag, packed lower J^T J, J^T f), NV doubles
    int sums_at(int jac, int fdtype, const double *theta, double *totals)
    {
        const int jm = jac ? 0 : (fdtype ? 2 : 1);
        if (const int rc = bind(jm))
            return rc;
        int ci[15] = {100, 0, 0, 0, 1, fdtype, 0, 0, 0, 0, 0, 0, 0, 1, 0};
        double cd[11] = {2, 3, 0.75, 1.4901161193847656e-08, 0.02, 1e-8, 1e-8, 1e-8, 0, 0, 0};
        LmParams prm = make_params(ci, cd, jac, 0, false, cur_sw != nullptr);
        WState &s0 = h_state[0];
        memset(&s0, 0, sizeof(WState));
        for (int k = 0; k < p; ++k)
            s0.x[k] = s0.xt[k] = theta[k];
        s0.phase = PH_TRIAL;
        s0.p = p;
        GSLNLS_HIP_OK(hipMemcpyAsync(d_state, &s0, sizeof(WState), hipMemcpyHostToDevice, stream));
        launch_pass_totals(jm, prm);
        GSLNLS_HIP_OK(hipMemcpyAsync(totals, d_totals, sizeof(double) * (size_t)NV, hipMemcpyDeviceToHost, stream));
        GSLNLS_HIP_OK(hipStreamSynchronize(stream));
        return hipGetLastError() == hipSuccess ? GSLNLS_SUCCESS : GSLNLS_E_NODEVICE;
    }

    // average milliseconds of one trial step (pass + reduce + advance) with the fit held alive at theta
    float time_pass(int jac, const double *theta, int reps) override
    {
        const int jm = jac ? 0 : 1;
        if (bind(jm) || reps < 1)
            return -1.f;
        int ci[15] = {1 << 30, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0};
        double cd[11] = {2, 3, 0.75, 1.4901161193847656e-08, 0.02, 1e-300, 1e-300, 0.0, 0, 0, 0};
        LmParams prm = make_params(ci, cd, jac, 0, false, cur_sw != nullptr);
        prm.bench_hold = 1;
        WState &s0 = h_state[0];
        memset(&s0, 0, sizeof(WState));
        for (int k = 0; k < p; ++k)
        {
            s0.x[k] = s0.xt[k] = theta[k];
            s0.diag[k] = 1.0;
            s0.lo[k] = -INFINITY;
            s0.up[k] = INFINITY;
        }
        s0.fnorm2 = INFINITY;
        s0.nu = 2.0;
        s0.chisq0 = s0.chisq1 = s0.chisq_init = INFINITY;
        s0.phase = PH_INIT;
        s0.status = s0.info = ST_CONTINUE;
        s0.p = p;
        if (hipMemcpyAsync(d_state, &s0, sizeof(WState), hipMemcpyHostToDevice, stream) != hipSuccess)
            return -1.f;
        seq += 1;
        WAdvanceArgs adv;
        memset(&adv, 0, sizeof adv);
        adv.pivoted = pivoted;
        adv.state = d_state;
        adv.totals = d_totals;
        adv.prm = prm;
        void *dptr = nullptr;
        if (hipHostGetDevicePointer(&dptr, h_state, 0) != hipSuccess)
            return -1.f;
        adv.host_mirror = reinterpret_cast<WState *>(dptr) + 1;
        adv.done_seq = reinterpret_cast<unsigned int *>(reinterpret_cast<char *>(dptr) + sizeof(WState) * 2);
        adv.seq = seq;
        adv.p = p;
        for (int k = 0; k < 4; ++k)
            launch_step(jm, prm, adv, k);
        hipEventRecord(ev0, stream);
        for (int k = 0; k < reps; ++k)
            launch_step(jm, prm, adv, 4 + k);
        hipEventRecord(ev1, stream);
        if (hipStreamSynchronize(stream) != hipSuccess)
            return -1.f;
        float ms = 0.f;
        hipEventElapsedTime(&ms, ev0, ev1);
        return ms / (float)reps;
    }

    int loop_event_stats(double *ms_total, long long *launches_total, int reset) override
    {
        if (ms_total)
            *ms_total = ev_ms_total;
        if (launches_total)
            *launches_total = ev_launches_total;
        if (reset)
        {
            ev_ms_total = 0.0;
            ev_launches_total = 0;
        }
        return 0;
    }
    int set_swts(const double *swts) override
    {
        if (!owns_data)
            return GSLNLS_E_UNSUPPORTED;
        if (!swts)
        {
            cur_sw = nullptr;
            return GSLNLS_SUCCESS;
        }
        if (!sw_owned)
        {
            GSLNLS_HIP_OK(hipMalloc(&d_sw, sizeof(double) * (size_t)n));
            sw_owned = true;
        }
        GSLNLS_HIP_OK(hipMemcpy(d_sw, swts, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
        cur_sw = d_sw;
        return GSLNLS_SUCCESS;
    }
    // gsl_multifit_nlinear_rho_driver (src/nls_irls.c:412-546) around the wide solve: the same driver as irls_host.hpp
    // (cold re-start from the ORIGINAL start with the current weights, radix-select median, psi family, stopping rule);
    // only the residual pass differs -- the formula's finalize kernel run without weights
    int irls(int jac, int fvv, const double *start, const double *lupars, const int *ci, const double *cd, int loss_rho,
             const double *loss_cc, gslnls_result *out) override
    {
        if (ci[2] > 1)
            return GSLNLS_E_UNSUPPORTED;
        if (fvv && prog->nfvv == 0)
            return GSLNLS_E_UNSUPPORTED;
        const int jm = jac ? 0 : (ci[5] ? 2 : 1);
        if (const int rc0 = bind(jm))
            return rc0;
        const bool trace = ci[1] != 0 && out->ssrtrace && out->partrace;
        const int irls_maxiter = ci[14];
        const double irls_xtol = cd[10];
        LossCfg Lc;
        Lc.rho = loss_rho;
        {
            static const int ncc[9] = {0, 1, 2, 1, 1, 1, 1, 3, 3};
            for (int k = 0; k < 3; ++k)
                Lc.cc[k] = (loss_rho >= 1 && loss_rho <= 8 && k < ncc[loss_rho]) ? loss_cc[k] : 0.0;
        }
        const size_t nb = sizeof(double) * (size_t)n;
        const double *user_sw = cur_sw;
        constexpr int TW = 256;
        int nblk = (int)(((long long)n + TW - 1) / TW);
        if (nblk > 1024)
            nblk = 1024;
        auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
        const size_t need = 6 * up(nb) + up(sizeof(unsigned long long) * (size_t)n) + up(sizeof(double) * nblk) +
                            up(sizeof(SelectState) * 2) + up(sizeof(IrlsScalars));
        if (irls_arena_bytes < need)
        {
            hipFree(irls_arena);
            irls_arena = nullptr;
            irls_arena_bytes = 0;
            GSLNLS_HIP_OK(hipMalloc(&irls_arena, need));
            irls_arena_bytes = need;
        }
        char *q = static_cast<char *>(irls_arena);
        auto take = [&](size_t b) {
            char *r = q;
            q += up(b);
            return r;
        };
        double *d_r = reinterpret_cast<double *>(take(nb)), *d_wt = reinterpret_cast<double *>(take(nb)),
               *d_psi = reinterpret_cast<double *>(take(nb)), *d_psip = reinterpret_cast<double *>(take(nb)),
               *d_swA = reinterpret_cast<double *>(take(nb)), *d_swB = reinterpret_cast<double *>(take(nb));
        unsigned long long *d_keys = reinterpret_cast<unsigned long long *>(take(sizeof(unsigned long long) * (size_t)n));
        double *d_part = reinterpret_cast<double *>(take(sizeof(double) * nblk));
        SelectState *d_sel = reinterpret_cast<SelectState *>(take(sizeof(SelectState) * 2));
        IrlsScalars *d_sc = reinterpret_cast<IrlsScalars *>(take(sizeof(IrlsScalars)));
        // the handle's weights are put back however this function is left
        struct RestoreSw
        {
            const double *&ref;
            const double *val;
            ~RestoreSw() { ref = val; }
        } restore_sw{cur_sw, user_sw};
        double *sw_now = d_swA, *sw_next = d_swB;
        if (user_sw)
            GSLNLS_HIP_OK(hipMemcpyAsync(sw_now, user_sw, nb, hipMemcpyDeviceToDevice, stream));
        else
        {
            std::vector<double> ones((size_t)n, 1.0);
            GSLNLS_HIP_OK(hipMemcpy(sw_now, ones.data(), nb, hipMemcpyHostToDevice));
        }
        std::vector<double> workp(p);
        int irls_iter = 0, irls_status = ST_FAILURE, status = ST_CONTINUE;
        double chisq_init = NAN, chisq_carry = NAN, sigma = 1.0;
        long long total_launches = 0;
        float total_ms = 0.f;
        LmParams prm = make_params(ci, cd, jac, fvv, lupars != nullptr, true);
        do
        {
            irls_iter += 1;
            if (irls_iter > 1)
            {
                std::swap(sw_now, sw_next);
                for (int k = 0; k < p; ++k)
                    workp[k] = h_state[1].x[k];
            }
            else
                for (int k = 0; k < p; ++k)
                    workp[k] = start[k];
            cur_sw = sw_now;
            prm.has_weights = 1;
            prm.chisq_in = (irls_iter > 1) ? chisq_carry : NAN;
            int rc = run_loop(jm, prm, start, lupars, trace, 0);
            if (rc)
                return rc;
            total_launches += last_launches;
            total_ms += last_ms;
            const WState &s = h_state[1];
            status = s.status;
            if (irls_iter == 1)
                chisq_init = s.chisq_init;
            chisq_carry = s.chisq1;
            trace_printf("IRLS iter: %3d, weighted ssr: %g, par: (", irls_iter, s.chisq1); // (src/nls_irls.c:466-472)
            trace_vector(s.x, p);
            if (status == ST_EBADFUNC || (status == ST_ENOPROG && irls_iter == 1))
                break;
            // ---- re-weighting chain, all on the device ----
            {
                WPassArgs pa = pass_args(prm);
                pa.sw = nullptr; // unweighted residuals r_i = model - y
                double *r = d_r, *g = nullptr;
                void *args[] = {(void *)&pa, (void *)&r, (void *)&g};
                int gf = (int)(((long long)n + 255) / 256);
                gf = gf > 2048 ? 2048 : (gf < 1 ? 1 : gf);
                (void)hipModuleLaunchKernel(fn_final[jm], gf, 1, 1, 256, 1, 1, 0, stream, args, nullptr);
                hipLaunchKernelGGL(wide_keys_kernel, dim3(gf), dim3(256), 0, stream, d_r, (long long)n, d_keys);
            }
            const unsigned long long k_lo = (unsigned long long)((n - 1) / 2), k_hi = (unsigned long long)(n / 2);
            const int nsel = (k_lo == k_hi) ? 1 : 2;
            for (int which = 0; which < nsel; ++which)
            {
                SelectState *st = d_sel + which;
                hipLaunchKernelGGL(select_init_kernel, dim3(1), dim3(1), 0, stream, st, which == 0 ? k_lo : k_hi);
                for (int pass = 7; pass >= 0; --pass)
                {
                    hipLaunchKernelGGL(select_hist_kernel, dim3(std::min(1024, (int)((n + 255) / 256))), dim3(256), 0, stream,
                                       d_keys, (long long)n, pass, st);
                    hipLaunchKernelGGL(select_pick_kernel, dim3(1), dim3(1), 0, stream, pass, st);
                }
            }
            hipLaunchKernelGGL(irls_sigma_kernel, dim3(1), dim3(1), 0, stream, d_sel, d_sel + (nsel - 1), d_sc);
            hipLaunchKernelGGL((irls_weight_kernel<TW>), dim3(nblk), dim3(TW), 0, stream, d_r, (long long)n, Lc, d_sc, d_wt,
                               d_psi, d_psip, d_part);
            hipLaunchKernelGGL(irls_scale_kernel, dim3(1), dim3(1), 0, stream, d_part, nblk, (long long)n, d_sc);
            hipLaunchKernelGGL((irls_apply_kernel<TW>), dim3(nblk), dim3(TW), 0, stream, d_wt, (long long)n, d_sc, user_sw,
                               sw_next);
            IrlsScalars hsc;
            GSLNLS_HIP_OK(hipMemcpyAsync(&hsc, d_sc, sizeof(hsc), hipMemcpyDeviceToHost, stream));
            GSLNLS_HIP_OK(hipStreamSynchronize(stream));
            sigma = hsc.sigma;
            // test_delta_irls (src/nls_irls.c:343-362)
            irls_status = ST_CONTINUE;
            for (int k = 0; k < p; ++k)
            {
                const double xi = s.x[k], dxi = fabs(workp[k] - xi);
                if (fmin(dxi / fabs(xi), dxi) < irls_xtol)
                    irls_status = ST_SUCCESS;
                else
                {
                    irls_status = ST_CONTINUE;
                    break;
                }
            }
            if (irls_status == ST_SUCCESS)
                break;
        } while (irls_status == ST_CONTINUE && irls_iter < irls_maxiter);
        if (!(status == ST_EBADFUNC || (status == ST_ENOPROG && irls_iter == 1)))
        {
            if (irls_iter >= irls_maxiter && irls_status != ST_SUCCESS)
            {
                irls_status = ST_EMAXITER;
                h_state[1].status = ST_EMAXITER;
                h_state[1].info = ST_EMAXITER;
            }
        }
        cur_sw = sw_now; // resid / grad are reported with the weights of the LAST solve (src/nls.c:695-737)
        h_state[1].chisq_init = chisq_init;
        last_launches = total_launches;
        last_ms = total_ms;
        const WState fin = h_state[1];
        int rc = pack(jm, prm, start, trace, out);
        const bool ok = (fin.status == ST_SUCCESS || fin.status == ST_EMAXITER);
        double irls_delta = 0.0;
        for (int k = 0; k < p; ++k)
            irls_delta = fmax(irls_delta, fabs(workp[k] - fin.x[k]));
        out->irls_sigma = sigma;
        out->irls_status = irls_status;
        out->irls_niter = irls_iter;
        out->irls_tol = irls_delta;
        if (ok)
        {
            if (out->irls_weights)
                GSLNLS_HIP_OK(hipMemcpy(out->irls_weights, d_wt, nb, hipMemcpyDeviceToHost));
            if (out->irls_psi)
                GSLNLS_HIP_OK(hipMemcpy(out->irls_psi, d_psi, nb, hipMemcpyDeviceToHost));
            if (out->irls_dpsi)
                GSLNLS_HIP_OK(hipMemcpy(out->irls_dpsi, d_psip, nb, hipMemcpyDeviceToHost));
        }
        else
            for (int i = 0; i < n; ++i)
            {
                if (out->irls_weights)
                    out->irls_weights[i] = NAN;
                if (out->irls_psi)
                    out->irls_psi[i] = NAN;
                if (out->irls_dpsi)
                    out->irls_dpsi[i] = NAN;
            }
        return rc;
    }
    // gsl_nls_large(formula) for 10 <= p <= 64: the operators of the large driver on the wide pass (WideLargeOps below)
    LargeOps *make_large_ops() override;
    // weighted residual at theta through the formula's finalize kernel (analytic-Jacobian unit)
    int residual_at(const double *theta, double *resid_host)
    {
        if (const int rc = bind(0))
            return rc;
        std::vector<double> tot((size_t)NV);
        int rc = sums_at(1, 0, theta, tot.data()); // (leaves x = theta in the device state)
        if (rc)
            return rc;
        if (!d_resid)
            GSLNLS_HIP_OK(hipMalloc(&d_resid, sizeof(double) * (size_t)n));
        int ci0[15] = {100, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0};
        double cd0[11] = {2, 3, 0.75, 1.4901161193847656e-08, 0.02, 1e-8, 1e-8, 1e-8, 0, 0, 0};
        LmParams prm = make_params(ci0, cd0, 1, 0, false, cur_sw != nullptr);
        WPassArgs pa = pass_args(prm);
        double *r = d_resid, *g = nullptr;
        void *args[] = {(void *)&pa, (void *)&r, (void *)&g};
        int gf = (int)(((long long)n + 255) / 256);
        gf = gf > 2048 ? 2048 : (gf < 1 ? 1 : gf);
        (void)hipModuleLaunchKernel(fn_final[0], gf, 1, 1, 256, 1, 1, 0, stream, args, nullptr);
        GSLNLS_HIP_OK(hipMemcpyAsync(resid_host, d_resid, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, stream));
        GSLNLS_HIP_OK(hipStreamSynchronize(stream));
        return hipGetLastError() == hipSuccess ? GSLNLS_SUCCESS : GSLNLS_E_NODEVICE;
    }
    // weighted residual, Jacobian (n x p, column-major) and (J^T J)^-1 at theta on the device; returns s^2 = ssr / (n - p).
    // 1: J^T J is singular (hat_values fails in the reference as well)
    int cooks_inputs(int jm, const double *theta, double **d_cinv_out, double *s2)
    {
        std::vector<double> tot((size_t)NV);
        int rc = sums_at(jm == 0, jm == 2, theta, tot.data());
        if (rc)
            return rc;
        std::vector<double> A((size_t)p * p);
        for (int i = 0; i < p; ++i)
            for (int j = 0; j <= i; ++j)
                A[(size_t)i * p + j] = A[(size_t)j * p + i] = tot[2 + i * (i + 1) / 2 + j];
        if (!lg_chol(p, A))
            return 1;
        lg_chol_invert(p, A);
        *s2 = tot[0] / (n - p);
        if (!d_resid)
            GSLNLS_HIP_OK(hipMalloc(&d_resid, sizeof(double) * (size_t)n));
        if (!d_grad)
            GSLNLS_HIP_OK(hipMalloc(&d_grad, sizeof(double) * (size_t)n * p));
        if (!d_cinv)
            GSLNLS_HIP_OK(hipMalloc(&d_cinv, sizeof(double) * WP * WP));
        GSLNLS_HIP_OK(hipMemcpyAsync(d_cinv, A.data(), sizeof(double) * (size_t)p * p, hipMemcpyHostToDevice, stream));
        GSLNLS_HIP_OK(hipStreamSynchronize(stream)); // (A is a local)
        int ci0[15] = {100, 0, 0, 0, 1, jm == 2, 0, 0, 0, 0, 0, 0, 0, 1, 0};
        double cd0[11] = {2, 3, 0.75, 1.4901161193847656e-08, 0.02, 1e-8, 1e-8, 1e-8, 0, 0, 0};
        LmParams prm = make_params(ci0, cd0, jm == 0, 0, false, cur_sw != nullptr);
        WPassArgs pa = pass_args(prm); // d_state holds x = theta (sums_at put it there)
        double *r = d_resid, *g = d_grad;
        void *args[] = {(void *)&pa, (void *)&r, (void *)&g};
        int gf = (int)(((long long)n + 255) / 256);
        gf = gf > 2048 ? 2048 : (gf < 1 ? 1 : gf);
        (void)hipModuleLaunchKernel(fn_final[jm], gf, 1, 1, 256, 1, 1, 0, stream, args, nullptr);
        *d_cinv_out = d_cinv;
        return 0;
    }

    // hat values and Cook's distances at theta (src/nls_utils.c:88-150; the S3 methods hatvalues() / cooks.distance())
    int diagnostics(int jac, const double *theta, const int *ci, const double *cd, double *hat, double *cooks) override
    {
        (void)cd;
        const int jm = jac ? 0 : (ci[5] ? 2 : 1);
        if (const int rc0 = bind(jm))
            return rc0;
        double *dc = nullptr, s2 = 0.0;
        int rc = cooks_inputs(jm, theta, &dc, &s2);
        if (rc)
            return rc == 1 ? GSLNLS_EINVAL : rc;
        double *d_d = nullptr, *d_h = nullptr;
        GSLNLS_HIP_OK(hipMalloc(&d_d, sizeof(double) * (size_t)n * 2));
        d_h = d_d + n;
        int gf = (int)(((long long)n + 255) / 256);
        gf = gf > 2048 ? 2048 : (gf < 1 ? 1 : gf);
        hipLaunchKernelGGL(wide_cooks_kernel, dim3(gf), dim3(256), 0, stream, d_resid, d_grad, (long long)n, p, dc, s2, d_d,
                           (unsigned long long *)nullptr, d_h);
        hipError_t e = hipSuccess;
        if (hat)
            e = hipMemcpyAsync(hat, d_h, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, stream);
        if (cooks && e == hipSuccess)
            e = hipMemcpyAsync(cooks, d_d, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess)
            e = hipStreamSynchronize(stream);
        (void)hipFree(d_d);
        return e == hipSuccess ? GSLNLS_SUCCESS : GSLNLS_E_NODEVICE;
    }

    // robust second pass of multi-start (src/nls.c:401-509): Cook's-distance outliers get zero weight.
    // returns 1 when a second pass has to run (d_sw_robust filled), 0 when not, < 0 on error
    int robust_weights(int jm, const double *mpopt, double *d_sw_robust)
    {
        double *dc = nullptr, s2 = 0.0;
        int rc = cooks_inputs(jm, mpopt, &dc, &s2);
        if (rc == 1)
            return 0; // cooks_d -> hat_values fails: no second pass (src/nls.c:419-421)
        if (rc)
            return rc < 0 ? rc : -1;
        // one allocation: distances, their keys, the two select states, the outlier count
        auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
        const size_t nb8 = up(sizeof(double) * (size_t)n);
        char *arena = nullptr;
        GSLNLS_HIP_OK(hipMalloc(&arena, 2 * nb8 + up(sizeof(SelectState) * 2) + 256));
        double *d_d = reinterpret_cast<double *>(arena);
        unsigned long long *d_keys = reinterpret_cast<unsigned long long *>(arena + nb8);
        SelectState *d_sel = reinterpret_cast<SelectState *>(arena + 2 * nb8);
        int *d_cnt = reinterpret_cast<int *>(arena + 2 * nb8 + up(sizeof(SelectState) * 2));
        if (hipMemsetAsync(d_cnt, 0, sizeof(int), stream) != hipSuccess)
        {
            (void)hipFree(arena);
            return GSLNLS_E_NODEVICE;
        }
        int gf = (int)(((long long)n + 255) / 256);
        gf = gf > 2048 ? 2048 : (gf < 1 ? 1 : gf);
        hipLaunchKernelGGL(wide_cooks_kernel, dim3(gf), dim3(256), 0, stream, d_resid, d_grad, (long long)n, p, dc, s2, d_d,
                           d_keys, (double *)nullptr);
        double med = 0.0, med2 = 0.0;
        rc = device_median(stream, d_keys, n, d_sel, &med);
        if (!rc)
        {
            hipLaunchKernelGGL(absdev_keys_kernel, dim3(gf), dim3(256), 0, stream, d_d, (long long)n, med, d_keys);
            rc = device_median(stream, d_keys, n, d_sel, &med2);
        }
        int noutlier = 0;
        if (!rc)
        {
            const double mad = 1.482602218505602 * med2;
            const double thresh = fmin(4.0 / n, 5 * mad);
            hipLaunchKernelGGL(outlier_weights_kernel, dim3(gf), dim3(256), 0, stream, d_d, (long long)n, thresh, cur_sw,
                               d_sw_robust, d_cnt);
            (void)hipMemcpyAsync(&noutlier, d_cnt, sizeof(int), hipMemcpyDeviceToHost, stream);
            (void)hipStreamSynchronize(stream);
        }
        (void)hipFree(arena);
        if (rc)
            return rc < 0 ? rc : -1;
        return (noutlier > 0 && noutlier < (n - p)) ? 1 : 0;
    }
    // multi-start branch of C_nls (src/nls.c:274-532): the host driver of mstart_driver.hpp (the reference's commit
    // order) around WideMsEvaluator, the robust second pass when a loss function is set, then the final solve.
    int mstart(int jac, int fvv, const double *start2p, const double *lupars, const int *ci, const double *cd,
               const int *has_start, const MsComm &comm, int loss_rho, const double *loss_cc, gslnls_result *out) override
    {
        if (ci[2] > 1)
            return GSLNLS_E_UNSUPPORTED;
        if (fvv && prog->nfvv == 0)
            return GSLNLS_E_UNSUPPORTED;
        const int jm = jac ? 0 : (ci[5] ? 2 : 1);
        if (const int rc0 = bind(jm))
            return rc0;
        MsState m;
        ms_init(m, p, ci, cd, start2p, has_start, lupars);
        WideMsEvaluator ev(*this);
        sobol_build(ev.tab, p);
        ev.prm = make_params(ci, cd, jac, fvv, lupars != nullptr, cur_sw != nullptr);
        ev.jm = jm;
        ev.lupars = lupars;
        int rc = ms_major_loop(m, ev, comm, start2p);
        if (rc)
            return rc < 0 && rc > -100 ? GSLNLS_FAILURE : rc;
        // robust second pass (src/nls.c:401-509): Cook's-distance outliers get zero weight, then the whole multi-start is
        // repeated (fresh counters and quasi-random sequence, ranges and exponents carried over)
        if (loss_rho != 0)
        {
            if (m.mssropt[1] < m.mssropt[0])
                m.mpopt = m.mpopt1;
            double *d_sw_robust = nullptr;
            GSLNLS_HIP_OK(hipMalloc(&d_sw_robust, sizeof(double) * (size_t)n));
            const int second = robust_weights(jm, m.mpopt.data(), d_sw_robust);
            if (second < 0)
            {
                (void)hipFree(d_sw_robust);
                return second;
            }
            if (second == 1)
            {
                const double *keep_sw = cur_sw;
                cur_sw = d_sw_robust;
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
                cur_sw = keep_sw; // "reset original weights" (src/nls.c:490-507)
            }
            (void)hipFree(d_sw_robust);
            if (rc)
                return rc < 0 && rc > -100 ? GSLNLS_FAILURE : rc;
        }
        ms_trace_finished(m);
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
                m.mpopt[0] = fmin(m.mpopt[0] + 1.0e-4, std::isfinite(lupars[1]) ? lupars[1] : INFINITY);
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
    // one concentration batch of fresh Sobol points, records to host memory (the per-point work of the driver, exposed
    // like the p <= 9 form for tests; no sharding: lo / hi select the points this call computes)
    int mstart_batch(int jac, const double *ranges, const double *kd, long long first_draw, int count, int lo, int hi,
                     int maxiter, double dtol, const int *ci, const double *cd, const double *lupars, double *records,
                     int records_on_device, float *kernel_ms) override
    {
        if (records_on_device || !records || lo < 0)
            return GSLNLS_E_UNSUPPORTED;
        const int jm = jac ? 0 : (ci[5] ? 2 : 1);
        if (const int rc0 = bind(jm))
            return rc0;
        WideMsEvaluator ev(*this);
        sobol_build(ev.tab, p);
        ev.prm = make_params(ci, cd, jac, 0, lupars != nullptr, cur_sw != nullptr);
        ev.jm = jm;
        ev.lupars = lupars;
        MsBatch b;
        b.count = count;
        b.p = p;
        b.K = 3 * p + 8;
        b.draw.resize(count);
        for (int i = 0; i < count; ++i)
            b.draw[i] = first_draw + i;
        b.start.assign((size_t)count * p, 0.0);
        b.range.assign(ranges, ranges + 2 * p);
        b.kd.resize(p);
        for (int k = 0; k < p; ++k)
            b.kd[k] = kd ? kd[k] : 0.75;
        b.maxiter = maxiter;
        b.dtol = dtol;
        b.always_fit = 0;
        const double t0 = now_s();
        const int rc = ev.run(b, lo, hi, records, false);
        if (kernel_ms)
            *kernel_ms = (float)(1e3 * (now_s() - t0));
        return rc;
    }
    int debug_stamps(int, const double *, int, unsigned long long *, int *) override { return GSLNLS_E_UNSUPPORTED; }
};

// The operators of gsl_nls_large (large_host.hpp: trust_iterate of gsl_multilarge_nlinear around eval / J^T J u) on a
// formula with 10 <= p <= 64: one EVAL is one wide pass (J^T J on the matrix cores) + reduce; with J^T J of the current
// point at hand the products J^T J u and ||J u||^2 = u^T J^T J u of the Steihaug-Toint iterations are p x p work on the
// host -- no further pass over the rows (the reference's R closures evaluate J u and J^T v on the full Jacobian).
struct WideLargeOps : LargeOps
{
    WideFit &fit;
    std::vector<double> tot, jcur, jtrial; // sums of the last pass; packed lower J^T J at the current / trial point
    bool have_cur = false;
    explicit WideLargeOps(WideFit &f) : fit(f), tot((size_t)f.NV)
    {
        // weights scale f only: the reference's gsl_df_large never weights J (src/nls_large.c:629-633), as in
        // large_row_kernel for the hand-written models (the handle serves the large driver from here on)
        f.wf_only = 1;
        n = f.n;
        p = f.p;
        const size_t na = (size_t)p * (p + 1) / 2;
        jcur.assign(na, 0.0);
        jtrial.assign(na, 0.0);
    }
    int eval(const double *x, double *ssr, double *g, double *diag, double *jtj, double *bad) override
    {
        const int rc = fit.sums_at(1, 0, x, tot.data());
        if (rc)
            return rc;
        ++npass;
        const int NA = p * (p + 1) / 2;
        *ssr = tot[0];
        *bad = tot[1];
        const double *A = tot.data() + 2, *gg = A + NA;
        for (int i = 0; i < p; ++i)
        {
            g[i] = gg[i];
            diag[i] = A[tri(i, i)];
            if (jtj)
                for (int j = 0; j <= i; ++j)
                    jtj[(size_t)i * p + j] = jtj[(size_t)j * p + i] = A[tri(i, j)];
        }
        std::copy(A, A + NA, jtrial.begin());
        if (!have_cur)
        {
            jcur = jtrial; // (the starting point is the current point)
            have_cur = true;
        }
        return 0;
    }
    void accept() override { jcur = jtrial; }
    int jtjv(const double *, const double *u, double *normw2, double *out) override
    {
        double nw = 0.0;
        for (int i = 0; i < p; ++i)
        {
            double s = 0.0;
            for (int j = 0; j < p; ++j)
                s += jcur[j <= i ? tri(i, j) : tri(j, i)] * u[j];
            out[i] = s;
            nw += u[i] * s;
        }
        *normw2 = nw;
        return 0;
    }
    int full_jtj(const double *xcur, double *jtj) override
    {
        std::vector<double> g(p), dg(p);
        double ssr, bad;
        const std::vector<double> keep = jtrial;
        const int rc = eval(xcur, &ssr, g.data(), dg.data(), jtj, &bad);
        jtrial = keep;
        return rc;
    }
    int residual(const double *xcur, double *resid_host) override { return fit.residual_at(xcur, resid_host); }
};
inline LargeOps *WideFit::make_large_ops() { return new WideLargeOps(*this); }

inline void *WideMsEvaluator::stream() { return fit.stream; }
inline int WideMsEvaluator::fetch(const double *src, bool src_on_device, double *dst, size_t nd)
{
    if (!nd)
        return 0;
    if (!src_on_device)
    {
        memcpy(dst, src, sizeof(double) * nd);
        return 0;
    }
    GSLNLS_HIP_OK(hipMemcpyAsync(dst, src, sizeof(double) * nd, hipMemcpyDeviceToHost, fit.stream));
    GSLNLS_HIP_OK(hipStreamSynchronize(fit.stream));
    return 0;
}
inline int WideMsEvaluator::fetch_stream(const double *dev_src, double *dst, size_t nd)
{
    if (nd)
        GSLNLS_HIP_OK(hipMemcpyAsync(dst, dev_src, sizeof(double) * nd, hipMemcpyDeviceToHost, fit.stream));
    GSLNLS_HIP_OK(hipStreamSynchronize(fit.stream));
    return 0;
}
inline int WideMsEvaluator::poke(double *dev_dst, double value)
{
    static double slot[4];
    slot[0] = value;
    GSLNLS_HIP_OK(hipMemcpyAsync(dev_dst, slot, sizeof(double), hipMemcpyHostToDevice, fit.stream));
    GSLNLS_HIP_OK(hipStreamSynchronize(fit.stream));
    return 0;
}
inline int WideMsEvaluator::run_async(MsBatch &b, int lo, int hi, double *dev_out)
{
    const size_t nd = (size_t)(hi - lo) * (3 * fit.p + 8);
    staged.assign(nd, 0.0);
    const int rc = run_host(b, lo, hi, staged.data());
    if (rc)
        return rc;
    // (`staged` lives as long as the evaluator: the copy is complete before the collective that follows it on the stream)
    GSLNLS_HIP_OK(hipMemcpyAsync(dev_out, staged.data(), sizeof(double) * nd, hipMemcpyHostToDevice, fit.stream));
    return 0;
}
inline int WideMsEvaluator::run(MsBatch &b, int lo, int hi, double *out, bool out_on_device)
{
    if (!out)
        return GSLNLS_E_UNSUPPORTED; // (records are produced on the host: somebody has to take them)
    if (!out_on_device)
        return run_host(b, lo, hi, out);
    const int rc = run_async(b, lo, hi, out);
    if (rc)
        return rc;
    GSLNLS_HIP_OK(hipStreamSynchronize(fit.stream));
    return 0;
}
inline int WideMsEvaluator::run_host(MsBatch &b, int lo, int hi, double *out)
{
    const int p = fit.p, K = 3 * p + 8, NA = p * (p + 1) / 2;
    std::vector<double> st(p), tot((size_t)fit.NV);
    if (hi > lo && fit.bind_fit(jm))
    {
        // every point of the block at once, one workgroup per point (wide_fit_kernel)
        std::vector<double> starts((size_t)(hi - lo) * p);
        for (int idx = lo; idx < hi; ++idx)
            for (int k = 0; k < p; ++k)
                starts[(size_t)(idx - lo) * p + k] =
                    b.draw[idx] >= 0 ? sobol_to_range(sobol_coord(tab, (unsigned int)b.draw[idx], k), b.range[2 * k], b.range[2 * k + 1], b.kd[k])
                                     : b.start[(size_t)idx * p + k];
        LmParams q = prm;
        q.maxiter = b.maxiter;
        q.gtol = 1e-3; // src/nls_mstart.c:91, :254
        q.chisq_in = NAN;
        return fit.fit_batch(jm, q, starts.data(), hi - lo, lupars, b.always_fit, b.dtol, out);
    }
    for (int idx = lo; idx < hi; ++idx)
    {
        for (int k = 0; k < p; ++k)
            st[k] = b.draw[idx] >= 0 ? sobol_to_range(sobol_coord(tab, (unsigned int)b.draw[idx], k), b.range[2 * k],
                                                       b.range[2 * k + 1], b.kd[k])
                                     : b.start[(size_t)idx * p + k];
        double *rec = out + (size_t)(idx - lo) * K;
        double *rx = rec, *rdiag = rec + p, *rx0 = rec + 2 * p, *sc = rec + 3 * p;
        // det_eval_jtj at the sampled point (src/nls_utils.c:23-53): one pass
        int rc = fit.sums_at(jm == 0, jm == 2, st.data(), tot.data());
        if (rc)
            return rc;
        double det0 = wide_det_cholesky(p, tot.data() + 2);
        if (jm == 0 && !(tot[1] == 0.0))
            det0 = 0.0; // eval_df failed (src/nls_utils.c:47-48)
        for (int k = 0; k < p; ++k)
            rx0[k] = st[k];
        sc[2] = det0;
        sc[4] = tot[0];
        if (b.always_fit || det0 > b.dtol)
        {
            LmParams q = prm;
            q.maxiter = b.maxiter;
            q.gtol = 1e-3; // src/nls_mstart.c:91, :254
            q.chisq_in = NAN;
            rc = fit.run_loop(jm, q, st.data(), lupars, false, 0);
            if (rc)
                return rc;
            const WState &s = fit.h_state[1];
            for (int k = 0; k < p; ++k)
            {
                rx[k] = s.x[k];
                rdiag[k] = s.diag[k];
            }
            sc[0] = s.chisq0;
            sc[1] = s.chisq1;
            sc[3] = wide_det_cholesky(p, s.A);
            sc[5] = (double)s.niter;
            sc[6] = (double)s.status;
            sc[7] = (double)s.nevalf;
            (void)NA;
        }
        else
        {
            for (int k = 0; k < p; ++k)
            {
                rx[k] = st[k];
                rdiag[k] = 1.0;
            }
            sc[0] = INFINITY;
            sc[1] = tot[0];
            sc[3] = 0.0;
            sc[5] = 0.0;
            sc[6] = (double)ST_CONTINUE;
            sc[7] = 1.0;
        }
    }
    return 0;
}

} // namespace gslnls
