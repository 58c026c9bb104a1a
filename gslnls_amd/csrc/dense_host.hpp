// dense_host.hpp -- host orchestration of one grid-per-fit problem resident in HBM.
//
// Replaces the numeric part of C_nls_internal for a single start (src/nls.c:533-576,
// :598-608, :632-753): upload once, run the device-resident LM loop as a chain of
// lm_step_kernel launches, read back the p-sized state, then (optionally) materialise
// resid / grad / covar with one more kernel.  The host never computes any part of the
// algorithm; it only enqueues launches and polls the device's own status word.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <time.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include <thread>
#include <sys/mman.h>
#include "../../include/gslnls_core.h"
#include "dense_kernels.hpp"
#include "dense_persist.hpp"
#include "mstart_driver.hpp"

namespace gslnls
{

#define GSLNLS_HIP_OK(expr)                                                                           \
    do                                                                                                \
    {                                                                                                 \
        hipError_t e__ = (expr);                                                                      \
        if (e__ != hipSuccess)                                                                        \
        {                                                                                             \
            fprintf(stderr, "gslnls: HIP error %s at %s:%d\n", hipGetErrorString(e__), __FILE__, __LINE__); \
            return GSLNLS_E_NODEVICE;                                                                 \
        }                                                                                             \
    } while (0)

inline double now_s()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

inline LmParams make_params(const int *ci, const double *cd, int jac, int fvv, bool has_bounds, bool has_w)
{
    // decode exactly as src/nls.c:94-152 does
    LmParams prm;
    prm.maxiter = ci[0];
    prm.trs = (ci[2] == 1) ? 1 : 0;
    prm.scale = ci[3];
    prm.fdtype = ci[5] ? 1 : 0;
    prm.jac_analytic = jac ? 1 : 0;
    prm.fvv_analytic = fvv ? 1 : 0;
    prm.has_bounds = has_bounds ? 1 : 0;
    prm.has_weights = has_w ? 1 : 0;
    prm.bench_hold = 0;
    prm.chisq_in = NAN;
    prm.factor_up = cd[0];
    prm.factor_down = cd[1];
    prm.avmax = cd[2];
    prm.h_df = cd[3];
    prm.h_fvv = cd[4];
    prm.xtol = cd[5];
    prm.ftol = cd[6];
    prm.gtol = cd[7];
    return prm;
}

// 2-norm condition number of C = S A S, S = diag(A)^-1/2, A the packed lower triangle of J^T J at the final point:
// cyclic Jacobi rotations on the p x p matrix (p <= 12).  A diagnostic of the boundary (gslnls_result::jtj_cond,
// gslnls_solver_served), not a step of the algorithm.
inline double scaled_jtj_cond(const double *Ap, int P)
{
    std::vector<double> C((size_t)P * P);
    for (int i = 0; i < P; ++i)
        for (int j = 0; j <= i; ++j)
        {
            const double di = Ap[i * (i + 1) / 2 + i], dj = Ap[j * (j + 1) / 2 + j];
            if (!(di > 0.0) || !(dj > 0.0) || !std::isfinite(di) || !std::isfinite(dj))
                return INFINITY; // a zero column: singular
            const double v = Ap[i * (i + 1) / 2 + j] / (sqrt(di) * sqrt(dj));
            C[(size_t)i * P + j] = C[(size_t)j * P + i] = v;
        }
    for (int sweep = 0; sweep < 60; ++sweep)
    {
        double off = 0.0;
        for (int i = 0; i < P; ++i)
            for (int j = 0; j < i; ++j)
                off += C[(size_t)i * P + j] * C[(size_t)i * P + j];
        if (off < 1e-30)
            break;
        for (int q = 1; q < P; ++q)
            for (int r = 0; r < q; ++r)
            {
                const double apq = C[(size_t)q * P + r];
                if (apq == 0.0)
                    continue;
                const double theta = (C[(size_t)q * P + q] - C[(size_t)r * P + r]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
                for (int k = 0; k < P; ++k)
                {
                    const double akr = C[(size_t)k * P + r], akq = C[(size_t)k * P + q];
                    C[(size_t)k * P + r] = c * akr - sn * akq;
                    C[(size_t)k * P + q] = sn * akr + c * akq;
                }
                for (int k = 0; k < P; ++k)
                {
                    const double ark = C[(size_t)r * P + k], aqk = C[(size_t)q * P + k];
                    C[(size_t)r * P + k] = c * ark - sn * aqk;
                    C[(size_t)q * P + k] = sn * ark + c * aqk;
                }
            }
    }
    double emin = INFINITY, emax = 0.0;
    for (int i = 0; i < P; ++i)
    {
        const double e = C[(size_t)i * P + i];
        emin = fmin(emin, e);
        emax = fmax(emax, e);
    }
    return (emin > 0.0) ? emax / emin : INFINITY;
}

// where the milliseconds of the last one-shot gslnls_nls() went (gslnls_last_call_profile): the number .Call(C_nls)
// delivers is create + H2D + loop + finalize + D2H + destroy, not the resident loop alone
struct CallProfile
{
    double create_ms = 0, h2d_ms = 0, loop_ms = 0, finalize_ms = 0, d2h_ms = 0, destroy_ms = 0, total_ms = 0;
};
inline CallProfile g_call_prof;

// Result vectors of a one-shot call (resid: n, grad: n x p doubles) are as a rule pages their process has not touched yet --
// a large Rf_allocVector / malloc is a fresh mmap -- and a device-to-host copy into such pages runs at the speed of the
// page faults it raises (scripts/pcie_probe/fresh_pages.hip: 1-12 GB/s against 55 GB/s into resident pages).  A helper
// thread faults them in while the data uploads and the fit runs.  MADV_POPULATE_WRITE leaves the contents alone, so it may
// still be running when the copy starts; where the kernel does not know it (< 5.14) the pages are touched with an atomic
// `or 0` (a write fault that changes nothing), and the copy waits for that to finish.
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
struct OutputPrefault
{
    std::thread th;
    bool running = false, touching = false;
    static constexpr size_t MIN_BYTES = (size_t)1 << 20;
    // does this kernel know MADV_POPULATE_WRITE (Linux >= 5.14)?  asked once, on a page of our own
    static bool populate_supported()
    {
        static const bool ok = [] {
            void *q = mmap(nullptr, 4096, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (q == MAP_FAILED)
                return false;
            const bool r = madvise(q, 4096, MADV_POPULATE_WRITE) == 0;
            munmap(q, 4096);
            return r;
        }();
        return ok;
    }
    static void populate(char *p, size_t bytes, bool touch)
    {
        const uintptr_t a = ((uintptr_t)p + 4095) & ~(uintptr_t)4095, e = ((uintptr_t)p + bytes) & ~(uintptr_t)4095;
        if (e <= a)
            return;
        if (!touch && madvise((void *)a, e - a, MADV_POPULATE_WRITE) == 0)
            return;
        if (!touch)
            return; // (refused for this range: leave the pages to the copy)
        for (uintptr_t q = a; q < e; q += 4096)
            (void)__atomic_fetch_or((unsigned char *)q, (unsigned char)0, __ATOMIC_RELAXED);
    }
    void start(double *a, size_t na, double *b, size_t nb)
    {
        // Measured on the MI355X host (scripts/pcie_probe, gpurun_out r04a): populating 32 MB while the upload and the fit
        // run takes 6 ms instead of 1.5 (the populate and the driver's own page pinning serialise on the address space's
        // lock), and a copy that starts before it has finished drops from 3.6 ms to 10 ms.  Off unless asked for.
        static const bool on = getenv("GSLNLS_PREFAULT") && getenv("GSLNLS_PREFAULT")[0] == '1';
        if (!on || running)
            return;
        if (!a || na < MIN_BYTES)
            na = 0;
        if (!b || nb < MIN_BYTES)
            nb = 0;
        if (na + nb == 0)
            return;
        running = true;
        touching = !populate_supported();
        const bool touch = touching;
        th = std::thread([=] {
            if (na)
                populate((char *)a, na, touch);
            if (nb)
                populate((char *)b, nb, touch);
        });
    }
    void join()
    {
        if (running)
        {
            th.join();
            running = false;
        }
    }
    // the touching fallback modifies (rewrites) bytes: it must be over before a copy engine writes the same pages
    void before_d2h()
    {
        static const bool wait = getenv("GSLNLS_PREFAULT_WAIT") != nullptr;
        if (running && (wait || touching))
            join();
    }
    ~OutputPrefault() { join(); }
};

struct LargeOps; // large_host.hpp

struct DenseBase
{
    virtual ~DenseBase() {}
    virtual int solve(int jac, int fvv, const double *start, const double *lupars, const int *ci, const double *cd,
                      int chunk, gslnls_result *out) = 0;
    // robust loss: IRLS around the dense solve (src/nls_irls.c:412-546, src/nls.c:577-596)
    virtual int irls(int jac, int fvv, const double *start, const double *lupars, const int *ci, const double *cd,
                     int loss_rho, const double *loss_cc, gslnls_result *out) = 0;
    virtual float time_pass(int jac, const double *theta, int reps) = 0;
    virtual int loop_event_stats(double *ms_total, long long *launches_total, int reset) = 0;
    // Give the object back instead of destroying it: true = it went into its type's pool of parked problems (stream,
    // events, workspaces, pinned mirror, cached multi-start evaluator stay allocated for the next create of the same
    // model), false = the caller deletes it.
    virtual bool park() { return false; }
    virtual int set_swts(const double *swts) = 0;
    // the matrix-free operator of gsl_nls_large over this problem's resident data (large_host.hpp)
    virtual LargeOps *make_large_ops() = 0;
    // whatever must be on the device before kernels of this problem run (expression models: their program)
    virtual int prepare_device() { return 0; }
    virtual int diagnostics(int jac, const double *theta, const int *ci, const double *cd, double *hat, double *cooks) = 0;
    // multi-start branch of C_nls (src/nls.c:274-532) followed by the final single-start solve
    virtual int mstart(int jac, int fvv, const double *start2p, const double *lupars, const int *ci, const double *cd,
                       const int *has_start, const MsComm &comm, int loss_rho, const double *loss_cc,
                       gslnls_result *out) = 0;
    // one concentration batch of `count` fresh Sobol points (first_draw + i), shard [lo, hi)
    virtual int mstart_batch(int jac, const double *ranges, const double *kd, long long first_draw, int count, int lo,
                             int hi, int maxiter, double dtol, const int *ci, const double *cd, const double *lupars,
                             double *records, int records_on_device, float *kernel_ms) = 0;
    virtual int debug_stamps(int jac, const double *theta, int warm, unsigned long long *out, int *nrows) = 0;
    int n = 0, p = 0;
    OutputPrefault *prefault = nullptr; // set by gslnls_nls() around its solve: pack() consults it before the D2H copies
    const MsComm *batch_comm = nullptr; // communicator of mstart_batch's gathered form (set by the C entry point)
};

// optional hook polled between launch chunks / passes / multi-start batches (gslnls_set_interrupt_hook)
inline int (*g_interrupt_hook)(void) = nullptr;

template <class M>
struct DenseFit : DenseBase
{
    static constexpr int P = M::P;
    static constexpr int NV = PassSums<P>::NV;
    // wide workgroups while the accumulators fit in 128 VGPRs, narrow ones beyond that
    static constexpr int T = (NV <= 24) ? 512 : (NV <= 70 ? 256 : 128); // lds_red[NV * T] must fit 160 KB

    // interpreted expression models: slots the program needs when they fit the workgroup's LDS (set by VmDenseFit
    // before every fit), 0 = slot file in scratch memory
    int vm_lds_slots = 0;
    // kernels of this model compiled in process for one formula (rtc_host.hpp): when set (by VmDenseFit, per Jacobian
    // kind, before a fit) they are launched instead of the interpreter's -- same signature, same state, same sums
    hipFunction_t native_step[3] = {nullptr, nullptr, nullptr}, native_finalize[3] = {nullptr, nullptr, nullptr};
    // slots of T doubles beside the kernel's static LDS (sums, broadcast block, the state copy of the interpreted
    // models: 0.5 KB at p = 2, 2.5 KB at p = 12)
    static constexpr int VM_LDS_STATIC_MAX = 4 * 1024;
    static constexpr int VM_LDS_CAP = (160 * 1024 - VM_LDS_STATIC_MAX) / (T * 8);
    bool vm_lds_ready[3] = {false, false, false};

    DenseCtx<P> ctx;
    bool owns_data = false;
    double *d_x = nullptr, *d_y = nullptr, *d_sw = nullptr;
    double *d_partials = nullptr;
    LmState<P> *d_state = nullptr;
    LmState<P> *h_state = nullptr; // pinned + mapped: the device writes the final state here
    volatile unsigned int *h_done = nullptr; // pinned word the device sets to ctx.seq when a fit ends
    double *d_ssrtrace = nullptr, *d_partrace = nullptr;
    int trace_cap = 0;
    double *d_resid = nullptr, *d_grad = nullptr, *d_covar = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // device time of the launch loops: one event pair per fit on the library's stream, read back lazily (at the
    // next fit or when the totals are asked for) so that no fit waits for its own trailing launches
    static constexpr int EV_RING = 4;
    hipEvent_t ev_fit0[EV_RING] = {}, ev_fit1[EV_RING] = {};
    bool ev_pending[EV_RING] = {};
    long long ev_launches[EV_RING] = {};
    int ev_head = 0; // slot of the fit in progress / the most recent fit
    double ev_ms_total = 0.0;
    long long ev_launches_total = 0;
    // one-launch-per-fit path (dense_persist.hpp): granule buffers of the two hops, and whether this handle may use it
    u64_t *d_pbufs = nullptr;
    int persist_state = -1; // -1 not decided, 0 off (env, occupancy, or a timed-out fit), 1 on
    long long last_steps = 0;
    MsEvaluator *ms_eval = nullptr; // cached batch evaluator (Sobol table, device buffers)
    void *irls_arena = nullptr;     // work arrays of the robust re-weighting (irls_host.hpp), kept between calls
    size_t irls_arena_bytes = 0;
    long long cap_rows = 0;         // rows the owned data buffers hold
    bool sw_owned = false;
    int device_ordinal = -1;

    // One-shot calls (gslnls_nls == C_nls: create, fit, destroy) on small problems used to spend ~3 ms in
    // hipMalloc / hipHostMalloc / hipFree / stream and event creation around a fit of ~0.1 ms.  A destroyed problem
    // that owns its data is parked instead (at most POOL_MAX per model type and process) and the next create of
    // that model re-binds it: data re-uploaded into the buffers it has (re-allocated only when they are too small).
    static constexpr int POOL_MAX = 2;
    static std::vector<DenseFit<M> *> &pool()
    {
        static std::vector<DenseFit<M> *> v;
        return v;
    }
    static DenseFit<M> *acquire()
    {
        int dev = -1;
        (void)hipGetDevice(&dev);
        auto &v = pool();
        for (size_t k = 0; k < v.size(); ++k)
            if (v[k]->device_ordinal == dev)
            {
                DenseFit<M> *d = v[k];
                v.erase(v.begin() + (long)k);
                return d;
            }
        return nullptr;
    }
    bool park() override
    {
        if (M::ID > 100 || !owns_data || !stream || (int)pool().size() >= POOL_MAX)
            return false; // (natively lowered expression models live in their own shared objects: not re-bound)
        if ((size_t)cap_rows * (M::NX + 2) * sizeof(double) > ((size_t)256 << 20))
            return false; // parking is for the small problems whose cost is allocation; big buffers go back at once
        (void)hipStreamSynchronize(stream); // trailing launches of the last fit
        pool().push_back(this);
        return true;
    }
    static void trim_pool()
    {
        for (DenseFit<M> *d : pool())
            delete d;
        pool().clear();
    }

    int init(const gslnls_model *fn, const double *y, int n_, const double *swts)
    {
        p = P;
        const bool first = (stream == nullptr);
        if (first)
        {
            memset(&ctx, 0, sizeof(ctx));
            (void)hipGetDevice(&device_ordinal);
            GSLNLS_HIP_OK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
            GSLNLS_HIP_OK(hipEventCreate(&ev0));
            GSLNLS_HIP_OK(hipEventCreate(&ev1));
            for (int k = 0; k < EV_RING; ++k)
            {
                GSLNLS_HIP_OK(hipEventCreate(&ev_fit0[k]));
                GSLNLS_HIP_OK(hipEventCreate(&ev_fit1[k]));
            }
            // sized for the largest grid: the buffers outlive the problem they were first created for
            GSLNLS_HIP_OK(hipMalloc(&d_partials, sizeof(double) * 2 * NV * MAX_G));
            GSLNLS_HIP_OK(hipMalloc(&d_state, sizeof(LmState<P>) * 2));
            GSLNLS_HIP_OK(hipHostMalloc(&h_state, sizeof(LmState<P>) * 2 + 64, hipHostMallocMapped));
            GSLNLS_HIP_OK(hipMalloc(&d_covar, sizeof(double) * P * P));
            GSLNLS_HIP_OK(hipMalloc(&d_pbufs, sizeof(u64_t) * PersistBufs<P>::WORDS));
            GSLNLS_HIP_OK(hipMemset(d_pbufs, 0, sizeof(u64_t) * PersistBufs<P>::WORDS));
        }
        else
        {
            // a parked problem: nothing of the previous tenant may leak into this one
            for (int k = 0; k < EV_RING; ++k)
                ev_pending[k] = false;
            ev_ms_total = 0.0;
            ev_launches_total = 0;
            pred_kind = -1;
            pred_launches = 0;
            persist_state = -1;
            last_steps = 0;
            last_parity = 0;
            last_launches = 0;
            last_ms = 0.f;
            if (n_ != n)
            {
                // result vectors are allocated on demand for the current n
                hipFree(d_resid);
                hipFree(d_grad);
                d_resid = d_grad = nullptr;
            }
            const unsigned int keep_seq = ctx.seq; // the completion word keeps counting upwards
            memset(&ctx, 0, sizeof(ctx));
            ctx.seq = keep_seq;
        }
        n = n_;
        const size_t nb = sizeof(double) * (size_t)n;
        if (fn->x_on_device)
        {
            if (owns_data)
            {
                hipFree(d_x);
                hipFree(d_y);
                hipFree(d_sw);
                cap_rows = 0;
                sw_owned = false;
            }
            owns_data = false;
            d_x = const_cast<double *>(fn->x);
            d_y = const_cast<double *>(y);
            d_sw = const_cast<double *>(swts);
        }
        else
        {
            if (!owns_data || cap_rows < n)
            {
                if (owns_data)
                {
                    hipFree(d_x);
                    hipFree(d_y);
                    hipFree(d_sw);
                }
                d_x = d_y = d_sw = nullptr;
                sw_owned = false;
                GSLNLS_HIP_OK(hipMalloc(&d_x, nb * M::NX));
                GSLNLS_HIP_OK(hipMalloc(&d_y, nb));
                cap_rows = n;
            }
            owns_data = true;
            if (swts && !sw_owned)
            {
                GSLNLS_HIP_OK(hipMalloc(&d_sw, sizeof(double) * (size_t)cap_rows));
                sw_owned = true;
            }
            // the three uploads go out back to back on the problem's stream and are waited for once
            const double t_h2d = now_s();
            GSLNLS_HIP_OK(hipMemcpyAsync(d_x, fn->x, nb * M::NX, hipMemcpyHostToDevice, stream));
            GSLNLS_HIP_OK(hipMemcpyAsync(d_y, y, nb, hipMemcpyHostToDevice, stream));
            if (swts)
                GSLNLS_HIP_OK(hipMemcpyAsync(d_sw, swts, nb, hipMemcpyHostToDevice, stream));
            GSLNLS_HIP_OK(hipStreamSynchronize(stream));
            g_call_prof.h2d_ms = 1e3 * (now_s() - t_h2d);
        }
        for (int c = 0; c < M::NX; ++c)
            ctx.x[c] = d_x + (size_t)c * n;
        ctx.y = d_y;
        ctx.sw = swts ? d_sw : nullptr;
        ctx.n = n;
        // one workgroup per CU at most; fewer when the problem is small (>= 2 rows per thread)
        int G = (int)(((long long)n + 2LL * T - 1) / (2LL * T));
        if (G < 1)
            G = 1;
        if (G > MAX_G)
            G = MAX_G;
        ctx.G = G;
        ctx.partials[0] = d_partials;
        ctx.partials[1] = d_partials + (size_t)NV * G;
        ctx.state[0] = d_state;
        ctx.state[1] = d_state + 1;
        {
            void *dptr = nullptr;
            GSLNLS_HIP_OK(hipHostGetDevicePointer(&dptr, h_state, 0));
            ctx.host_mirror = reinterpret_cast<LmState<P> *>(dptr);
            h_done = reinterpret_cast<volatile unsigned int *>(reinterpret_cast<char *>(h_state) + sizeof(LmState<P>) * 2);
            ctx.done_seq = reinterpret_cast<unsigned int *>(reinterpret_cast<char *>(dptr) + sizeof(LmState<P>) * 2);
            if (first)
            {
                h_done[0] = 0;
                h_done[1] = 0;
                h_done[2] = 0;
                h_done[3] = 0;
                ctx.seq = 0;
            }
        }
        return GSLNLS_SUCCESS;
    }

    ~DenseFit() override
    {
        delete ms_eval;
        hipFree(irls_arena);
        if (owns_data)
        {
            hipFree(d_x);
            hipFree(d_y);
            hipFree(d_sw);
        }
        hipFree(d_partials);
        hipFree(d_pbufs);
        hipFree(d_state);
        if (h_state)
            hipHostFree(h_state);
        hipFree(d_ssrtrace);
        hipFree(d_partrace);
        hipFree(d_resid);
        hipFree(d_grad);
        hipFree(d_covar);
        if (ev0)
            hipEventDestroy(ev0);
        if (ev1)
            hipEventDestroy(ev1);
        for (int k = 0; k < EV_RING; ++k)
        {
            if (ev_fit0[k])
                hipEventDestroy(ev_fit0[k]);
            if (ev_fit1[k])
                hipEventDestroy(ev_fit1[k]);
        }
        if (stream)
            hipStreamDestroy(stream);
    }

    int set_swts(const double *swts) override
    {
        if (!owns_data)
            return GSLNLS_E_UNSUPPORTED;
        const size_t nb = sizeof(double) * (size_t)n;
        if (!swts)
        {
            ctx.sw = nullptr;
            return GSLNLS_SUCCESS;
        }
        if (!sw_owned)
        {
            GSLNLS_HIP_OK(hipMalloc(&d_sw, sizeof(double) * (size_t)cap_rows));
            sw_owned = true;
        }
        GSLNLS_HIP_OK(hipMemcpy(d_sw, swts, nb, hipMemcpyHostToDevice));
        ctx.sw = d_sw;
        return GSLNLS_SUCCESS;
    }

    // fresh: first launch of a fit -- the kernel builds the start state from ctx.sa instead of reading a previous one
    void launch_step(int jacmode, int parity, bool fresh = false, long long index = 0)
    {
        const dim3 grid(ctx.G), block(T);
        const LmState<P> *prev = ctx.state[parity ^ 1];
        const double *pp = ctx.partials[parity ^ 1];
        if (fresh)
            parity |= FRESH_LAUNCH;
        parity |= (int)((index & 0x0fffffff) << 2);
        if constexpr (M::ID == 100)
        {
            const int jn = jacmode == JAC_ANALYTIC ? 0 : (jacmode == JAC_FORWARD ? 1 : 2);
            if (native_step[jn])
            {
                const double *x0 = ctx.x[0], *yy = ctx.y, *ss = ctx.sw;
                long long nn = ctx.n;
                int gg = ctx.G;
                void *args[] = {(void *)&prev, (void *)&pp, (void *)&x0, (void *)&yy, (void *)&ss, (void *)&nn, (void *)&gg,
                                (void *)&parity, (void *)&ctx};
                (void)hipModuleLaunchKernel(native_step[jn], grid.x, 1, 1, block.x, 1, 1, 0, stream, args, nullptr);
                return;
            }
            if (vm_lds_slots > 0 && vm_lds_slots <= VM_LDS_CAP)
            {
                using ML = typename M::LdsTwin;
                const int rows_lds = vm_lds_slots > NV ? vm_lds_slots : NV;
                const size_t dyn = sizeof(double) * (size_t)rows_lds * T;
                const int jm = jacmode == JAC_ANALYTIC ? 0 : (jacmode == JAC_FORWARD ? 1 : 2);
                if (!vm_lds_ready[jm])
                {
                    // more than 64 KB of dynamic LDS has to be asked for once per kernel
                    const int cap = 160 * 1024 - VM_LDS_STATIC_MAX;
                    hipError_t e = hipSuccess;
                    if (jm == 0)
                        e = hipFuncSetAttribute((const void *)lm_step_kernel<ML, JAC_ANALYTIC, T>,
                                                hipFuncAttributeMaxDynamicSharedMemorySize, cap);
                    else if (jm == 1)
                        e = hipFuncSetAttribute((const void *)lm_step_kernel<ML, JAC_FORWARD, T>,
                                                hipFuncAttributeMaxDynamicSharedMemorySize, cap);
                    else
                        e = hipFuncSetAttribute((const void *)lm_step_kernel<ML, JAC_CENTER, T>,
                                                hipFuncAttributeMaxDynamicSharedMemorySize, cap);
                    if (e != hipSuccess)
                    {
                        vm_lds_slots = 0;         // (the scratch form below)
                        (void)hipGetLastError(); // not an error of the fit: do not leave it for the loop's check
                    }
                    vm_lds_ready[jm] = (e == hipSuccess);
                }
                if (vm_lds_slots > 0)
                {
                    if (jm == 0)
                        hipLaunchKernelGGL((lm_step_kernel<ML, JAC_ANALYTIC, T>), grid, block, dyn, stream, prev, pp, ctx.x[0],
                                           ctx.y, ctx.sw, ctx.n, ctx.G, parity, ctx);
                    else if (jm == 1)
                        hipLaunchKernelGGL((lm_step_kernel<ML, JAC_FORWARD, T>), grid, block, dyn, stream, prev, pp, ctx.x[0],
                                           ctx.y, ctx.sw, ctx.n, ctx.G, parity, ctx);
                    else
                        hipLaunchKernelGGL((lm_step_kernel<ML, JAC_CENTER, T>), grid, block, dyn, stream, prev, pp, ctx.x[0],
                                           ctx.y, ctx.sw, ctx.n, ctx.G, parity, ctx);
                    return;
                }
            }
        }
        switch (jacmode)
        {
        case JAC_ANALYTIC:
            hipLaunchKernelGGL((lm_step_kernel<M, JAC_ANALYTIC, T>), grid, block, 0, stream, prev, pp, ctx.x[0], ctx.y,
                               ctx.sw, ctx.n, ctx.G, parity, ctx);
            break;
        case JAC_FORWARD:
            hipLaunchKernelGGL((lm_step_kernel<M, JAC_FORWARD, T>), grid, block, 0, stream, prev, pp, ctx.x[0], ctx.y,
                               ctx.sw, ctx.n, ctx.G, parity, ctx);
            break;
        default:
            hipLaunchKernelGGL((lm_step_kernel<M, JAC_CENTER, T>), grid, block, 0, stream, prev, pp, ctx.x[0], ctx.y,
                               ctx.sw, ctx.n, ctx.G, parity, ctx);
            break;
        }
    }

    void launch_finalize(int jacmode, int parity, double *resid, double *grad, double *covar)
    {
        int Gf = (int)(((long long)n + T - 1) / T);
        if (Gf > 2048)
            Gf = 2048;
        if (Gf < 1)
            Gf = 1;
        const dim3 grid(Gf), block(T);
        if constexpr (M::ID == 100)
        {
            const int jn = jacmode == JAC_ANALYTIC ? 0 : (jacmode == JAC_FORWARD ? 1 : 2);
            if (native_finalize[jn])
            {
                void *args[] = {(void *)&ctx, (void *)&parity, (void *)&resid, (void *)&grad, (void *)&covar};
                (void)hipModuleLaunchKernel(native_finalize[jn], grid.x, 1, 1, block.x, 1, 1, 0, stream, args, nullptr);
                return;
            }
        }
        switch (jacmode)
        {
        case JAC_ANALYTIC:
            hipLaunchKernelGGL((lm_finalize_kernel<M, JAC_ANALYTIC, T>), grid, block, 0, stream, ctx, parity, resid,
                               grad, covar);
            break;
        case JAC_FORWARD:
            hipLaunchKernelGGL((lm_finalize_kernel<M, JAC_FORWARD, T>), grid, block, 0, stream, ctx, parity, resid,
                               grad, covar);
            break;
        default:
            hipLaunchKernelGGL((lm_finalize_kernel<M, JAC_CENTER, T>), grid, block, 0, stream, ctx, parity, resid,
                               grad, covar);
            break;
        }
    }

    // the resident kernel's geometry: PT row threads + one control wave per workgroup, PG workgroups
    static constexpr int PT = T;
    int PGrid() const { return ctx.G; }
    // dynamic LDS a workgroup of the resident kernel may take: 160 KB minus its static arrays
    static constexpr int persist_lds_budget() { return 160 * 1024 - (NV * PT * 8 + 2048); }
    // rows per thread that stay in LDS: all of them when the budget allows
    int PRows() const
    {
        const int nc = M::NX + 1 + (ctx.sw ? 1 : 0);
        const long long need = ((long long)n + (long long)PGrid() * PT - 1) / ((long long)PGrid() * PT);
        const long long fit = persist_lds_budget() / ((long long)PT * nc * 8);
        return (int)(need < fit ? need : fit);
    }
    bool lds_attr_set[3] = {false, false, false};
    // false: the kernel could not be given its LDS budget -- the caller takes the launch-per-step kernel instead
    bool launch_fit(int jacmode, PersistArgs pa)
    {
        const dim3 grid(PGrid()), block(PT);
        pa.rows_resident = PRows();
        {
            // Hops 1 and 3 as plain (workgroup-scope) stores that stop in the XCD's L2 rely on such stores being visible
            // to agent-scope loads of other CUs of the same XCD -- true of gfx950's shared L2, measured 3x faster, but
            // not something the HIP memory model promises: opt-in (GSLNLS_PERSIST_FAST=1), write-through by default
            static const int fast_env = getenv("GSLNLS_PERSIST_FAST") ? atoi(getenv("GSLNLS_PERSIST_FAST")) : 0;
            pa.fast = fast_env;
        }
        const size_t dyn = (size_t)pa.rows_resident * PT * (M::NX + 1 + (ctx.sw ? 1 : 0)) * 8;
        if (!lds_attr_set[jacmode])
        {
            // dynamic LDS beyond the default 64 KB has to be announced once per kernel
            if constexpr (PERSIST_BUILT)
            {
                const void *fn = jacmode == JAC_ANALYTIC ? (const void *)lm_fit_kernel<M, JAC_ANALYTIC, PT>
                               : jacmode == JAC_FORWARD ? (const void *)lm_fit_kernel<M, JAC_FORWARD, PT>
                                                        : (const void *)lm_fit_kernel<M, JAC_CENTER, PT>;
                if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, persist_lds_budget()) != hipSuccess)
                {
                    (void)hipGetLastError(); // not an error of the fit: the other kernel serves it
                    persist_state = 0;
                    return false;
                }
            }
            lds_attr_set[jacmode] = true;
        }
        if constexpr (PERSIST_BUILT)
        switch (jacmode)
        {
        case JAC_ANALYTIC:
            hipLaunchKernelGGL((lm_fit_kernel<M, JAC_ANALYTIC, PT>), grid, block, dyn, stream, ctx.x[0], ctx.y, ctx.sw, ctx.n,
                               (int)grid.x, pa, ctx);
            break;
        case JAC_FORWARD:
            hipLaunchKernelGGL((lm_fit_kernel<M, JAC_FORWARD, PT>), grid, block, dyn, stream, ctx.x[0], ctx.y, ctx.sw, ctx.n,
                               (int)grid.x, pa, ctx);
            break;
        default:
            hipLaunchKernelGGL((lm_fit_kernel<M, JAC_CENTER, PT>), grid, block, dyn, stream, ctx.x[0], ctx.y, ctx.sw, ctx.n,
                               (int)grid.x, pa, ctx);
            break;
        }
        return true;
    }

    // May this handle run whole fits in one launch?  All G workgroups have to be resident together (one per CU):
    // decided once per handle from the device's CU count and the kernel's occupancy; GSLNLS_PERSIST=0 turns it off.
    // interpreted expression models and p > 4 (whose state does not fit the control wave's registers) keep the launch-per-step kernel
    // (natively lowered expression models, ID > 100, neither: every extra kernel is seconds of hipcc in front of the
    // first fit of a formula)
    static constexpr bool PERSIST_BUILT = (M::ID < 100) && (P <= 4);
    bool persist_ok()
    {
        if constexpr (!PERSIST_BUILT)
            return false;
        else if (persist_state < 0)
        {
            persist_state = 0;
            // Measured on MI355X (profiles/r02_persist_*.txt): per trial step the resident kernel
            // takes 2.9 us on one workgroup (launch-per-step: 5.8), 5.9 vs 5.6 us on 8, 7.9 vs 6.3 us on 256 -- the
            // in-kernel all-reduce (three hops, 2.4-3 us) costs more than the kernel boundary it replaces (about 2.5).
            // So by default only problems that fit a couple of workgroups take it; GSLNLS_PERSIST=1 forces it for
            // every grid the device can hold, GSLNLS_PERSIST=0 turns it off.
            int maxg = 2;
            if (const char *e = getenv("GSLNLS_PERSIST"))
            {
                if (e[0] == '0')
                    return false;
                maxg = MAX_G;
            }
            int cus = 0, per_cu = 0;
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_ordinal) != hipSuccess)
                return false;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, lm_fit_kernel<M, JAC_ANALYTIC, PT>, PT, 0) != hipSuccess ||
                per_cu < 1)
                return false;
            // one workgroup per CU: the dispatcher deals the workgroups of a launch round-robin over idle CUs, so a
            // grid of at most `cus` workgroups on an otherwise idle device is resident as a whole; a device that is
            // busy with something else makes the bounded spins give up, and the fit then takes the other kernel
            if (PGrid() <= cus && PGrid() <= maxg)
                persist_state = 1;
        }
        return persist_state == 1;
    }

    static constexpr int PERSIST_CHUNK = 2048; // steps per launch: the host's interrupt hook gets its say in between

    // the whole fit in one launch (dense_persist.hpp); returns 1 when a bounded spin gave up and the caller has to
    // run the fit through the launch-per-step kernel instead
    int run_loop_persist(int jacmode, const double *start, const double *lupars, int max_steps_override = 0)
    {
        const int maxiter = ctx.prm.maxiter;
        if (ctx.ssrtrace)
        {
            GSLNLS_HIP_OK(hipMemsetAsync(d_ssrtrace, 0xFF, sizeof(double) * (maxiter + 1), stream));
            GSLNLS_HIP_OK(hipMemsetAsync(d_partrace, 0xFF, sizeof(double) * (size_t)(maxiter + 1) * P, stream));
        }
        ev_head = (ev_head + 1) % EV_RING;
        collect_fit_events(ev_head);
        const double t_begin = now_s();
        hipEventRecord(ev_fit0[ev_head], stream);
        ctx.seq += 1;
        {
            StartArgs<P> sa;
            for (int k = 0; k < P; ++k)
            {
                sa.start[k] = start[k];
                sa.lo[k] = lupars ? lupars[2 * k] : -INFINITY;
                sa.up[k] = lupars ? lupars[2 * k + 1] : INFINITY;
            }
            h_state[0].phase = PH_INIT;
            ctx.sa = sa;
        }
        PersistArgs pa;
        pa.bufs = d_pbufs;
        pa.tag_base = ((ctx.seq & 0x7ffffu) | 0x80000u) << 12;
        pa.step0 = 0;
        pa.max_steps = max_steps_override > 0 ? max_steps_override : PERSIST_CHUNK;
        if (max_steps_override <= 0)
        {
            // test hook: a small chunk exercises the leave-and-resume path of the resident kernel
            static const int chunk_env = getenv("GSLNLS_PERSIST_CHUNK") ? atoi(getenv("GSLNLS_PERSIST_CHUNK")) : 0;
            if (chunk_env > 0)
                pa.max_steps = chunk_env;
        }
        pa.resume = 0;
        const long long max_steps_total = ((long long)maxiter * 17 + 2) * (ctx.prm.trs ? 2 : 1) + 2;
        long long launches = 0;
        for (;;)
        {
            if (!launch_fit(jacmode, pa))
            {
                ev_pending[ev_head] = false;
                return 1; // no LDS budget for the resident kernel: same fit through the launch-per-step kernel
            }
            if (hipGetLastError() != hipSuccess)
                return GSLNLS_E_NODEVICE;
            launches += 1;
            hipEvent_t ev_chunk = ev_fit1[ev_head];
            hipEventRecord(ev_chunk, stream);
            bool done = false;
            for (;;)
            {
                if (*h_done == ctx.seq)
                {
                    done = true;
                    break;
                }
                const hipError_t q = hipEventQuery(ev_chunk);
                if (q == hipSuccess)
                {
                    done = (*h_done == ctx.seq);
                    break;
                }
                if (q != hipErrorNotReady)
                    return GSLNLS_E_NODEVICE; // a launch failure or device fault: do not spin on it
            }
            if (done)
                break;
            __sync_synchronize();
            if (h_done[2] == ctx.seq)
            {
                persist_state = 0; // the device is shared with something that keeps our workgroups apart
                ev_pending[ev_head] = false;
                return 1;
            }
            if (max_steps_override > 0)
                break; // timing mode: the requested number of steps has run
            pa.step0 += pa.max_steps;
            pa.resume = 1;
            if (pa.step0 > max_steps_total)
                return GSLNLS_FAILURE;
            if (g_interrupt_hook && g_interrupt_hook())
                return GSLNLS_E_INTERRUPTED;
        }
        __sync_synchronize();
        ev_pending[ev_head] = true;
        ev_launches[ev_head] = launches;
        last_ms = (float)(1e3 * (now_s() - t_begin));
        last_parity = 0;
        last_launches = launches;
        last_steps = (long long)h_done[1];
        return GSLNLS_SUCCESS;
    }

    // ---- pieces of one fit -------------------------------------------------------------------
    int last_parity = 0;
    long long last_launches = 0;
    int pred_kind = -1, pred_launches = 0; // launches the previous fit of this kind needed (adaptive first chunk)
    float last_ms = 0.f;

    int prepare(int jac, int fvv, const double *lupars, const int *ci, const double *cd, bool trace)
    {
        if (ci[2] > 1)
            return GSLNLS_E_UNSUPPORTED; // dogleg / ddogleg / subspace2D are not lowered
        if (fvv && !M::HAS_FVV)
            return GSLNLS_E_UNSUPPORTED;
        ctx.prm = make_params(ci, cd, jac, fvv, lupars != nullptr, ctx.sw != nullptr);
        const int maxiter = ctx.prm.maxiter;
        if (trace)
        {
            if (trace_cap < maxiter + 1)
            {
                hipFree(d_ssrtrace);
                hipFree(d_partrace);
                GSLNLS_HIP_OK(hipMalloc(&d_ssrtrace, sizeof(double) * (maxiter + 1)));
                GSLNLS_HIP_OK(hipMalloc(&d_partrace, sizeof(double) * (size_t)(maxiter + 1) * P));
                trace_cap = maxiter + 1;
            }
            ctx.ssrtrace = d_ssrtrace;
            ctx.partrace = d_partrace;
        }
        else
        {
            ctx.ssrtrace = nullptr;
            ctx.partrace = nullptr;
        }
        return GSLNLS_SUCCESS;
    }

    // read back one slot of the ring (waits for the fit's trailing launches if they have not drained yet)
    void collect_fit_events(int slot)
    {
        if (!ev_pending[slot])
            return;
        float ms = 0.f;
        if (hipEventSynchronize(ev_fit1[slot]) == hipSuccess &&
            hipEventElapsedTime(&ms, ev_fit0[slot], ev_fit1[slot]) == hipSuccess)
        {
            ev_ms_total += ms;
            ev_launches_total += ev_launches[slot];
        }
        ev_pending[slot] = false;
    }
    int loop_event_stats(double *ms_total, long long *launches_total, int reset) override
    {
        for (int k = 0; k < EV_RING; ++k)
            collect_fit_events(k);
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

    // device-resident LM loop from `start` with the current ctx (prm, sw, traces); on return the final
    // state is in h_state[0] (written by the device through the mapped mirror) and in ctx.state[last_parity]
    int run_loop(int jacmode, const double *start, const double *lupars, int chunk)
    {
        if (chunk == 0 && persist_ok())
        {
            const int rc = run_loop_persist(jacmode, start, lupars);
            if (rc != 1)
                return rc;
            // a spin gave up (workgroups not co-resident): same fit, launch-per-step kernel
        }
        const int maxiter = ctx.prm.maxiter;
        if (ctx.ssrtrace)
        {
            // NaN-fill so rows of iterations that never ran stay NA like the reference's allocMatrix
            GSLNLS_HIP_OK(hipMemsetAsync(d_ssrtrace, 0xFF, sizeof(double) * (maxiter + 1), stream));
            GSLNLS_HIP_OK(hipMemsetAsync(d_partrace, 0xFF, sizeof(double) * (size_t)(maxiter + 1) * P, stream));
        }
        ev_head = (ev_head + 1) % EV_RING;
        collect_fit_events(ev_head); // the slot's previous tenant finished EV_RING fits ago: no wait in practice
        const double t_begin = now_s();
        hipEventRecord(ev_fit0[ev_head], stream);
        ctx.seq += 1; // this fit's sequence number (0 is never used)
        // brand-new state is built on device in slot 1; the first step launch has parity 0 and reads slot 1
        {
            StartArgs<P> sa;
            for (int k = 0; k < P; ++k)
            {
                sa.start[k] = start[k];
                sa.lo[k] = lupars ? lupars[2 * k] : -INFINITY;
                sa.up[k] = lupars ? lupars[2 * k + 1] : INFINITY;
            }
            h_state[0].phase = PH_INIT;
            ctx.sa = sa;
        }
        // Launches are enqueued in chunks and the completion word is polled between them; what is enqueued beyond the
        // launch that ends the fit still runs (as ~2.7 us no-ops).  With the default chunking the first chunk is sized
        // by the previous fit of the same kind on this handle (repeated fits -- IRLS re-solves, bootstrap, a benchmark
        // loop -- end where the last one did), followed by small top-ups; an explicit chunk is taken as given.
        const int kind = jacmode * 8 + ctx.prm.trs * 2 + (ctx.sw ? 1 : 0);
        const bool adaptive = chunk <= 0;
        int next_chunk = 16;
        if (adaptive)
        {
            chunk = (pred_kind == kind && pred_launches > 0) ? pred_launches : 16;
            next_chunk = (pred_kind == kind && pred_launches > 0) ? 4 : 16;
        }
        else
            next_chunk = chunk;
        // upper bound on launches: every iteration may take 16 trials (x2 passes with acceleration) + init
        const long long max_launches = ((long long)maxiter * 17 + 2) * (ctx.prm.trs ? 2 : 1) + chunk + next_chunk;
        long long launches = 0;
        int parity = 0;
        for (;;)
        {
            for (int k = 0; k < chunk; ++k)
            {
                launch_step(jacmode, parity, launches == 0 && k == 0, launches + k);
                parity ^= 1;
            }
            launches += chunk;
            chunk = next_chunk;
            // the device stamps the fit's sequence number into pinned host memory when it ends; poll that word
            // (bounded) instead of draining the stream, so trailing launches overlap with the caller
            bool done = false;
            // (the same event closes the fit's timing interval: re-recorded after every chunk, the last one counts)
            hipEvent_t ev_chunk = ev_fit1[ev_head];
            hipEventRecord(ev_chunk, stream);
            for (;;)
            {
                if (*h_done == ctx.seq)
                {
                    done = true;
                    break;
                }
                const hipError_t q = hipEventQuery(ev_chunk);
                if (q == hipSuccess)
                {
                    done = (*h_done == ctx.seq);
                    break;
                }
                if (q != hipErrorNotReady)
                    return GSLNLS_E_NODEVICE; // a launch failure or device fault: do not spin on it
            }
            if (done)
                break;
            if (launches > max_launches)
                return GSLNLS_FAILURE;
            // between chunks: give the embedding application its say (R: R_CheckUserInterrupt)
            if (g_interrupt_hook && g_interrupt_hook())
            {
                (void)hipStreamSynchronize(stream);
                return GSLNLS_E_INTERRUPTED;
            }
        }
        __sync_synchronize();
        pred_kind = kind;
        pred_launches = (int)h_done[1] + 1;
        if (pred_launches < 1 || pred_launches > 4096)
            pred_launches = 0;
        ev_pending[ev_head] = true;
        ev_launches[ev_head] = launches;
        last_ms = (float)(1e3 * (now_s() - t_begin));
        last_parity = parity ^ 1;
        last_launches = launches;
        last_steps = launches;
        return GSLNLS_SUCCESS;
    }

    // resid / grad / covar / traces / scalars -> gslnls_result, like src/nls.c:648-753
    int pack(int jacmode, const double *start, gslnls_result *out, bool trace)
    {
        const LmState<P> &s = h_state[0];
        const int maxiter = ctx.prm.maxiter;
        const bool ok = (s.status == ST_SUCCESS || s.status == ST_EMAXITER);
        const bool want_vecs = ok && (out->resid || out->grad);
        double t_fin = 0.0;
        if (want_vecs || (ok && out->covar))
        {
            if (out->resid && !d_resid)
                GSLNLS_HIP_OK(hipMalloc(&d_resid, sizeof(double) * (size_t)n));
            if (out->grad && !d_grad)
                GSLNLS_HIP_OK(hipMalloc(&d_grad, sizeof(double) * (size_t)n * P));
            if (prefault)
                prefault->before_d2h();
            t_fin = now_s();
            hipEventRecord(ev0, stream);
            launch_finalize(jacmode, last_parity, out->resid ? d_resid : nullptr, out->grad ? d_grad : nullptr,
                            d_covar);
            hipEventRecord(ev1, stream);
            if (out->resid)
                GSLNLS_HIP_OK(hipMemcpyAsync(out->resid, d_resid, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost,
                                             stream));
            if (out->grad)
                GSLNLS_HIP_OK(hipMemcpyAsync(out->grad, d_grad, sizeof(double) * (size_t)n * P,
                                             hipMemcpyDeviceToHost, stream));
            if (out->covar)
                GSLNLS_HIP_OK(hipMemcpyAsync(out->covar, d_covar, sizeof(double) * P * P, hipMemcpyDeviceToHost,
                                             stream));
        }
        if (trace)
        {
            GSLNLS_HIP_OK(hipMemcpyAsync(out->ssrtrace, d_ssrtrace, sizeof(double) * (maxiter + 1),
                                         hipMemcpyDeviceToHost, stream));
            GSLNLS_HIP_OK(hipMemcpyAsync(out->partrace, d_partrace, sizeof(double) * (size_t)(maxiter + 1) * P,
                                         hipMemcpyDeviceToHost, stream));
        }
        if (trace || want_vecs || (ok && out->covar))
            GSLNLS_HIP_OK(hipStreamSynchronize(stream)); // only when something n- or trace-sized was requested
        if (t_fin > 0.0)
        {
            float fms = 0.f;
            (void)hipEventElapsedTime(&fms, ev0, ev1);
            g_call_prof.finalize_ms = fms;
            g_call_prof.d2h_ms = 1e3 * (now_s() - t_fin) - fms;
        }
        for (int k = 0; k < P; ++k)
            if (out->par)
                out->par[k] = ok ? s.x[k] : start[k];
        if (!ok)
        {
            if (out->covar)
                for (int k = 0; k < P * P; ++k)
                    out->covar[k] = NAN;
            if (out->resid)
                for (int i = 0; i < n; ++i)
                    out->resid[i] = NAN;
            if (out->grad)
                for (size_t i = 0; i < (size_t)n * P; ++i)
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
        out->n_steps = (int)last_steps;
        out->jtj_cond = ok ? scaled_jtj_cond(s.A, P) : NAN;
        {
            const int jn = jacmode == JAC_ANALYTIC ? 0 : (jacmode == JAC_FORWARD ? 1 : 2);
            out->code_path = M::ID < 100 ? 0 : ((M::ID == 100 && !native_step[jn]) ? 1 : 2);
        }
        return s.status;
    }

    int solve(int jac, int fvv, const double *start, const double *lupars, const int *ci, const double *cd, int chunk,
              gslnls_result *out) override
    {
        const bool trace = ci[1] != 0 && out->ssrtrace && out->partrace;
        int rc = prepare(jac, fvv, lupars, ci, cd, trace);
        if (rc)
            return rc;
        ctx.prm.chisq_in = NAN;
        const int jacmode = jac ? JAC_ANALYTIC : (ci[5] ? JAC_CENTER : JAC_FORWARD);
        rc = run_loop(jacmode, start, lupars, chunk);
        if (rc)
            return rc;
        return pack(jacmode, start, out, trace);
    }

    int irls(int jac, int fvv, const double *start, const double *lupars, const int *ci, const double *cd, int loss_rho,
             const double *loss_cc, gslnls_result *out) override;
    int sums_at(const double *theta, int jacmode, PassSums<P> &out);
    int robust_weights(int jacmode, const double *mpopt, double *d_sw_robust);
    LargeOps *make_large_ops() override;
    int diagnostics(int jac, const double *theta, const int *ci, const double *cd, double *hat, double *cooks) override;

    int mstart(int jac, int fvv, const double *start2p, const double *lupars, const int *ci, const double *cd,
               const int *has_start, const MsComm &comm, int loss_rho, const double *loss_cc, gslnls_result *out) override;
    int mstart_batch(int jac, const double *ranges, const double *kd, long long first_draw, int count, int lo, int hi,
                     int maxiter, double dtol, const int *ci, const double *cd, const double *lupars, double *records,
                     int records_on_device, float *kernel_ms) override;

    // diagnostic: per-wave s_memtime stamps of one steady-state launch (needs a -DGSLNLS_STAMPS build)
    int debug_stamps(int jac, const double *theta, int warm, unsigned long long *out, int *nrows) override
    {
        if (warm < 0)
        {
            // resident kernel: [workgroup][role][8] stamps of one step in the middle of a 32-step launch
            const int prow = PGrid() * 2;
            unsigned long long *dp = nullptr;
            GSLNLS_HIP_OK(hipMalloc(&dp, sizeof(unsigned long long) * 8 * prow));
            GSLNLS_HIP_OK(hipMemset(dp, 0, sizeof(unsigned long long) * 8 * prow));
            ctx.stamps = dp;
            const float t = time_pass(jac, theta, -32);
            ctx.stamps = nullptr;
            GSLNLS_HIP_OK(hipMemcpy(out, dp, sizeof(unsigned long long) * 8 * prow, hipMemcpyDeviceToHost));
            hipFree(dp);
            *nrows = prow;
            return t < 0 ? GSLNLS_FAILURE : GSLNLS_SUCCESS;
        }
        const int rows = ctx.G * (T / 64);
        unsigned long long *d = nullptr;
        GSLNLS_HIP_OK(hipMalloc(&d, sizeof(unsigned long long) * 8 * rows));
        GSLNLS_HIP_OK(hipMemset(d, 0, sizeof(unsigned long long) * 8 * rows));
        ctx.stamps = d;
        const float t = time_pass(jac, theta, warm);
        ctx.stamps = nullptr;
        GSLNLS_HIP_OK(hipMemcpy(out, d, sizeof(unsigned long long) * 8 * rows, hipMemcpyDeviceToHost));
        hipFree(d);
        *nrows = rows;
        return t < 0 ? GSLNLS_FAILURE : GSLNLS_SUCCESS;
    }

    float time_pass(int jac, const double *theta, int reps) override
    {
        // a state parked in PH_TRIAL at theta with huge mu: every launch performs the full
        // prologue + pass (the trial is rejected or accepted on device like any other)
        const int jacmode = jac ? JAC_ANALYTIC : JAC_FORWARD;
        int ci[15] = {1 << 30, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0};
        double cd[11] = {2, 3, 0.75, 1.4901161193847656e-08, 0.02, 1e-300, 1e-300, 0.0, 0, 0, 0};
        ctx.prm = make_params(ci, cd, jac, 0, false, ctx.sw != nullptr);
        ctx.prm.bench_hold = 1;
        ctx.ssrtrace = nullptr;
        ctx.partrace = nullptr;
        {
            StartArgs<P> sa;
            for (int k = 0; k < P; ++k)
            {
                sa.start[k] = theta[k];
                sa.lo[k] = -INFINITY;
                sa.up[k] = INFINITY;
            }
            ctx.sa = sa;
        }
        if (reps < 0 && persist_ok())
        {
            // resident kernel: -reps steps in one launch (bench_hold keeps the fit alive), average per step
            const double st[P > 0 ? P : 1] = {};
            (void)st;
            // (1 = a bounded spin gave up or the kernel could not be set up: no timing of the resident kernel exists)
            if (run_loop_persist(jacmode, theta, nullptr, 8) != 0) // warm-up launch
                return -1.f;
            (void)hipStreamSynchronize(stream);
            hipEventRecord(ev0, stream);
            if (run_loop_persist(jacmode, theta, nullptr, -reps) != 0)
                return -1.f;
            hipEventRecord(ev1, stream);
            if (hipStreamSynchronize(stream) != hipSuccess)
                return -1.f;
            float pms = 0.f;
            hipEventElapsedTime(&pms, ev0, ev1);
            ev_pending[ev_head] = false;
            return pms / (float)(-reps);
        }
        if (reps < 0)
            reps = -reps;
        int parity = 0;
        for (int k = 0; k < 4; ++k)
        {
            launch_step(jacmode, parity, k == 0);
            parity ^= 1;
        }
        if (getenv("GSLNLS_GRAPH_PROBE"))
        {
            // developer probe: the same launches replayed from a captured hipGraph (16 kernel nodes per replay)
            hipGraph_t graph = nullptr;
            hipGraphExec_t exec = nullptr;
            if (hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal) != hipSuccess)
                return -1.f;
            for (int k = 0; k < 16; ++k)
            {
                launch_step(jacmode, parity);
                parity ^= 1;
            }
            if (hipStreamEndCapture(stream, &graph) != hipSuccess ||
                hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess)
                return -1.f;
            const int replays = reps / 16 > 0 ? reps / 16 : 1;
            (void)hipGraphLaunch(exec, stream);
            hipEventRecord(ev0, stream);
            for (int r = 0; r < replays; ++r)
                (void)hipGraphLaunch(exec, stream);
            hipEventRecord(ev1, stream);
            if (hipStreamSynchronize(stream) != hipSuccess)
                return -1.f;
            float gms = 0.f;
            hipEventElapsedTime(&gms, ev0, ev1);
            (void)hipGraphExecDestroy(exec);
            (void)hipGraphDestroy(graph);
            return gms / (float)(replays * 16);
        }
        hipEventRecord(ev0, stream);
        for (int k = 0; k < reps; ++k)
        {
            launch_step(jacmode, parity);
            parity ^= 1;
        }
        hipEventRecord(ev1, stream);
        if (hipStreamSynchronize(stream) != hipSuccess)
            return -1.f;
        float ms = 0.f;
        hipEventElapsedTime(&ms, ev0, ev1);
        return ms / (float)reps;
    }
};

} // namespace gslnls
