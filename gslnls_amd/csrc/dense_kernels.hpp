// dense_kernels.hpp -- grid-per-fit kernels: one nonlinear least-squares problem whose
// n residual rows are spread over the whole chip (BASELINE config C2: n = 1e6, p = 3).
//
// One launch of lm_step_kernel == one trial step of trust_iterate_lu_LD
// (src/trust.c:445-546), with everything the reference does between two model
// evaluations folded into the front of the next pass:
//
//   launch t:  [prologue, every workgroup]  reduce the G partial sums of launch t-1 in a
//              fixed order -> one wavefront runs lm_advance() (rho, accept/reject, mu,
//              D, modified-Cholesky solve, convergence test, bound projection) -> the new
//              trial point is broadcast through LDS;
//              [pass]  every thread streams its rows of (x, y[, sqrt w]) once, computes
//              f_i and the Jacobian row in registers (analytic, forward or central FD)
//              and accumulates ssr, J^T J, J^T f; wavefront shuffle reduction -> LDS
//              -> one partial set per workgroup.
//
// Nothing n-sized is written and nothing crosses PCIe inside the loop.  The kernel
// boundary is the grid-wide barrier (cheaper on gfx950 than an in-kernel barrier:
// MI355X_MICROARCH.md "boundary" 1.45 us vs "barrier-xcd" 4.1 us).  State and partials
// are double-buffered by launch parity so no workgroup reads what another one writes
// in the same launch.  All reductions have a fixed shape => results are run-to-run
// bit-identical.
#pragma once
#if !defined(__HIPCC_RTC__)
#include <hip/hip_runtime.h>
#include <cstddef>
#endif
#include "lm_core.hpp"
#include "models.hpp"
#include "rowops.hpp"

namespace gslnls
{

// Diagnostic build only (-DGSLNLS_STAMPS): s_memtime stamps of the step kernel's sections go to a
// debug buffer that nothing else reads; the shipped library compiles them out.
#ifdef GSLNLS_STAMPS
#define GSLNLS_STAMP(slot)                                                                         \
    do                                                                                             \
    {                                                                                              \
        unsigned long long t__;                                                                    \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");                 \
        if (ctx.stamps && (threadIdx.x & 63) == 0)                                                 \
            ctx.stamps[((size_t)blockIdx.x * (T / 64) + (threadIdx.x >> 6)) * 8 + (slot)] = t__;   \
    } while (0)
#else
#define GSLNLS_STAMP(slot) \
    do                     \
    {                      \
    } while (0)
#endif

constexpr int NX_MAX = 4;
constexpr int MAX_G = 256; // workgroups per fit == partial sets (one per CU)

// Start of a fit: the brand-new state is built on device from kernel arguments (no H2D copy, no reset launch):
// the first step launch of a fit carries FRESH_LAUNCH in its parity argument and takes the start from ctx.sa.
template <int P>
struct StartArgs
{
    double start[P];
    double lo[P], up[P];
};
constexpr int FRESH_LAUNCH = 2; // bits 2.. of the same argument: index of the launch within its fit

template <int P>
struct DenseCtx
{
    const double *x[NX_MAX]; // regressor columns, each n contiguous doubles
    const double *y;
    const double *sw; // sqrt(weights) or nullptr
    long long n;
    int G; // workgroups == partial sets
    int fresh_parity; // unused by kernels; host bookkeeping
    double *partials[2];  // [NV][G]
    LmState<P> *state[2]; // ping-pong by launch parity
    LmParams prm;
    double *ssrtrace; // maxiter+1, or nullptr
    double *partrace; // (maxiter+1) x P column-major, or nullptr
    LmState<P> *host_mirror; // pinned host memory mapped into the device: final state lands here
    unsigned long long *stamps; // diagnostic builds only
    unsigned int *done_seq;      // pinned host word: sequence number of the last finished fit
    unsigned int seq;            // sequence number of this fit
    StartArgs<P> sa;             // read by the first launch of a fit only
};

// Cross-lane move of a double through DPP (VALU, ~8 cycles) instead of ds_bpermute (LDS, ~100
// cycles): the 64-lane sums below sit on the critical path of every launch.
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v)
{
    const long long bits = __double_as_longlong(v);
    int lo = (int)(bits & 0xffffffffll), hi = (int)(bits >> 32);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// wavefront (64 lanes) sum in a fixed order, result valid in every lane:
// butterflies inside each row of 16 lanes (quad_perm xor 1, xor 2, row_half_mirror, row_mirror),
// then the four row totals are combined as (r0 + r1) + (r2 + r3).
__device__ __forceinline__ double wave_sum(double v)
{
    v += dpp_mov<0xB1>(v);  // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);  // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v); // row_half_mirror
    v += dpp_mov<0x140>(v); // row_mirror
    const long long bits = __double_as_longlong(v);
    const int lo = (int)(bits & 0xffffffffll), hi = (int)(bits >> 32);
    double r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
    {
        const int l = __builtin_amdgcn_readlane(lo, 16 * k);
        const int h = __builtin_amdgcn_readlane(hi, 16 * k);
        r[k] = __longlong_as_double(((long long)h << 32) | (unsigned int)l);
    }
    return (r[0] + r[1]) + (r[2] + r[3]);
}

// Block-wide sum of NV values held per thread, fixed order.  Every thread parks its NV values in LDS
// ([v][thread]); wave w then owns values v = w, w + NW, ...: each lane adds the T/64 entries of its
// column, one DPP wave_sum finishes the value.  (Doing NV wave_sums per wave instead costs ~270 cycles
// each on the critical path of every launch.)  lds must hold NV * T doubles; out[v] receives total v.
template <int NV, int T>
__device__ __forceinline__ void block_sum_to(const double *vals, double *lds, double *out, size_t out_stride = 1)
{
    constexpr int NW = T / 64;
    constexpr int VPW = (NV + NW - 1) / NW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int v = 0; v < NV; ++v)
        lds[v * T + threadIdx.x] = vals[v];
    __syncthreads();
    // the (at most VPW) values of this wave are independent chains: unrolled so that they interleave
    double a[VPW];
#pragma unroll
    for (int q = 0; q < VPW; ++q)
    {
        const int v = wave + q * NW;
        a[q] = 0.0;
        if (v < NV)
        {
#pragma unroll
            for (int w = 0; w < NW; ++w)
                a[q] += lds[v * T + w * 64 + lane];
        }
    }
#pragma unroll
    for (int q = 0; q < VPW; ++q)
        a[q] = wave_sum(a[q]);
#pragma unroll
    for (int q = 0; q < VPW; ++q)
    {
        const int v = wave + q * NW;
        if (lane == 0 && v < NV)
            out[(size_t)v * out_stride] = a[q];
    }
}

// legacy shape kept for the kernels that reduce a handful of values once per launch
template <int NV, int T>
__device__ __forceinline__ double block_sum_slots(const double *vals, double *lds)
{
    constexpr int NW = T / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int v = 0; v < NV; ++v)
    {
        const double s = wave_sum(vals[v]);
        if (lane == 0)
            lds[wave * NV + v] = s;
    }
    __syncthreads();
    double tot = 0.0;
    if (threadIdx.x < NV)
    {
#pragma unroll
        for (int w = 0; w < NW; ++w)
            tot += lds[w * NV + threadIdx.x];
    }
    return tot;
}

// An offset that is zero at run time but opaque to the compiler: forces the p-sized state through
// vector registers.  Without it the uniform loads go to SGPRs, and a p-sized fp64 algebra that
// lives in ~100 SGPRs is spilled lane by lane (v_writelane/v_readlane) -- measured 3x slower.
__device__ __forceinline__ int opaque_zero()
{
    int z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    return z;
}

// make a value opaque to the compiler while it sits in scalar registers (readfirstlane is a no-op for a value
// that is already scalar; it keeps the constraint legal where the compiler holds the kernel argument in VGPRs)
__device__ __forceinline__ void pin_sgpr(int &v)
{
    v = __builtin_amdgcn_readfirstlane(v);
    asm volatile("" : "+s"(v));
}
__device__ __forceinline__ void pin_sgpr(unsigned int &v)
{
    int t = (int)v;
    pin_sgpr(t);
    v = (unsigned int)t;
}
__device__ __forceinline__ void pin_sgpr(double &v)
{
    const long long bits = __double_as_longlong(v);
    int lo = (int)(bits & 0xffffffffll), hi = (int)(bits >> 32);
    pin_sgpr(lo);
    pin_sgpr(hi);
    v = __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <class Q>
__device__ __forceinline__ void pin_sgpr(Q *&v)
{
    const long long bits = (long long)reinterpret_cast<uintptr_t>(v);
    int lo = (int)(bits & 0xffffffffll), hi = (int)(bits >> 32);
    pin_sgpr(lo);
    pin_sgpr(hi);
    v = reinterpret_cast<Q *>((uintptr_t)(((long long)hi << 32) | (unsigned int)lo));
}
__device__ __forceinline__ void pin_params(LmParams &q)
{
    pin_sgpr(q.maxiter);
    pin_sgpr(q.trs);
    pin_sgpr(q.scale);
    pin_sgpr(q.fdtype);
    pin_sgpr(q.jac_analytic);
    pin_sgpr(q.fvv_analytic);
    pin_sgpr(q.has_bounds);
    pin_sgpr(q.has_weights);
    pin_sgpr(q.bench_hold);
    pin_sgpr(q.factor_up);
    pin_sgpr(q.factor_down);
    pin_sgpr(q.avmax);
    pin_sgpr(q.h_df);
    pin_sgpr(q.h_fvv);
    pin_sgpr(q.xtol);
    pin_sgpr(q.ftol);
    pin_sgpr(q.gtol);
    pin_sgpr(q.chisq_in);
}

// lm_advance out of line: its own register allocation, nothing of the caller's row loop live across it.
// Used where the caller is already short of registers -- the workgroup-per-dataset kernel (irls_batch.hpp) and the
// step kernel of the interpreted expression models.  The state LIVES IN LDS and is passed as 32-bit LDS byte offsets.
// Through generic pointers the out-of-line function cannot know which memory they name: for an LDS state the compiler
// proved "LDS or null" and guarded each of the ~700 accesses of the p = 8 instance with a 64-bit null test and two
// selects (718 v_cmp_ne_u64 + 828 v_cndmask in 12 k instructions); for a state in the caller's private memory every
// access was a scratch round trip (6 us per launch of the interpreted kernel).  Rebuilt from an LDS offset inside the
// function, every access is a plain ds_read / ds_write.
__device__ __forceinline__ unsigned lds_offset_of(const void *p)
{
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}
// (the control values too: a pointer to the caller's copy of them is a pointer into its private memory)
template <int P>
__device__ __attribute__((noinline)) void lm_advance_lds3(unsigned s_off, unsigned r_off, unsigned prm_off)
{
    typedef __attribute__((address_space(3))) LmState<P> *S3;
    typedef __attribute__((address_space(3))) const PassSums<P> *R3;
    typedef __attribute__((address_space(3))) const LmParams *Q3;
    LmState<P> *s = (LmState<P> *)(S3)(uintptr_t)s_off;
    const PassSums<P> *r = (const PassSums<P> *)(R3)(uintptr_t)r_off;
    const LmParams *prm = (const LmParams *)(Q3)(uintptr_t)prm_off;
    lm_advance<P>(*s, *r, *prm);
}

template <int P>
struct StepBcast
{
    int phase;
    double th[P];  // point the pass evaluates at (xt, or x for PH_FVV)
    double vel[P]; // velocity (PH_FVV only)
};

// models that keep per-thread data in the workgroup's dynamic LDS (ModelVM<P, true>)
template <class M>
struct vm_lds_twin
{
    static constexpr bool get()
    {
        if constexpr (M::ID == 100)
            return M::LDS_SLOTS;
        else
            return false;
    }
    static constexpr bool value = get();
};
extern __shared__ double gslnls_dyn_lds[];
template <bool VML>
__device__ __forceinline__ double *vm_lds_region(double *static_area)
{
    if constexpr (VML)
        return gslnls_dyn_lds;
    else
        return static_area;
}

// rows prefetched into registers before the prologue so their latency hides behind it
constexpr int ROWS_AHEAD = 8;
#ifndef GSLNLS_VM_ROWS_AHEAD
#define GSLNLS_VM_ROWS_AHEAD 2
#endif

// The first arguments are the pointers every wave needs before it can issue a single load; they are
// plain scalars so that the backend can preload them into SGPRs at wave launch
// (-mllvm -amdgpu-kernarg-preload-count=16): otherwise each launch starts with a ~2000-cycle wait on
// the kernarg segment.  The bulky rest (tolerances, trace pointers) is only needed after the loads.
template <class M, int JAC, int T>
__global__ __launch_bounds__(T) void lm_step_kernel(const LmState<M::P> *prev, const double *prev_partials,
                                                    const double *x0, const double *yv_, const double *swv_,
                                                    long long n, int G, int parity_and_flags, DenseCtx<M::P> ctx)
{
    constexpr int P = M::P;
    constexpr int NX = M::NX;
    using Sums = PassSums<P>;
    const int parity = parity_and_flags & 1;
    const unsigned int launch_idx = (unsigned int)parity_and_flags >> 2; // position of this launch in its fit
    const bool fresh_launch = (parity_and_flags & FRESH_LAUNCH) != 0; // first launch of a fit: no previous state
    constexpr int NV = Sums::NV;
    constexpr int NW = T / 64;
    // interpreted expression models: every unrolled row is one more inlined copy of the interpreter (8 copies are
    // ~300 KB of code per kernel)
    constexpr int R = (M::ID == 100) ? GSLNLS_VM_ROWS_AHEAD : ROWS_AHEAD;

    // interpreted model with its slot file in LDS (vm_model.hpp): the block reduction's staging area shares the
    // dynamic region with the slot files -- they are never live together (a barrier separates the row loop from the
    // reduction) -- so that the whole 160 KB minus a few hundred bytes is available: max(NV, slots) * T doubles
    constexpr bool VML = vm_lds_twin<M>::value;
    __shared__ double lds_red_static[VML ? 1 : NV * T];
    double *const lds_red = vm_lds_region<VML>(lds_red_static);
    __shared__ double lds_tot[NV];
    __shared__ StepBcast<P> lds_bc;
    __shared__ LmState<P> lds_vm_state[1]; // used by the interpreted models only (their state machine runs on it)
    __shared__ LmParams lds_vm_prm[1];

    const int tid = threadIdx.x;

    GSLNLS_STAMP(0);
    GSLNLS_ADV_STAMP(6);
    // ---------------- loads, in the order their data is needed ---------------------------------
    // Vector-memory results return in issue order and the L1 moves 64 B/clk, so what the critical path waits
    // for goes first, and it is issued by the very first instructions of the wave (the pointers are preloaded
    // kernel arguments): wave 0 owns the state (same address in every lane: one request each), waves 1.. load
    // the G partial sets of the previous launch.  Everything else -- LDS set-up, the barrier that publishes it,
    // the scalar loads of the control values -- happens while those loads are in flight (they take ~2700
    // cycles: the partial sets were written by other XCDs).  The reducing waves then publish the totals through
    // LDS + an arrival counter and only then issue their row prefetch (64 KB per CU, 16 MB for the grid), which
    // streams in under wave 0's lm_advance instead of ahead of the partial sums.
    const double *__restrict__ yv = yv_;
    const double *__restrict__ swv = swv_;
    const long long stride = (long long)G * T;
    const long long i0 = (long long)blockIdx.x * T + tid;
    LmState<P> s;
    constexpr int PB = MAX_G / 64;
    constexpr int RW = NW - 1;                 // waves that reduce partials
    constexpr int VPW = (NV + RW - 1) / RW;    // values per reducing wave
    // values are taken QC at a time (all of them at once for the usual p <= 6; the chunking only keeps the
    // register file bounded when one wave owns dozens of values, p >= 9)
    constexpr int QC = VPW < 8 ? VPW : 8;
    const int lane = tid & 63, wave = tid >> 6;
    const int z0 = opaque_zero();
    const double *pp = prev_partials + z0;
    double pv[QC][PB];
    auto load_partials = [&](int q0) {
#pragma unroll
        for (int q = 0; q < QC; ++q)
        {
            const int v = (wave - 1) + (q0 + q) * RW;
#pragma unroll
            for (int j = 0; j < PB; ++j)
            {
                const int b = lane + 64 * j;
                pv[q][j] = (v < NV && b < G) ? pp[(size_t)v * G + b] : 0.0;
            }
        }
    };
    // row prefetch into registers: x, y[, sqrt w] of this thread's first R rows
    double px[R][NX], py[R], pw[R];
    // rows b0 + k stride, k < R (8-byte requests: 16-byte pairs of consecutive rows were measured and are slower,
    // 6.6 vs 6.3 us per launch at n = 1e6 and 4.3 vs 5.7 TB/s at n = 6.4e7); rows beyond n are clamped to row n-1
    auto fetch_rows = [&](long long b0) {
#pragma unroll
        for (int k = 0; k < R; ++k)
        {
            const long long i = b0 + k * stride;
            const long long ic = i < n ? i : (n - 1);
#pragma unroll
            for (int c = 0; c < NX; ++c)
                px[k][c] = x0[(size_t)c * n + ic];
            py[k] = yv[ic];
            pw[k] = swv ? swv[ic] : 1.0;
        }
    };
    auto row_of = [&](long long b0, int k) { return b0 + k * stride; };
    double cnt[4];
    if (fresh_launch)
    {
        // nothing to wait for: the state comes from the kernel arguments after the barrier
    }
    else if (wave == 0)
    {
        // Wave 0 has nothing to do until the totals are there, so its own rows are requested right away (one
        // eighth of the workgroup's prefetch; the other waves hold theirs back until the totals are published,
        // because 64 KB per CU in front of the partial sums delays them).
        const double *src = reinterpret_cast<const double *>(prev) + z0;
        double *dst = reinterpret_cast<double *>(&s);
        constexpr int ND = (int)(offsetof(LmState<P>, bad_steps) / 8);
        static_assert(offsetof(LmState<P>, bad_steps) % 8 == 0 && sizeof(LmState<P>) == ND * 8 + 8 * sizeof(int),
                      "state = ND doubles followed by 8 counters");
#pragma unroll
        for (int k = 0; k < ND; ++k)
            dst[k] = src[k];
        // the eight counters behind them arrive as four 64-bit words and are unpacked in registers: copied into
        // the struct as halves of doubles they cannot be promoted out of scratch memory, and every access to them
        // becomes a scratch round trip behind an s_waitcnt vmcnt(0) in the middle of lm_advance
#pragma unroll
        for (int k = 0; k < 4; ++k)
            cnt[k] = src[ND + k];
        // the rows behind the state: vector-memory results return in issue order, and the state is what the critical
        // path waits for.  The scheduling barriers keep the compiler from moving a use of a state register (and the
        // wait that goes with it) in front of the row requests.  (6.24 against 6.32 us per launch, same box.)
        __builtin_amdgcn_sched_barrier(0);
        fetch_rows(i0);
        __builtin_amdgcn_sched_barrier(0);
    }
    else
        load_partials(0);

    // The control values are read here, once, into SGPRs that the compiler may not re-derive from the kernarg
    // segment: left alone it loads each field where it is first used, i.e. a dozen scalar loads (the first a
    // cold miss: every launch has a fresh kernarg block) in the middle of wave 0's lm_advance.
    // (Not for the interpreted expression models, M::ID == 100: their kernels already spill scalar registers
    // into vector lanes by the hundred, and 44 more that are live from here to the end made the compiler hand
    // lm_scale a wrong `scale` -- seen on Roszman1 with forward differences.  Those launches are bound by the
    // interpreter's slot file, not by this.)
    // (... and not for natively compiled expression models with four parameters: see OUTLINE_ADVANCE below.)
    constexpr bool OUTLINE_ADVANCE = (M::ID == 100) || (M::ID > 100 && P == 4);
    constexpr bool PIN = !OUTLINE_ADVANCE;
    LmParams prm = ctx.prm;
    LmState<P> *state_out = ctx.state[parity];
    double *partials_out = ctx.partials[parity];
    double *ssrtrace = ctx.ssrtrace, *partrace = ctx.partrace;
    LmState<P> *host_mirror = ctx.host_mirror;
    unsigned int *done_seq = ctx.done_seq;
    unsigned int seq = ctx.seq;
    if constexpr (PIN)
    {
        pin_params(prm);
        pin_sgpr(state_out);
        pin_sgpr(partials_out);
        pin_sgpr(ssrtrace);
        pin_sgpr(partrace);
        pin_sgpr(host_mirror);
        pin_sgpr(done_seq);
        pin_sgpr(seq);
    }

    if (wave != 0 && !fresh_launch)
    {
        // wave w sums the G partials of values v = w-1, w-1 + RW, ...: lane-strided partial sums, DPP butterfly
#pragma unroll 1
        for (int q0 = 0; q0 < VPW; q0 += QC)
        {
            if (q0 > 0)
                load_partials(q0);
            double a[QC];
#pragma unroll
            for (int q = 0; q < QC; ++q)
            {
                a[q] = pv[q][0];
#pragma unroll
                for (int j = 1; j < PB; ++j)
                    a[q] += pv[q][j];
            }
#pragma unroll
            for (int q = 0; q < QC; ++q)
                a[q] = wave_sum(a[q]);
            if (lane == 0)
            {
#pragma unroll
                for (int q = 0; q < QC; ++q)
                {
                    const int v = (wave - 1) + (q0 + q) * RW;
                    if (v < NV)
                        lds_tot[v] = a[q];
                }
            }
        }
    }
    // The barrier is the signal: wave 0 leaves it when every reducing wave has parked its totals in LDS.
    __syncthreads();
    GSLNLS_ADV_STAMP(7);
    GSLNLS_STAMP(7);
    if (wave != 0 || fresh_launch)
        fetch_rows(i0);

    GSLNLS_STAMP(1);
    if (fresh_launch)
    {
        if (wave == 0)
        {
            double lu[2 * P];
#pragma unroll
            for (int k = 0; k < P; ++k)
            {
                lu[2 * k] = ctx.sa.lo[k];
                lu[2 * k + 1] = ctx.sa.up[k];
            }
            lm_state_reset<P>(s, ctx.sa.start, lu);
            s.bad_steps = -1; // "fresh": this launch skips the advance
        }
    }
    else if (wave == 0)
    {
        auto lo32 = [](double d) { return (int)(__double_as_longlong(d) & 0xffffffffll); };
        auto hi32 = [](double d) { return (int)(__double_as_longlong(d) >> 32); };
        s.bad_steps = lo32(cnt[0]);
        s.niter = hi32(cnt[0]);
        s.phase = lo32(cnt[1]);
        s.status = hi32(cnt[1]);
        s.info = lo32(cnt[2]);
        s.nevalf = hi32(cnt[2]);
        s.nevaldf = lo32(cnt[3]);
        s.nevalfvv = hi32(cnt[3]);
    }
    int niter_before = 0, phase_before = PH_DONE;
    bool advanced = false;
    if (tid < 64)
    {
        const bool fresh = s.bad_steps < 0; // host marks a brand-new state with bad_steps = -1
        if (s.phase == PH_DONE)
        {
            // fit finished in an earlier launch: nothing to do but keep the final state under both parities
        }
        else if (fresh)
            s.bad_steps = 0;
        else
        {
            Sums r;
            double *rf = reinterpret_cast<double *>(&r);
#pragma unroll
            for (int v = 0; v < NV; ++v)
                rf[v] = lds_tot[v];
            niter_before = s.niter;
            phase_before = s.phase;
            advanced = true;
            GSLNLS_STAMP(2);
            if constexpr (OUTLINE_ADVANCE)
            {
                // Natively compiled formulas with p = 4 (round 4: scripts/dev_dbg_native_p.py -- of p = 2 .. 9 exactly p = 4
                // took other steps than the interpreter, the first one with a wrong damping parameter: the same
                // miscompilation as below, now by the in-process compiler; every other p is bit-identical to the
                // interpreter, and so is p = 4 with the state machine compiled on its own).
                // Interpreted expression model: the kernel around this call spills registers by the kilobyte, and with
                // lm_advance inlined into it the forward-difference instance for p = 4 took its first step with the
                // damping parameter of the reset state (mu = 0) instead of the one just computed -- seen on Roszman1,
                // gone with the state machine compiled on its own.
                // ... and on a copy of the state in LDS, addressed by LDS offsets: through pointers into this
                // wavefront's private memory every access of the state machine was a scratch round trip
                if (lane == 0)
                {
                    lds_vm_state[0] = s;
                    lds_vm_prm[0] = prm;
                }
                lm_advance_lds3<P>(lds_offset_of(&lds_vm_state[0]), lds_offset_of(lds_tot), lds_offset_of(&lds_vm_prm[0]));
                s = lds_vm_state[0];
            }
            else
                lm_advance<P>(s, r, prm);
            if (prm.bench_hold && s.phase == PH_DONE)
            {
                // timing mode: never finish, so that every launch pays the full prologue and a full pass
                s.phase = PH_TRIAL;
                s.status = ST_CONTINUE;
                s.mu = 1.0;
                s.nu = 2.0;
                s.bad_steps = 0;
            }
            GSLNLS_STAMP(3);
        }
        if (tid == 0)
        {
            lds_bc.phase = s.phase;
#pragma unroll
            for (int k = 0; k < P; ++k)
            {
                lds_bc.th[k] = (s.phase == PH_FVV) ? s.x[k] : s.xt[k];
                lds_bc.vel[k] = s.vel[k];
            }
        }
    }
    __syncthreads();
    GSLNLS_STAMP(4);

    // Workgroup 0 keeps the record: state for the next launch, trace rows, and -- when the fit has ended -- the
    // copy in pinned host memory.  All of it after the barrier (the other waves are already on their rows) and
    // as global stores: a flat store also counts as an LDS operation, and the wait in front of the barrier would
    // hold the whole workgroup until the 23 stores of the state had completed.
    if (blockIdx.x == 0 && tid == 0)
    {
        typedef __attribute__((address_space(1))) double GDouble;
        // the state as ND doubles + the eight counters packed in four 64-bit words (the mirror image of the load)
        constexpr int ND = (int)(offsetof(LmState<P>, bad_steps) / 8);
        double img[ND + 4];
        {
            const double *src = reinterpret_cast<const double *>(&s);
#pragma unroll
            for (int k = 0; k < ND; ++k)
                img[k] = src[k];
            auto pack = [](int lo, int hi) { return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo); };
            img[ND + 0] = pack(s.bad_steps, s.niter);
            img[ND + 1] = pack(s.phase, s.status);
            img[ND + 2] = pack(s.info, s.nevalf);
            img[ND + 3] = pack(s.nevaldf, s.nevalfvv);
        }
        auto store_state = [&](LmState<P> *where) {
            GDouble *dst = (GDouble *)reinterpret_cast<double *>(where);
#pragma unroll
            for (int k = 0; k < ND + 4; ++k)
                dst[k] = img[k];
        };
        if (advanced && ssrtrace)
        {
            GDouble *st = (GDouble *)ssrtrace;
            GDouble *pt = (GDouble *)partrace;
            // callback (src/nls.c:980-995): trace row 0 after init, row niter after each iteration
            if (phase_before == PH_INIT)
            {
                st[0] = s.chisq_init;
                for (int k = 0; k < P; ++k)
                    pt[(size_t)(prm.maxiter + 1) * k] = s.x[k];
            }
            else if (s.niter != niter_before && s.status != ST_EBADFUNC &&
                     !(s.status == ST_ENOPROG && niter_before == 0))
            {
                st[s.niter] = s.chisq1;
                for (int k = 0; k < P; ++k)
                    pt[s.niter + (size_t)(prm.maxiter + 1) * k] = s.x[k];
            }
        }
        store_state(state_out);
        if (s.phase == PH_DONE)
        {
            // final state to pinned host memory, then the fit's sequence number with system-scope
            // release: the host polls the sequence word and can return without draining the stream
            store_state(host_mirror);
            if (advanced)
                done_seq[1] = launch_idx; // the launch that ended the fit (the host sizes the next fit's first chunk by it)
            __threadfence_system();
            __hip_atomic_store(done_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }

    const int phase = lds_bc.phase;
    if (phase == PH_DONE)
        return;

    // ---------------- pass over this workgroup's rows ---------------------------------------
    double th[P], vel[P], delta[P];
#pragma unroll
    for (int k = 0; k < P; ++k)
    {
        th[k] = lds_bc.th[k];
        vel[k] = lds_bc.vel[k];
    }
    fd_deltas<P>(th, prm.h_df, delta);

    Sums acc;
    pass_zero<P>(acc);
    // Rows beyond n were prefetched from row n-1 (clamped index) and are switched off through their weight:
    // sw = 0 zeroes the Jacobian row, a select zeroes f.  No branch per row, so the accumulators stay plain
    // FMA chains (a predicated row costs a zero-fill + add per accumulator).  The phase is uniform: one
    // loop nest per kind of pass.
    if (phase == PH_FVV)
    {
        auto do_row = [&](const double *xr, double y, double sw, bool live) {
            double Jrow[P];
            double fv = row_fvv<M, JAC>(th, vel, delta, prm.h_fvv, prm.fvv_analytic != 0, xr, y, sw, Jrow,
                                        &acc.badj);
            fv = live ? fv : 0.0;
#pragma unroll
            for (int k = 0; k < P; ++k)
                acc.g[k] += Jrow[k] * fv;
        };
#pragma unroll
        for (int k = 0; k < R; ++k)
        {
            const bool live = row_of(i0, k) < n;
            do_row(px[k], py[k], live ? pw[k] : 0.0, live);
        }
        // rows beyond the prefetch window (n > R * G * T): again R rows at a time, loads first
        for (long long b0 = i0 + R * stride; b0 < n; b0 += R * stride)
        {
            fetch_rows(b0);
#pragma unroll
            for (int k = 0; k < R; ++k)
            {
                const bool live = row_of(b0, k) < n;
                do_row(px[k], py[k], live ? pw[k] : 0.0, live);
            }
        }
    }
    else
    {
        auto do_row = [&](const double *xr, double y, double sw, bool live) {
            double Jrow[P];
            double f = row_fj<M, JAC>(th, delta, xr, y, sw, Jrow, &acc.badj);
            f = live ? f : 0.0;
            acc_fj<P>(acc, f, Jrow);
        };
#pragma unroll
        for (int k = 0; k < R; ++k)
        {
            const bool live = row_of(i0, k) < n;
            do_row(px[k], py[k], live ? pw[k] : 0.0, live);
        }
        for (long long b0 = i0 + R * stride; b0 < n; b0 += R * stride)
        {
            fetch_rows(b0);
#pragma unroll
            for (int k = 0; k < R; ++k)
            {
                const bool live = row_of(b0, k) < n;
                do_row(px[k], py[k], live ? pw[k] : 0.0, live);
            }
        }
    }

    GSLNLS_STAMP(5);
    // ---------------- workgroup reduction -> one partial set ---------------------------------
    block_sum_to<NV, T>(reinterpret_cast<const double *>(&acc), lds_red, partials_out + blockIdx.x, (size_t)G);
    GSLNLS_STAMP(6);
}

// After the fit: weighted residual and Jacobian at the final point, in the layout C_nls
// returns them (resid n; grad n x p column-major, src/nls.c:695-737), plus (J^T J)^-1.
template <class M, int JAC, int T>
__global__ __launch_bounds__(T) void lm_finalize_kernel(DenseCtx<M::P> ctx, int parity, double *resid, double *grad,
                                                        double *covar)
{
    constexpr int P = M::P;
    constexpr int NX = M::NX;
    const LmState<P> *s = ctx.state[parity];
    double th[P], delta[P];
#pragma unroll
    for (int k = 0; k < P; ++k)
        th[k] = s->x[k];
    fd_deltas<P>(th, ctx.prm.h_df, delta);
    const long long n = ctx.n;
    const long long stride = (long long)gridDim.x * T;
    for (long long i = (long long)blockIdx.x * T + threadIdx.x; i < n; i += stride)
    {
        double xr[NX];
#pragma unroll
        for (int c = 0; c < NX; ++c)
            xr[c] = ctx.x[c][i];
        const double sw = ctx.sw ? ctx.sw[i] : 1.0;
        double Jrow[P], nb = 0.0;
        const double f = row_fj<M, JAC>(th, delta, xr, ctx.y[i], sw, Jrow, &nb);
        if (resid)
            resid[i] = f;
        if (grad)
        {
#pragma unroll
            for (int k = 0; k < P; ++k)
                grad[i + (size_t)n * k] = Jrow[k];
        }
    }
    if (covar && blockIdx.x == 0 && threadIdx.x == 0)
        covar_from_jtj<P>(s->A, covar);
}

} // namespace gslnls
