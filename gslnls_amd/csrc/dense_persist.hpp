// dense_persist.hpp -- the whole trust-region loop of one fit in ONE launch.
//
// lm_step_kernel (dense_kernels.hpp) uses the kernel boundary as its grid barrier: one launch per trial
// step.  lm_fit_kernel keeps the G workgroups resident for the whole fit (src/nls_fit.c:58-103 +
// src/trust.c:408-549 end to end):
//
//   * every thread loads its first RK rows of (x, y[, sqrt w]) ONCE into LDS and re-reads them from there in
//     every step (C2: 8 rows x 512 threads x 16 B = 64 KB per workgroup: the whole 16 MB data set is resident in
//     the chip's LDS; rows that do not fit the LDS budget are streamed per step exactly as in lm_step_kernel);
//   * per step: rows -> workgroup sums (LDS) -> an all-reduce in three hops of tagged 8-byte granules:
//       hop 1: workgroup b publishes its NV sums; the leader of group g = b mod NG (workgroup g, NG <= 8 groups)
//              gathers its <= 32 members and adds them in a fixed lane tree (waves 1.., two values each),
//       hop 2: the same leader waves publish their group totals, gather those of the other leaders and add the NG
//              of them in a fixed order,
//       hop 3: they publish the totals to their own group; wave 0 of EVERY workgroup picks them up and runs the same
//              lm_advance() redundantly -- no broadcast of the new trial point;
//   * a granule is {tag, half of a double} in one naturally aligned 8-byte word written by one store and polled
//     with agent-scope (sc1) loads: the data is its own flag, no fence (cdna_hip_programming.md Guideline 16,
//     form R2).  tag = (fit sequence number, step) never repeats within the two-deep slot ring, so nothing has to
//     be zeroed between fits;
//   * placement-independent by default: groups are defined by blockIdx, every store agent-scope (write-through).
//     `fast` mode (PersistArgs::fast) stores hops 1 and 3 -- which stay inside a group -- as plain stores that stop
//     in the XCD's L2: valid only when all workgroups of a group share an XCD (see dense_host.hpp, which only sets
//     it after the device has reported that placement);
//   * every spin is bounded: a workgroup that waits too long marks the fit as timed out and all of them leave;
//     the host then runs the fit through the launch-per-step kernel (another HIP path of this library -- not a
//     CPU fallback).  Requires all G workgroups co-resident: G <= number of CUs, one workgroup per CU.
//
// Fixed summation shape => run-to-run bit-identical results (not bit-identical to lm_step_kernel, whose
// partial sums are added in a different tree; both agree with the oracle to the documented tolerance).
#pragma once
#include "dense_kernels.hpp"

namespace gslnls
{

typedef unsigned long long u64_t;
typedef __attribute__((address_space(1))) u64_t gu64_t;

constexpr int PG_MAX = 8;    // groups (leaders)
constexpr int PM_MAX = 32;   // members per group: PG_MAX * PM_MAX = MAX_G workgroups
constexpr int PH_ABORT = 7;  // broadcast value: a bounded spin gave up
constexpr unsigned PERSIST_SPIN_LIMIT = 1u << 18;

// granule pair of one double: slot[0] = tag:lo32, slot[1] = tag:hi32
__device__ __forceinline__ void gran_store(u64_t *slot, unsigned tag, double v)
{
    const u64_t b = (u64_t)__double_as_longlong(v);
    gu64_t *g = (gu64_t *)slot;
    __hip_atomic_store(g, ((u64_t)tag << 32) | (b & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(g + 1, ((u64_t)tag << 32) | (b >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool gran_load(const u64_t *slot, unsigned tag, double &v)
{
    gu64_t *g = (gu64_t *)slot;
    const u64_t a = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const u64_t b = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v = __longlong_as_double((long long)((b << 32) | (a & 0xffffffffull)));
    return (unsigned)(a >> 32) == tag && (unsigned)(b >> 32) == tag;
}

template <int P>
struct PersistBufs
{
    static constexpr int NV = PassSums<P>::NV;
    // part[parity][group][value][member][2] | xtot[parity][value][group][2] | bc[parity][group][value][2]
    static constexpr size_t PART_WORDS = (size_t)2 * PG_MAX * NV * PM_MAX * 2;
    static constexpr size_t XTOT_WORDS = (size_t)2 * NV * PG_MAX * 2;
    static constexpr size_t BC_WORDS = (size_t)2 * PG_MAX * (NV + 1) * 2; // + the group's `fast` verdict
    static constexpr size_t XCC_WORDS = MAX_G; // diagnostic: the XCC id every workgroup ran on (last launch)
    static constexpr size_t WORDS = PART_WORDS + XTOT_WORDS + BC_WORDS + XCC_WORDS;
    __host__ __device__ static size_t part_at(int parity, int g, int v, int m)
    {
        return ((((size_t)parity * PG_MAX + g) * NV + v) * PM_MAX + m) * 2;
    }
    __host__ __device__ static size_t xtot_at(int parity, int v, int g)
    {
        return PART_WORDS + (((size_t)parity * NV + v) * PG_MAX + g) * 2;
    }
    __host__ __device__ static size_t bc_at(int parity, int g, int v)
    {
        return PART_WORDS + XTOT_WORDS + (((size_t)parity * PG_MAX + g) * (NV + 1) + v) * 2;
    }
    __host__ __device__ static size_t xcc_at(int b) { return PART_WORDS + XTOT_WORDS + BC_WORDS + (size_t)b; }
};

// sum over the 8 lanes of an aligned group, result in all 8 (fixed tree)
__device__ __forceinline__ double oct_sum(double v)
{
    v += dpp_mov<0xB1>(v);  // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);  // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v); // row_half_mirror: lane i <-> 7 - i inside each 8
    return v;
}
__device__ __forceinline__ double lane_bcast(double v, int lane)
{
    const long long bits = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(bits & 0xffffffffll), lane);
    const int hi = __builtin_amdgcn_readlane((int)(bits >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// Diagnostic build only (-DGSLNLS_STAMPS): 100 MHz s_memrealtime stamps (one clock for the whole chip) of step
// PERSIST_STAMP_STEP go to a debug buffer nothing else reads: [workgroup][wave 0 / wave 1][8 slots]
#ifdef GSLNLS_STAMPS
constexpr int PERSIST_STAMP_STEP = 20;
#define GSLNLS_PSTAMP(role, slot)                                                                     \
    do                                                                                                \
    {                                                                                                 \
        if (ctx.stamps && step == PERSIST_STAMP_STEP && lane == 0)                                    \
            ctx.stamps[((size_t)blockIdx.x * 2 + (role)) * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define GSLNLS_PSTAMP(role, slot) \
    do                            \
    {                             \
    } while (0)
#endif

struct PersistArgs
{
    u64_t *bufs;          // PersistBufs<P>::WORDS words of device memory
    unsigned tag_base;    // ((fit sequence number & 0x7ffff) | 0x80000) << 12: never zero
    int step0;            // steps this fit has already taken (resume)
    int max_steps;        // leave after this many steps of this launch (the host looks in and resumes)
    int resume;           // 0: fresh fit from ctx.sa; 1: continue from ctx.state[0]
    int rows_resident;    // rows every thread keeps in LDS for the whole fit (dynamic LDS of the launch is sized for it)
    int fast;             // 0: every store write-through (placement-independent); 1: groups whose workgroups all report the
                          // leader's XCD switch hops 1 and 3 to stores that stop in that XCD's L2
};

// granule store that stays in the issuing XCD's L2 (a plain 8-byte store; the compiler may not split or merge it)
__device__ __forceinline__ void gran_store_l2(u64_t *slot, unsigned tag, double v)
{
    const u64_t b = (u64_t)__double_as_longlong(v);
    gu64_t *g = (gu64_t *)slot;
    __hip_atomic_store(g, ((u64_t)tag << 32) | (b & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_store(g + 1, ((u64_t)tag << 32) | (b >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <class M, int JAC, int T>
__global__ __launch_bounds__(T) void lm_fit_kernel(const double *x0, const double *yv_, const double *swv_,
                                                   long long n, int G, PersistArgs pa, DenseCtx<M::P> ctx)
{
    constexpr int P = M::P;
    constexpr int NX = M::NX;
    using Sums = PassSums<P>;
    using PB = PersistBufs<P>;
    constexpr int NV = Sums::NV;
    constexpr int NW = T / 64;

    extern __shared__ double lds_rows[]; // resident rows, sized by the host (PersistArgs::rows_resident)
    __shared__ double lds_red[NV * T];
    __shared__ double lds_tot[NV];
    __shared__ StepBcast<P> lds_bc;
    __shared__ int lds_fast;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    const int NG = G < PG_MAX ? G : PG_MAX;
    const int grp = b % NG, mem = b / NG;
    const int nmem = (G - grp + NG - 1) / NG; // members of my group (used by its leader: b == grp)
    const bool leader = b < NG;
    const double *__restrict__ yv = yv_;
    const double *__restrict__ swv = swv_;
    const long long stride = (long long)G * T;
    const long long i0 = (long long)b * T + tid;
    typedef __attribute__((address_space(1))) double GDouble;

    // Which XCD this workgroup runs on (HW_REG_XCC_ID = hwreg 20, bits 3:0), published once, write-through, and
    // drained before anything else is stored: a leader that sees a member's first sums also sees its XCC id.
    const unsigned my_xcc = (unsigned)__builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11)) & 0xfu;
    const unsigned xcc_tag = pa.tag_base | (unsigned)((pa.step0 + 1) & 0xfff); // tag of this launch's first step
    if (tid == 0)
    {
        __hip_atomic_store((gu64_t *)(pa.bufs + PB::xcc_at(b)), ((u64_t)xcc_tag << 32) | (u64_t)my_xcc, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // fast: hops 1 and 3 of my group may stop in the XCD's L2.  Decided by the group's leader in the first step of
    // the launch (all members on its own XCD), handed to the members with that step's totals; never assumed.
    bool fast = false;

    // ---- this thread's resident rows: loaded once into LDS slots only this thread touches, SoA
    // [(k * NC + c) * T + tid] (c: regressors, y, sqrt w) so that every ds_read_b64 of a wave is contiguous ----
    const int has_w = swv ? 1 : 0;
    const int NC = NX + 1 + has_w;
    const int RK = pa.rows_resident;
    for (int k = 0; k < RK; ++k)
    {
        const long long i = i0 + k * stride;
        const long long ic = i < n ? i : (n - 1);
#pragma unroll
        for (int c = 0; c < NX; ++c)
            lds_rows[(k * NC + c) * T + tid] = x0[(size_t)c * n + ic];
        lds_rows[(k * NC + NX) * T + tid] = yv[ic];
        if (swv)
            lds_rows[(k * NC + NX + 1) * T + tid] = swv[ic];
    }
    // rows beyond n were filled from row n-1 and are switched off: weight 0 zeroes the Jacobian row, a select zeroes f
    auto resident_row = [&](int k, double *xr, double &y, double &sw, bool &live) {
        live = (i0 + k * stride) < n;
#pragma unroll
        for (int c = 0; c < NX; ++c)
            xr[c] = lds_rows[(k * NC + c) * T + tid];
        y = lds_rows[(k * NC + NX) * T + tid];
        // (without weights the slot read is y again: one more ds_read instead of a branch per row)
        const double w = lds_rows[(k * NC + NX + has_w) * T + tid];
        sw = live ? (has_w ? w : 1.0) : 0.0;
    };

    LmParams prm = ctx.prm;
    double *ssrtrace = ctx.ssrtrace, *partrace = ctx.partrace;
    LmState<P> *host_mirror = ctx.host_mirror;
    unsigned int *done_seq = ctx.done_seq;
    unsigned int seq = ctx.seq;
    if constexpr (M::ID != 100)
    {
        pin_params(prm);
        pin_sgpr(ssrtrace);
        pin_sgpr(partrace);
        pin_sgpr(host_mirror);
        pin_sgpr(done_seq);
        pin_sgpr(seq);
    }

    LmState<P> s; // wave 0 only: in registers for the whole fit
    auto publish = [&]() {
        if (lane == 0)
        {
            lds_bc.phase = s.phase;
#pragma unroll
            for (int k = 0; k < P; ++k)
            {
                lds_bc.th[k] = (s.phase == PH_FVV) ? s.x[k] : s.xt[k];
                lds_bc.vel[k] = s.vel[k];
            }
        }
    };
    if (tid == 0)
        lds_fast = 0;
    if (wave == 0)
    {
        if (pa.resume)
        {
            const double *src = reinterpret_cast<const double *>(ctx.state[0]) + opaque_zero();
            double *dst = reinterpret_cast<double *>(&s);
            constexpr int ND = (int)(offsetof(LmState<P>, bad_steps) / 8);
#pragma unroll
            for (int k = 0; k < ND; ++k)
                dst[k] = src[k];
            double cnt[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                cnt[k] = src[ND + k];
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
        else
        {
            double lu[2 * P];
#pragma unroll
            for (int k = 0; k < P; ++k)
            {
                lu[2 * k] = ctx.sa.lo[k];
                lu[2 * k + 1] = ctx.sa.up[k];
            }
            lm_state_reset<P>(s, ctx.sa.start, lu);
        }
        publish();
    }
    auto store_state = [&](LmState<P> *where) {
        constexpr int ND = (int)(offsetof(LmState<P>, bad_steps) / 8);
        GDouble *dst = (GDouble *)reinterpret_cast<double *>(where);
        const double *src = reinterpret_cast<const double *>(&s);
#pragma unroll
        for (int k = 0; k < ND; ++k)
            dst[k] = src[k];
        auto pack = [](int lo, int hi) { return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo); };
        dst[ND + 0] = pack(s.bad_steps, s.niter);
        dst[ND + 1] = pack(s.phase, s.status);
        dst[ND + 2] = pack(s.info, s.nevalf);
        dst[ND + 3] = pack(s.nevaldf, s.nevalfvv);
    };

    bool timed_out = false;
    for (int step = 0;; ++step)
    {
        if (tid == 0 && timed_out)
            lds_bc.phase = PH_ABORT;
        __syncthreads(); // B1: the point of this step is in lds_bc
        if (wave <= 1)
            GSLNLS_PSTAMP(wave, 0);
        if (step == 1 && pa.fast)
            fast = lds_fast != 0; // the group's verdict, picked up by wave 0 with the first totals
        const int phase = lds_bc.phase;
        if (phase == PH_DONE || phase == PH_ABORT || step >= pa.max_steps)
        {
            if (b == 0 && tid == 0)
            {
                store_state(ctx.state[0]); // the resume point (and what lm_finalize_kernel reads)
                if (phase == PH_ABORT)
                    __hip_atomic_store(done_seq + 2, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (phase == PH_DONE)
                {
                    store_state(host_mirror);
                    done_seq[1] = (unsigned int)(pa.step0 + step);
                    __threadfence_system();
                    __hip_atomic_store(done_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
            return;
        }

        // ---- rows ----------------------------------------------------------------------------------
        Sums acc;
        pass_zero<P>(acc);
        {
            double th[P], vel[P], delta[P];
#pragma unroll
            for (int k = 0; k < P; ++k)
            {
                th[k] = lds_bc.th[k];
                vel[k] = lds_bc.vel[k];
            }
            fd_deltas<P>(th, prm.h_df, delta);
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
                for (int k = 0; k < RK; ++k)
                {
                    double xr[NX], y, sw;
                    bool live;
                    resident_row(k, xr, y, sw, live);
                    do_row(xr, y, sw, live);
                }
                for (long long i = i0 + (long long)RK * stride; i < n; i += stride)
                {
                    double xr[NX];
#pragma unroll
                    for (int c = 0; c < NX; ++c)
                        xr[c] = x0[(size_t)c * n + i];
                    do_row(xr, yv[i], swv ? swv[i] : 1.0, true);
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
                for (int k = 0; k < RK; ++k)
                {
                    double xr[NX], y, sw;
                    bool live;
                    resident_row(k, xr, y, sw, live);
                    do_row(xr, y, sw, live);
                }
                // rows beyond the resident window: streamed, 8 at a time, loads first
                constexpr int RS = 8;
                for (long long b0 = i0 + (long long)RK * stride; b0 < n; b0 += RS * stride)
                {
                    double qx[RS][NX], qy[RS], qw[RS];
#pragma unroll
                    for (int k = 0; k < RS; ++k)
                    {
                        const long long i = b0 + k * stride;
                        const long long ic = i < n ? i : (n - 1);
#pragma unroll
                        for (int c = 0; c < NX; ++c)
                            qx[k][c] = x0[(size_t)c * n + ic];
                        qy[k] = yv[ic];
                        qw[k] = (i < n) ? (swv ? swv[ic] : 1.0) : 0.0;
                    }
#pragma unroll
                    for (int k = 0; k < RS; ++k)
                        do_row(qx[k], qy[k], qw[k], (b0 + k * stride) < n);
                }
            }
        }
        if (wave <= 1)
            GSLNLS_PSTAMP(wave, 1);

        // ---- workgroup sums: every wave reduces its share of the NV values (fixed order) ---------------
        constexpr int VPW = (NV + NW - 1) / NW;
#pragma unroll
        for (int v = 0; v < NV; ++v)
            lds_red[v * T + tid] = reinterpret_cast<const double *>(&acc)[v];
        __syncthreads(); // B2
        if (wave <= 1)
            GSLNLS_PSTAMP(wave, 7);
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
                    a[q] += lds_red[v * T + w * 64 + lane];
            }
        }
#pragma unroll
        for (int q = 0; q < VPW; ++q)
            a[q] = wave_sum(a[q]);
        const unsigned tag = pa.tag_base | (unsigned)((pa.step0 + step + 1) & 0xfff);
        const int parity = (pa.step0 + step) & 1;
        Sums r;
        double *rf = reinterpret_cast<double *>(&r);
        if (G == 1)
        {
#pragma unroll
            for (int q = 0; q < VPW; ++q)
            {
                const int v = wave + q * NW;
                if (lane == 0 && v < NV)
                    lds_tot[v] = a[q];
            }
            __syncthreads(); // B3: one workgroup -- its sums are the totals
            if (wave == 0)
            {
#pragma unroll
                for (int v = 0; v < NV; ++v)
                    rf[v] = lds_tot[v];
            }
        }
        else
        {
            // ---- hop 1: publish this workgroup's sums ----
#pragma unroll
            for (int q = 0; q < VPW; ++q)
            {
                const int v = wave + q * NW;
                if (lane == 0 && v < NV)
                {
                    u64_t *slot = pa.bufs + PB::part_at(parity, grp, v, mem);
                    if (fast)
                        gran_store_l2(slot, tag, a[q]);
                    else
                        gran_store(slot, tag, a[q]);
                }
            }
            if (wave <= 1)
                GSLNLS_PSTAMP(wave, 2);
            // ---- leaders: waves 1.. own two values each through hops 1 -> 2 -> 3 ----
            if (leader && wave >= 1 && !timed_out)
            {
                constexpr int NPAIR = (NV + 1) / 2;
                bool leader_fast = fast;
                for (int q = wave - 1; q < NPAIR; q += NW - 1)
                {
                    const int half = lane >> 5, sub = lane & 31;
                    const int v = 2 * q + half;
                    // hop 1: lane (half, m) gathers member m's sum of value v
                    double val = 0.0;
                    {
                        const bool act = (v < NV) && (sub < nmem);
                        const u64_t *slot = pa.bufs + PB::part_at(parity, grp, act ? v : 0, act ? sub : 0);
                        unsigned spins = 0;
                        for (;;)
                        {
                            double got = 0.0;
                            const bool ok = !act || gran_load(slot, tag, got);
                            if (__all(ok))
                            {
                                val = act ? got : 0.0;
                                break;
                            }
                            if (++spins > PERSIST_SPIN_LIMIT)
                            {
                                timed_out = true;
                                break;
                            }
                            __builtin_amdgcn_s_sleep(1);
                        }
                    }
                    if (timed_out)
                        break; // nothing is published: every workgroup's wave 0 runs into its own limit
                    if (wave == 1)
                        GSLNLS_PSTAMP(1, 3);
                    if (step == 0 && pa.fast)
                    {
                        // my members' sums are here, so their XCC ids are too (stored and drained first)
                        bool same = true;
                        if (sub < nmem)
                        {
                            const u64_t w = __hip_atomic_load((gu64_t *)(pa.bufs + PB::xcc_at(grp + NG * sub)), __ATOMIC_RELAXED,
                                                              __HIP_MEMORY_SCOPE_AGENT);
                            same = ((unsigned)(w >> 32) == xcc_tag) && ((unsigned)(w & 0xfu) == my_xcc);
                        }
                        leader_fast = __all(same);
                    }
                    // 32-lane sums in a fixed tree: butterflies inside each row of 16, then row 0 + row 1
                    val += dpp_mov<0xB1>(val);
                    val += dpp_mov<0x4E>(val);
                    val += dpp_mov<0x141>(val);
                    val += dpp_mov<0x140>(val);
                    const double ta = lane_bcast(val, 0) + lane_bcast(val, 16);
                    const double tb = lane_bcast(val, 32) + lane_bcast(val, 48);
                    // hop 2: group totals to the other leaders (always write-through: they sit on other XCDs)
                    if (lane == 0)
                        gran_store(pa.bufs + PB::xtot_at(parity, 2 * q, grp), tag, ta);
                    if (lane == 32 && 2 * q + 1 < NV)
                        gran_store(pa.bufs + PB::xtot_at(parity, 2 * q + 1, grp), tag, tb);
                    if (wave == 1)
                        GSLNLS_PSTAMP(1, 4);
                    // lane (half, g < NG) gathers group g's total of value v; my own comes from registers
                    double gv = 0.0;
                    {
                        const bool act = (v < NV) && (sub < NG) && (sub != grp);
                        const u64_t *slot = pa.bufs + PB::xtot_at(parity, act ? v : 0, act ? sub : 0);
                        unsigned spins = 0;
                        for (;;)
                        {
                            double got = 0.0;
                            const bool ok = !act || gran_load(slot, tag, got);
                            if (__all(ok))
                            {
                                gv = act ? got : 0.0;
                                break;
                            }
                            if (++spins > PERSIST_SPIN_LIMIT)
                            {
                                timed_out = true;
                                break;
                            }
                            __builtin_amdgcn_s_sleep(1);
                        }
                        if (sub == grp && v < NV)
                            gv = half ? tb : ta;
                    }
                    if (timed_out)
                        break;
                    if (wave == 1)
                        GSLNLS_PSTAMP(1, 5);
                    gv = oct_sum(gv); // lanes 0..7 and 32..39 hold the two totals
                    // hop 3: the totals to my own group
                    if (sub == 0 && v < NV)
                    {
                        u64_t *slot = pa.bufs + PB::bc_at(parity, grp, v);
                        if (fast)
                            gran_store_l2(slot, tag, gv);
                        else
                            gran_store(slot, tag, gv);
                    }
                    // the first wave-pass also hands the group its verdict on `fast` (first step only)
                    if (step == 0 && q == wave - 1 && wave == 1 && lane == 0)
                        gran_store(pa.bufs + PB::bc_at(parity, grp, NV), tag, leader_fast ? 1.0 : 0.0);
                    if (step == 0)
                        fast = leader_fast; // this wave's own later steps
                    if (wave == 1)
                        GSLNLS_PSTAMP(1, 6);
                }
            }
            // ---- wave 0 of every workgroup picks up the totals of its group's leader ----
            if (wave == 0)
            {
                constexpr int NL = (NV + 1 + 63) / 64;
                const int nv_here = NV + ((step == 0 && pa.fast) ? 1 : 0); // first step: + the verdict on `fast`
                double tot[NL];
                unsigned spins = 0;
                for (;;)
                {
                    bool ok = true;
#pragma unroll
                    for (int k = 0; k < NL; ++k)
                    {
                        const int v = k * 64 + lane;
                        double got = 0.0;
                        if (v < nv_here)
                            ok = gran_load(pa.bufs + PB::bc_at(parity, grp, v), tag, got) && ok;
                        tot[k] = got;
                    }
                    if (__all(ok))
                        break;
                    if (++spins > PERSIST_SPIN_LIMIT)
                    {
                        timed_out = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
#pragma unroll
                for (int v = 0; v < NV; ++v)
                    rf[v] = lane_bcast(tot[v >> 6], v & 63);
                if (step == 0 && pa.fast && !timed_out)
                    lds_fast = (lane_bcast(tot[NV >> 6], NV & 63) == 1.0) ? 1 : 0;
            }
        }
        if (wave == 0)
            GSLNLS_PSTAMP(0, 3);

        // ---- the state machine, redundantly in wave 0 of every workgroup ---------------------------
        if (wave == 0 && !timed_out)
        {
            const int niter_before = s.niter, phase_before = s.phase;
            lm_advance<P>(s, r, prm);
            if (prm.bench_hold && s.phase == PH_DONE)
            {
                s.phase = PH_TRIAL;
                s.status = ST_CONTINUE;
                s.mu = 1.0;
                s.nu = 2.0;
                s.bad_steps = 0;
            }
            publish();
            GSLNLS_PSTAMP(0, 4);
            if (b == 0 && tid == 0 && ssrtrace)
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
        }
    }
}

} // namespace gslnls
