// wide_kernels.hpp -- the pass over the n residual rows for 10 <= p <= 64 parameters: J^T J on the matrix cores.
//
// north_star: "MFMA tall-skinny GEMM for J^T J only when p fills a 16-wide tile".  The reference materialises the
// n x p Jacobian (src/nls.c:266, :885-912) and GSL's Cholesky solver forms J^T J with dsyrk (lower); here J is never
// stored: a wavefront evaluates 64 rows (lane = row: residual f_i and the p gradient entries, analytic from the
// compiled formula or by the reference's forward / central differences, src/fdjac.c), parks them in an LDS tile and
// immediately contracts the tile with v_mfma_f64_16x16x4_f64:
//
//   * tile layout: column-major J^T: tile[k][row], leading dimension 68 doubles -- the row phase writes 64
//     consecutive doubles per gradient entry (conflict-free), the MFMA phase reads, for the 4-row chunk c, lane l
//     (kk = l / 16, i = l % 16) the entry J[4c + kk][16 b + i] of every 16-column block b: i * 68 + kk walks the
//     banks in steps of 4 doubles + {0..3}: two lanes per 8-byte bank, the minimum for 64 lanes;
//   * operand layout of the instruction (probed on gfx950, scripts/mfma_probe): lane l supplies A'[l % 16][l / 16]
//     and B[l / 16][l % 16] and holds D[4 r + l / 16][l % 16] in result register r.  The register loaded for block b
//     is at once the A' operand of block row b and the B operand of block column b, so NB = PW / 16 loads feed the
//     NB (NB + 1) / 2 lower-triangle blocks of one chunk;
//   * J^T f rides along on the vector pipe: the same registers times f[4c + kk], one FMA per block and chunk, the four
//     row groups of a lane column added at the end; ssr and the non-finite flag are per-lane sums of the row phase;
//   * PW = 16 ceil(p / 16): the padding columns of the tile are zeroed once and never written.
//
// Compiled in process for one formula (rtc_host.hpp): M = the generated row model, M::P = the actual p.
#pragma once
#if !defined(__HIPCC_RTC__)
#include <hip/hip_runtime.h>
#endif
#include "lm_core.hpp"
#include "devmath.hpp"
#include "rowops.hpp"
#include "wide_core.hpp"

namespace gslnls
{

constexpr int WIDE_LD = 68;     // leading dimension of the tile (doubles)
constexpr int WIDE_T = 256;     // threads per workgroup of the pass
constexpr int WIDE_MAX_G = 512; // workgroups == partial sets (two per CU where registers and LDS allow)

// what one launch of the pass does beyond the per-workgroup partial sets
constexpr int WIDE_FUSE_NONE = 0;   // partial sets only (wide_reduce_kernel / wide_advance_kernel follow as launches)
constexpr int WIDE_FUSE_REDUCE = 1; // + the sets are reduced in the launch: totals[NV] in HBM when it ends
constexpr int WIDE_FUSE_STEP = 2;   // + the workgroup that completes the totals runs the LM step: ONE launch per trial step
constexpr int WIDE_NGRP = 16;       // level-1 groups of the in-launch reduction: group q = sets q, q + 16, q + 32, ...

// hand-off words of the in-launch reduction and of the speculative solve (one allocation, zeroed when it is made)
struct WFuseBuf
{
    unsigned int tickets[WIDE_NGRP]; // level 1: arrivals of group q's workgroups
    unsigned int ticket2;            // level 2: arrivals of the group reducers
    unsigned int pad[15];
    unsigned long long spec_tag;     // {launch epoch, fold of the payload}: the speculative solve of this launch is complete
    double spec_mu;                  // the damping it was computed for
    double spec_vel[64];             // (WP; wide_core.hpp)
};

struct WPassArgs
{
    const double *x;  // n x NX column-major
    const double *y;
    const double *sw; // sqrt(weights) or nullptr
    long long n;
    const WState *state;
    double *partials; // [G][NV]: ssr, badj, packed lower J^T J, J^T f -- one contiguous set per workgroup
    double h_df, h_fvv;
    int fvv_analytic;
    int wf_only; // gsl_nls_large: the weights scale f only -- the reference's callback never weights J (src/nls_large.c:629-633)
    // ---- one launch per trial step ----
    int fuse;       // WIDE_FUSE_*
    int G;          // workgroups that own rows (the grid has spec more: workgroup 0 is then the speculator)
    int spec;       // 1: workgroup 0 solves the damped system of the step that FOLLOWS A REJECTION of this launch's trial
    int spec_fault; // test switch (GSLNLS_WIDE_SPEC_FAULT): 1 the speculator publishes a payload that does not match its tag
                    // (a half-written payload), 2 a tag of another launch (a stale epoch), 3 nothing at all -- in every
                    // case the stepping workgroup must find "not there" and solve itself: the same fit bit for bit
    double *gsums;  // [WIDE_NGRP][NV] level-1 sums
    double *totals; // [NV]
    WFuseBuf *fb;
    unsigned long long *stamps; // developer diagnostic (GSLNLS_WIDE_STAMPS=1): [launch][8] 100 MHz stamps of the stepping workgroup
    WAdvanceArgs adv; // WIDE_FUSE_STEP
};

typedef double wide_v4f64 __attribute__((ext_vector_type(4)));

// theta as the generated model reads it: th[k] with a compile-time k, from the workgroup's LDS copy
struct WideTheta
{
    const double *p;
    __device__ __forceinline__ double operator[](int k) const { return p[k]; }
};
// theta + d e_j (finite differences): j is a run-time value, k a literal
struct WideThetaPert
{
    const double *p;
    int j;
    double d;
    __device__ __forceinline__ double operator[](int k) const { return k == j ? p[k] + d : p[k]; }
};
// theta + h v (second directional derivative by differences, src/fdfvv.c:35-77)
struct WideThetaDir
{
    const double *p, *v;
    double h;
    __device__ __forceinline__ double operator[](int k) const { return p[k] + h * v[k]; }
};
// what depends on the parameters alone, computed once per wavefront by M::prologue (rtc_host.hpp, round 5) and read back
// per row as broadcasts.  Every lane (and every wavefront of the workgroup) stores the same value to the same word.
struct WidePre
{
    const double *p;
    __device__ __forceinline__ double operator[](int k) const { return p[k]; }
};
struct WidePreOut
{
    double *p;
    __device__ __forceinline__ void set(int k, double v) { p[k] = v; }
};
// gradient entry k of this lane's row -> tile[k][lane], weighted; the non-finite flag as in row_fj
struct WideTileSink
{
    double *col; // &tile[0][lane]
    double sw;
    double bad;
    __device__ __forceinline__ void set(int k, double v)
    {
        bad = fma(v, 0.0, bad);
        col[k * WIDE_LD] = v * sw;
    }
};
struct WideGradSink
{
    double *dst; // &grad[i]
    long long n;
    double sw;
    __device__ __forceinline__ void set(int k, double v) { dst[(size_t)n * k] = v * sw; }
};

template <class M, class TH, class XR>
__device__ __forceinline__ double wide_resid(const TH &th, const XR &xr, double y, double sw)
{
    const double m = M::value(th, xr);
    const double f = isfinite(m) ? m - y : INFINITY;
    return f * sw;
}

// developer switches of the contraction loop (GSLNLS_RTC_EXTRA_FLAGS="-DGSLNLS_WIDE_CHUNK_AHEAD=3 ..."): how many 4-row
// chunks ahead of the matrix instructions the LDS reads run, and the wavefront's issue priority while it feeds the matrix pipe
#ifndef GSLNLS_WIDE_CHUNK_AHEAD
#define GSLNLS_WIDE_CHUNK_AHEAD 2
#endif
#ifndef GSLNLS_WIDE_MFMA_PRIO
#define GSLNLS_WIDE_MFMA_PRIO 0
#endif
struct wide_true
{
    static constexpr bool value = true;
};
struct wide_false
{
    static constexpr bool value = false;
};
#ifndef GSLNLS_WIDE_WAVES
#define GSLNLS_WIDE_WAVES(PW) ((PW) <= 32 ? 2 : 1) // workgroups per CU the register budget is cut for (LDS allows 2 up to PW = 32)
#endif
// ---- hand-off helpers of the in-launch reduction (cdna_hip_programming.md guideline 16) ----------------------------------
// payload stores are write-through (sc1: relaxed agent-scope atomic stores), so the producer needs no release fence: every
// storing wave drains its stores, the workgroup meets at a barrier, ONE lane adds to the arrival counter; the workgroup
// whose add came last acquires (agent scope: its L1 may hold lines of an earlier launch) and reads with plain loads.
__device__ __forceinline__ void wide_store_wt(double *p, double v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// true in every thread of the ONE workgroup whose arrival is number `members` of counter `ctr` (all threads call)
__device__ __forceinline__ bool wide_arrive_last(unsigned int *ctr, unsigned int members, int *flag_s)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every storing wave: its payload has left
    __syncthreads();
    if (threadIdx.x == 0)
    {
        const unsigned int old = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = old + 1u == members;
        if (last)
        {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        *flag_s = last;
    }
    __syncthreads();
    return *flag_s != 0;
}

// ---- the two levels of the in-launch reduction -------------------------------------------------------------------------
// The summation order of wide_reduce_kernel: group q adds the sets q, q + 16, ... in index order (level 1), then the group
// sums are added in group order (level 2).  Fixed order => bit-identical totals whichever workgroups do the adding.
// Sets and group sums are NVP = NV rounded up to even doubles apart: every load moves two values, 16 bytes, and the loads
// of a round are all in flight before the first is added.
typedef double wide_v2f64 __attribute__((ext_vector_type(2)));
template <int NVP>
__device__ __forceinline__ void wide_reduce_group(const double *partials, double *gsums, int G, int q, int tid)
{
    // one value pair per thread per trip, the 32 loads of a trip in flight together (two pairs per trip were 256 registers
    // of loads in flight: more than there are, and the loads that came back from scratch memory waited for each other)
    for (int va = tid; va < NVP / 2; va += WIDE_T)
    {
        constexpr int PER = WIDE_MAX_G / WIDE_NGRP;
        wide_v2f64 ta[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k)
        {
            const int g = q + WIDE_NGRP * k;
            ta[k] = g < G ? *reinterpret_cast<const wide_v2f64 *>(partials + (size_t)g * NVP + 2 * va) : (wide_v2f64){0.0, 0.0};
        }
        wide_v2f64 sa = {0.0, 0.0};
#pragma unroll
        for (int k = 0; k < PER; ++k)
            sa += ta[k];
        wide_store_wt(gsums + (size_t)q * NVP + 2 * va, sa[0]);
        wide_store_wt(gsums + (size_t)q * NVP + 2 * va + 1, sa[1]);
    }
}
// put(v, r0, r1): values 2 v and 2 v + 1 of the totals
template <int NVP, class PUT>
__device__ __forceinline__ void wide_reduce_final(const double *gsums, int ngrp, int tid, PUT put)
{
    for (int v0 = 0; v0 < NVP / 2; v0 += 2 * WIDE_T)
    {
        const int va = v0 + tid, vb = va + WIDE_T;
        wide_v2f64 ta[WIDE_NGRP], tb[WIDE_NGRP];
#pragma unroll
        for (int k = 0; k < WIDE_NGRP; ++k)
            ta[k] = (va < NVP / 2 && k < ngrp) ? *reinterpret_cast<const wide_v2f64 *>(gsums + (size_t)k * NVP + 2 * va)
                                               : (wide_v2f64){0.0, 0.0};
#pragma unroll
        for (int k = 0; k < WIDE_NGRP; ++k)
            tb[k] = (vb < NVP / 2 && k < ngrp) ? *reinterpret_cast<const wide_v2f64 *>(gsums + (size_t)k * NVP + 2 * vb)
                                               : (wide_v2f64){0.0, 0.0};
        wide_v2f64 ra = ta[0], rb2 = tb[0];
#pragma unroll
        for (int k = 1; k < WIDE_NGRP; ++k)
            ra += ta[k];
#pragma unroll
        for (int k = 1; k < WIDE_NGRP; ++k)
            rb2 += tb[k];
        if (va < NVP / 2)
            put(va, ra[0], ra[1]);
        if (vb < NVP / 2)
            put(vb, rb2[0], rb2[1]);
    }
}

// entry e of the four wavefronts' copies, added in wave order
template <int PW>
__device__ __forceinline__ double wide_sum4(double (*tile)[PW * WIDE_LD], int e)
{
    return ((tile[0][e] + tile[1][e]) + tile[2][e]) + tile[3][e];
}
__device__ __forceinline__ double wide_sum4g(double (*ftile)[64], int k)
{
    return ((ftile[0][k] + ftile[1][k]) + ftile[2][k]) + ftile[3][k];
}

// One pass of one workgroup over the row tiles t, t + tstride, ... (64 rows per wavefront and tile): the row phase, the
// MFMA contraction, then each wavefront's sums in its own tile (PW x PW doubles, row-major) and its own row of ftile
// (J^T f) -- wide_sum4 / wide_sum4g add the four --, ssr / non-finite flag per wavefront in red_s.  The first tile's rows arrive prefetched
// in xr_n / yy_n / sw_n; `load_rows(tt)` refills them for tile tt.  Ends behind a workgroup barrier.
template <class M, int JAC, int PW, class LOAD>
__device__ __forceinline__ void wide_rows_to_sums(const WPassArgs &a, int phase, long long t, long long tstride, long long ntile,
                                                  double (&xr_n)[M::NX], double &yy_n, double &sw_n, LOAD &&load_rows,
                                                  double (*tile)[PW * WIDE_LD], double (*ftile)[64], const double *th_s,
                                                  const double *vel_s, const double *delta_s, double (*red_s)[2], double *pre_s)
{
    constexpr int P = M::P, NX = M::NX, NB = PW / 16, NW = WIDE_T / 64, NQ = NB * (NB + 1) / 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    (void)tid;
    // developer diagnostic (GSLNLS_WIDE_STAMPS=1): where the row phase of the first row block's first wavefront goes
    unsigned long long *const rst = (a.stamps && blockIdx.x == (unsigned)a.spec && tid == 0) ? a.stamps + 10 * 1024 : nullptr;
    if (rst)
        rst[0] = __builtin_amdgcn_s_memrealtime();
    const WideTheta th{th_s};
    if constexpr (JAC == JAC_ANALYTIC && M::NPRE > 0)
    {
        // the parameter-only part of value + gradient, once for all the tiles of this wavefront (the other wavefronts of
        // the workgroup store the same values: no barrier, a wavefront reads what it has written itself)
        WidePreOut po{pre_s};
        M::prologue(th, po);
        wide_lds_sync();
    }
    const WidePre pre{pre_s};
    double *const mytile = tile[wave];
    const int kk = lane >> 4, ii = lane & 15;

    wide_v4f64 acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
        acc[q] = (wide_v4f64){0.0, 0.0, 0.0, 0.0};
    double gacc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
        gacc[b] = 0.0;
    double ssr = 0.0, bad = 0.0;

    for (; t < ntile; t += tstride)
    {
        // ---------------- row phase: lane = row ----------------
        if (rst)
            rst[8] = __builtin_amdgcn_s_memrealtime();
        const long long i = t * 64 + lane;
        const bool live = i < a.n;
        double xr[NX];
#pragma unroll
        for (int c = 0; c < NX; ++c)
            xr[c] = xr_n[c];
        const double yy = yy_n;
        const double sw = sw_n;
        if (t + tstride < ntile)
            load_rows(t + tstride); // (in flight while this tile is evaluated and contracted)
        double f;
        if constexpr (JAC == JAC_ANALYTIC)
        {
            WideTileSink sink{mytile + lane, a.wf_only ? (live ? 1.0 : 0.0) : sw, bad};
            const double m = M::value_grad_sink_pre(th, pre, xr, sink);
            bad = sink.bad;
            f = (isfinite(m) ? m - yy : INFINITY) * sw;
        }
        else
        {
            f = wide_resid<M>(th, xr, yy, sw);
            for (int j = 0; j < P; ++j)
            {
                const double d = delta_s[j];
                double col;
                if constexpr (JAC == JAC_FORWARD)
                {
                    const double fn = wide_resid<M>(WideThetaPert{th_s, j, d}, xr, yy, sw);
                    col = (fn - f) * (1.0 / d);
                }
                else
                {
                    const double fp = wide_resid<M>(WideThetaPert{th_s, j, 0.5 * d}, xr, yy, sw);
                    const double fm = wide_resid<M>(WideThetaPert{th_s, j, -0.5 * d}, xr, yy, sw);
                    col = (fp - fm) * (1.0 / d);
                }
                mytile[j * WIDE_LD + lane] = live ? col : 0.0;
            }
        }
        if (phase == PH_FVV)
        {
            // second directional derivative along the velocity at x (src/fdf.c:200-233, FD form src/fdfvv.c:35-77);
            // the accumulated vector is J^T fvv, J^T J is not needed
            double fv;
            if (a.fvv_analytic)
            {
                const double r = M::fvv(th, WideTheta{vel_s}, xr);
                bad = fma(r, 0.0, bad);
                fv = r * sw;
            }
            else
            {
                double u = 0.0;
                for (int j = 0; j < P; ++j)
                    u += mytile[j * WIDE_LD + lane] * vel_s[j];
                const double fip = wide_resid<M>(WideThetaDir{th_s, vel_s, a.h_fvv}, xr, yy, sw);
                const double hinv = 1.0 / a.h_fvv;
                fv = (2.0 * hinv) * ((fip - f) * hinv - u);
            }
            f = fv;
        }
        f = live ? f : 0.0;
        ssr += f * f;
        ftile[wave][lane] = f;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        if (rst)
        {
            rst[1] = __builtin_amdgcn_s_memrealtime(); // (last tile's) rows evaluated, gradient entries in the tile
            rst[5] = (t < tstride ? 0ull : rst[5]) + (rst[1] - rst[8]); // row phases of this pass, summed
        }
        // ---------------- contraction phase: 16 chunks of 4 rows ----------------
#if GSLNLS_WIDE_MFMA_PRIO
        __builtin_amdgcn_s_setprio(GSLNLS_WIDE_MFMA_PRIO);
#endif
        // The operands of chunk c + GSLNLS_WIDE_CHUNK_AHEAD are requested before the matrix instructions of chunk c issue
        // (round 5: with "load, wait, three MFMAs" per chunk and the phase test inside the loop every chunk paid one
        // LDS round trip in the open -- 138 clocks per MFMA where the pipe needs 64); the phase is tested once, outside.
        const auto contract = [&](auto with_jtj) {
            constexpr bool JTJ = decltype(with_jtj)::value;
            constexpr int AH = GSLNLS_WIDE_CHUNK_AHEAD;
            double v[AH + 1][NB], fl[AH + 1];
#pragma unroll
            for (int c = 0; c < AH; ++c)
            {
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    v[c][b] = mytile[(b * 16 + ii) * WIDE_LD + c * 4 + kk];
                fl[c] = ftile[wave][c * 4 + kk];
            }
#pragma unroll
            for (int c = 0; c < 16; ++c)
            {
                constexpr int R = AH + 1;
                if (c + AH < 16)
                {
#pragma unroll
                    for (int b = 0; b < NB; ++b)
                        v[(c + AH) % R][b] = mytile[(b * 16 + ii) * WIDE_LD + (c + AH) * 4 + kk];
                    fl[(c + AH) % R] = ftile[wave][(c + AH) * 4 + kk];
                }
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    gacc[b] = fma(v[c % R][b], fl[c % R], gacc[b]);
                if constexpr (JTJ)
                {
                    int q = 0;
#pragma unroll
                    for (int ba = 0; ba < NB; ++ba)
#pragma unroll
                        for (int bb = 0; bb <= ba; ++bb, ++q)
                            acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(v[c % R][ba], v[c % R][bb], acc[q], 0, 0, 0);
                }
            }
        };
        if (phase != PH_FVV)
            contract(wide_true{});
        else
            contract(wide_false{});
#if GSLNLS_WIDE_MFMA_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        if (rst)
        {
            rst[6] = (t < tstride ? 0ull : rst[6]) + (__builtin_amdgcn_s_memrealtime() - rst[1]); // contractions, summed
            rst[7] = (t < tstride ? 0ull : rst[7]) + 1;
        }
    }

    if (rst)
        rst[2] = __builtin_amdgcn_s_memrealtime(); // contraction done
    // ---------------- the four wavefronts' sums, side by side ----------------
    // Every wavefront parks its blocks in ITS OWN tile (PW x PW doubles, row-major: (i, j) -> i * PW + j; PW * PW <=
    // PW * 68) and its J^T f in its own row of ftile: no wavefront waits for another before the one barrier below.  The
    // reader adds the four copies as ((w0 + w1) + w2) + w3 (wide_sum4): the association of the wave-after-wave form this
    // replaces (three more barriers, 2.7 us at p = 32), so not a bit changes.
    ssr = wave_sum_wide(ssr);
    bad = wave_sum_wide(bad);
    if (lane == 0)
    {
        red_s[wave][0] = ssr;
        red_s[wave][1] = bad;
    }
    {
        double *const mine_full = tile[wave];
        int q = 0;
#pragma unroll
        for (int ba = 0; ba < NB; ++ba)
#pragma unroll
            for (int bb = 0; bb <= ba; ++bb, ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    mine_full[(ba * 16 + 4 * r + kk) * PW + bb * 16 + ii] = acc[q][r];
        // J^T f: entry 16 b + i is the sum over the four row groups kk of a lane column, kk ascending
#pragma unroll
        for (int b = 0; b < NB; ++b)
        {
            double s = gacc[b];
            const double s1 = wide_shfl(s, ii + 16), s2 = wide_shfl(s, ii + 32), s3 = wide_shfl(s, ii + 48);
            s = ((wide_shfl(s, ii) + s1) + s2) + s3;
            if (kk == 0)
                ftile[wave][b * 16 + ii] = s;
        }
    }
    __syncthreads();
    if (rst)
        rst[4] = __builtin_amdgcn_s_memrealtime(); // the four wavefronts' sums added
}

// FUSED = false: rows -> one partial set per workgroup, nothing else is compiled (about a second in the in-process
// compiler: what the first fit of a formula waits for).  FUSED = true: the same pass + the in-launch reduction + the LM
// step + the speculative solve (several seconds: built in the background, bound when ready -- wide_host.hpp).
template <class M, int JAC, int PW, bool FUSED>
__device__ __forceinline__ void wide_pass_body(const WPassArgs &a)
{
    constexpr int P = M::P, NX = M::NX, NB = PW / 16, NW = WIDE_T / 64, NQ = NB * (NB + 1) / 2;
    constexpr int NA = P * (P + 1) / 2, NV = 2 + NA + P, NVP = (NV + 1) & ~1; // NVP: doubles from one set to the next
    static_assert(P <= PW && PW <= 64 && PW % 16 == 0, "PW = 16 ceil(p / 16)");
    // one LDS region, two tenants: the tiles of the pass, then (in the one workgroup that runs the LM step, and in the
    // speculator) the working set of wide_advance followed by the totals of the pass
    constexpr int TILE_D = NW * PW * WIDE_LD, ADV_D = (int)(sizeof(WideLds) / sizeof(double)) + ((NV + 1) & ~1);
    constexpr int LDS_D = (!FUSED || TILE_D > ADV_D) ? TILE_D : ADV_D;
    __shared__ __attribute__((aligned(16))) double lds_raw[LDS_D];
    __shared__ double ftile[NW][64];
    __shared__ double th_s[P], vel_s[P], delta_s[P];
    __shared__ double pre_s[M::NPRE > 0 ? M::NPRE : 1];
    __shared__ double red_s[NW][2];
    __shared__ int flag_s;
    double(*const tile)[PW * WIDE_LD] = reinterpret_cast<double(*)[PW * WIDE_LD]>(lds_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const WState *S = a.state;
    // everything this workgroup needs from the previous launch is asked for at once (the state was written on another
    // XCD: a load is most of two microseconds, and phase -> theta -> rows one after the other was six)
    double st_x = 0.0, st_xt = 0.0, st_vel = 0.0;
    if (tid < P)
    {
        st_x = S->x[tid];
        st_xt = S->xt[tid];
        st_vel = S->vel[tid];
    }
    const int phase = S->phase;
    const int rb = FUSED ? (int)blockIdx.x - a.spec : (int)blockIdx.x; // row-block index; -1: the speculator
    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto stamp = [&](int k) {
        if (FUSED && a.stamps)
            st[k] = __builtin_amdgcn_s_memrealtime();
    };
    stamp(0);
    // ---- roles that end in the damped solve: the speculator now, the workgroup that completes the totals later ----
    bool solver = false;       // this wavefront runs the solve below
    const bool is_spec = FUSED && rb < 0;
    WideLds &L = *reinterpret_cast<WideLds *>(lds_raw);
    double *const totals_s = lds_raw + sizeof(WideLds) / sizeof(double);
    WideCtx ctx;
    double mu_solve = 0.0;
    double *sol_dst = nullptr;
    WAdvanceArgs adv = a.adv;
    if (is_spec)
    {
        // A rejected trial leaves x, J^T J, D, g as they are and multiplies mu by nu (src/trust.c:530-545): the damped
        // system of the step after a rejection is known before the pass has run.  This workgroup solves it while the
        // others stream the rows; the LM step takes the solution if it rejects (bit for bit what it would compute).
        if (wave != 0 || phase != PH_TRIAL) // (PH_DONE included: a launch past the end of the fit)
            return;
        const int p = P;
        const bool mine = lane < p;
        {
            constexpr int IT = (NA + 63) / 64;
            double ba[IT];
#pragma unroll
            for (int i = 0; i < IT; ++i)
                ba[i] = lane + 64 * i < NA ? S->A[lane + 64 * i] : 0.0;
            const double dg = mine ? S->diag[lane] : 0.0, gg = mine ? S->g[lane] : 0.0;
            mu_solve = S->mu * S->nu;
#pragma unroll
            for (int i = 0; i < IT; ++i)
                if (lane + 64 * i < NA)
                    L.A[lane + 64 * i] = ba[i];
            if (mine)
            {
                L.diag[lane] = dg;
                L.rhs[lane] = -gg;
            }
        }
        wide_lds_sync();
        solver = true;
        sol_dst = L.sol;
    }
    else
    {
    // the rows of this wavefront's first tile: requested before the state is waited for
    const long long ntile = (a.n + 63) / 64;
    const int nrb = FUSED ? (int)gridDim.x - a.spec : (int)gridDim.x;
    const long long tstride = (long long)nrb * NW;
    long long t = (long long)rb * NW + wave;
    double xr_n[NX], yy_n = 0.0, sw_n = 0.0;
    auto load_rows = [&](long long tt) {
        const long long i = tt * 64 + lane;
        const bool live = i < a.n;
        const long long ic = live ? i : a.n - 1;
#pragma unroll
        for (int c = 0; c < NX; ++c)
            xr_n[c] = a.x[(size_t)c * a.n + ic];
        yy_n = a.y[ic];
        sw_n = live ? (a.sw ? a.sw[ic] : 1.0) : 0.0;
    };
#pragma unroll
    for (int c = 0; c < NX; ++c)
        xr_n[c] = 0.0;
    if (t < ntile)
        load_rows(t);
    if (phase == PH_DONE)
        return;
    if (tid < P)
    {
        const double t0 = (phase == PH_FVV) ? st_x : st_xt;
        th_s[tid] = t0;
        vel_s[tid] = st_vel;
        double d = a.h_df * fabs(t0); // src/fdjac.c:36-38
        if (d == 0.0)
            d = a.h_df;
        delta_s[tid] = d;
    }
    // the padding columns of the tiles (P .. PW - 1) are zeroed once; the row phase writes every other entry
    if constexpr (P < PW)
        for (int e = lane; e < (PW - P) * WIDE_LD; e += 64)
            tile[wave][P * WIDE_LD + e] = 0.0;
    __syncthreads();
    wide_rows_to_sums<M, JAC, PW>(a, phase, t, tstride, ntile, xr_n, yy_n, sw_n, load_rows, tile, ftile, th_s, vel_s, delta_s, red_s, pre_s);
    static_assert(NW == 4, "wide_sum4 adds four wavefronts");
    stamp(1); // rows done, workgroup sums staged
    double *out = a.partials + (size_t)rb * NVP;
    const bool fuse = FUSED && a.fuse != WIDE_FUSE_NONE;
    if (tid == 0)
    {
        double s0 = red_s[0][0], s1 = red_s[0][1];
        for (int w = 1; w < NW; ++w)
        {
            s0 += red_s[w][0];
            s1 += red_s[w][1];
        }
        if (fuse)
        {
            wide_store_wt(out, s0);
            wide_store_wt(out + 1, s1);
        }
        else
        {
            out[0] = s0;
            out[1] = s1;
        }
    }
    for (int e = tid; e < NA; e += WIDE_T)
    {
        // packed index e -> (i, j), j <= i
        int i = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
        while (i * (i + 1) / 2 > e)
            --i;
        while ((i + 1) * (i + 2) / 2 <= e)
            ++i;
        const int j = e - i * (i + 1) / 2;
        const double v = wide_sum4<PW>(tile, i * PW + j);
        if (fuse)
            wide_store_wt(out + 2 + e, v);
        else
            out[2 + e] = v;
    }
    for (int k = tid; k < P; k += WIDE_T)
    {
        const double v = wide_sum4g(ftile, k);
        if (fuse)
            wide_store_wt(out + 2 + NA + k, v);
        else
            out[2 + NA + k] = v;
    }
    if (fuse && NVP != NV && tid == 0)
        wide_store_wt(out + NV, 0.0); // (the pad of an odd set: read, never used)
    if (!fuse)
        return;
    if constexpr (FUSED)
    {
    // ---------------- the G partial sets -> one, inside the launch ----------------
    // Two levels, the summation order of wide_reduce_kernel (group q adds the sets q, q + 16, ... in index order, then the
    // group sums are added in group order): the LAST workgroup of group q to arrive adds the group's sets, the last of
    // those group reducers adds the group sums.  Fixed order => bit-identical totals whichever workgroups do the adding.
    const int G = a.G, q = rb % WIDE_NGRP, ngrp = G < WIDE_NGRP ? G : WIDE_NGRP;
    const unsigned int members = (unsigned int)((G - q + WIDE_NGRP - 1) / WIDE_NGRP);
    if (!wide_arrive_last(&a.fb->tickets[q], members, &flag_s))
        return;
    stamp(2); // last of its group: acquired
    wide_reduce_group<NVP>(a.partials, a.gsums, G, q, tid);
    stamp(3); // group sums stored
    // one of the (up to) 16 group reducers will run the LM step: each asks for the state now, so that the loads (written by
    // the previous launch on another XCD: most of two microseconds) are under way while the arrival is counted
    WideStateRegs<(NA + 63) / 64> sregs;
    // (... and for the speculator's result, which by now has been there for a while: tag, then the payload.  Nothing orders
    // these loads against each other -- the fold in the tag and the bitwise comparison of mu below are what make a
    // half-new payload "not there")
    unsigned long long pre_tag = 0, pre_tag2 = ~0ull;
    double pre_sv = 0.0, pre_smu = 0.0;
    if (a.fuse == WIDE_FUSE_STEP && wave == 0)
    {
        wide_state_load<P>(adv.state, lane, sregs);
        if (a.spec)
        {
            pre_tag = __hip_atomic_load(&a.fb->spec_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pre_sv = lane < P ? __hip_atomic_load(&a.fb->spec_vel[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
            pre_smu = __hip_atomic_load(&a.fb->spec_mu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // (round 5, seqlock style: the tag once more BEHIND the payload -- a wavefront's loads are issued and returned in
            // order, so a payload read between two equal tags of this launch is the payload that tag was written behind)
            pre_tag2 = __hip_atomic_load(&a.fb->spec_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (!wide_arrive_last(&a.fb->ticket2, (unsigned int)ngrp, &flag_s))
        return;
    stamp(4); // last group reducer: acquired
    wide_reduce_final<NVP>(a.gsums, ngrp, tid, [&](int v, double r0, double r1) {
        if (a.fuse == WIDE_FUSE_STEP)
        {
            totals_s[2 * v] = r0; // (behind the working set of the step: the tiles are dead)
            totals_s[2 * v + 1] = r1;
        }
        else
        {
            a.totals[2 * v] = r0;
            if (2 * v + 1 < NV)
                a.totals[2 * v + 1] = r1;
        }
    });
    // the counters go back to zero for the next launch (which starts after this one has ended)
    if (tid <= WIDE_NGRP)
        __hip_atomic_store(tid < WIDE_NGRP ? &a.fb->tickets[tid] : &a.fb->ticket2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (a.fuse != WIDE_FUSE_STEP || wave != 0)
        return;
    // ---------------- the LM step, by the first wavefront of this workgroup ----------------
    stamp(5); // totals in LDS
    adv.totals = totals_s;
    wide_advance_pre<P, WideStateRegs<(NA + 63) / 64>>(adv, L, ctx, &sregs);
    if (!ctx.active)
        return;
    stamp(6); // decision taken, right-hand side ready
    if (ctx.want)
    {
        solver = true;
        mu_solve = ctx.mu;
        sol_dst = ctx.want == 1 ? L.acc : L.vel;
        if (a.spec && ctx.want == 2 && ctx.rejected && ctx.phase_before == PH_TRIAL)
        {
            // the speculator has had the whole pass for this solve: take its result when it is there (bounded wait: it
            // started before this workgroup did), else solve here.  {epoch, fold} guards against a payload that is not this
            // launch's: a mismatch of either is treated as "not there".
            const unsigned int epoch = ((unsigned int)adv.seq << 16) ^ (unsigned int)(adv.launch_idx + 1);
            auto fold_of = [&](double sv) {
                unsigned long long h = lane < P ? (unsigned long long)__double_as_longlong(sv) : 0ull;
                h = (h >> 32) ^ (h & 0xffffffffull);
                unsigned int f = (unsigned int)h;
#pragma unroll
                for (int m = 32; m >= 1; m >>= 1)
                    f ^= (unsigned int)__shfl_xor((int)f, m, 64);
                return f;
            };
            auto take = [&](unsigned long long tag, unsigned long long tag_behind, double sv, double smu) {
                if (tag == tag_behind && (unsigned int)(tag >> 32) == epoch && fold_of(sv) == (unsigned int)(tag & 0xffffffffull) &&
                    __double_as_longlong(smu) == __double_as_longlong(ctx.mu))
                {
                    if (lane < P)
                        L.vel[lane] = sv;
                    wide_lds_sync();
                    solver = false;
                    return true;
                }
                return false;
            };
            // what was asked for before the arrival (the common case: the speculator finished 10 us ago) ...
            if (!take(pre_tag, pre_tag2, pre_sv, pre_smu))
            {
                // ... else wait for the tag of this launch, then fetch the payload
                const unsigned long long t0 = __builtin_readcyclecounter();
                bool got = false;
                unsigned long long tag = 0;
                for (;;)
                {
                    tag = __hip_atomic_load(&a.fb->spec_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((unsigned int)(tag >> 32) == epoch)
                    {
                        got = true;
                        break;
                    }
                    if (__builtin_readcyclecounter() - t0 > 200000ull) // ~100 us: the speculator is not coming
                        break;
                    __builtin_amdgcn_s_sleep(8);
                }
                if (got)
                {
                    const double sv = lane < P ? __hip_atomic_load(&a.fb->spec_vel[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
                    const double smu = __hip_atomic_load(&a.fb->spec_mu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned long long tag2 = __hip_atomic_load(&a.fb->spec_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    (void)take(tag, tag2, sv, smu);
                }
            }
        }
    }
    }
    }
    if constexpr (FUSED)
    {
    // ---------------- ONE call site of the damped solve: the LM step's, or the speculator's ----------------
    if (solver)
        wide_solve_pw<PW>(L, P, mu_solve, L.rhs, sol_dst, lane, adv.pivoted);
    if (is_spec)
    {
        const double sv = lane < P ? L.sol[lane] : 0.0;
        unsigned long long h = lane < P ? (unsigned long long)__double_as_longlong(sv) : 0ull;
        h = (h >> 32) ^ (h & 0xffffffffull);
        unsigned int f = (unsigned int)h;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1)
            f ^= (unsigned int)__shfl_xor((int)f, m, 64);
        if (a.spec_fault == 3)
            return; // (test: the speculator never publishes -- the stepping workgroup's bounded wait runs out)
        if (lane < P) // (test, fault 1: a payload whose first half is not the one the tag was folded from)
            wide_store_wt(&a.fb->spec_vel[lane], (a.spec_fault == 1 && lane < (P + 1) / 2) ? sv * 1.0000001 + 1e-9 : sv);
        if (lane == 0)
            wide_store_wt(&a.fb->spec_mu, mu_solve);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the payload has left before the tag says so
        unsigned int epoch = ((unsigned int)adv.seq << 16) ^ (unsigned int)(adv.launch_idx + 1);
        if (a.spec_fault == 2)
            epoch ^= 0x00010001u; // (test: the tag of some other launch)
        if (lane == 0)
            __hip_atomic_store(&a.fb->spec_tag, ((unsigned long long)epoch << 32) | f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    stamp(7); // solve done (or the speculator's result taken)
    wide_advance_post<P>(adv, L, ctx);
    if (a.stamps && lane == 0)
    {
        unsigned long long *o = a.stamps + (size_t)(adv.launch_idx & 1023) * 10;
        for (int k = 0; k < 8; ++k)
            o[k] = st[k];
        o[8] = __builtin_amdgcn_s_memrealtime();
        o[9] = (unsigned long long)ctx.want | ((unsigned long long)(solver ? 1 : 0) << 8) | ((unsigned long long)ctx.rejected << 16);
    }
    }
}

template <class M, int JAC, int PW>
__global__ __launch_bounds__(WIDE_T, GSLNLS_WIDE_WAVES(PW)) void wide_pass_kernel(WPassArgs a)
{
    wide_pass_body<M, JAC, PW, false>(a);
}
template <class M, int JAC, int PW>
__global__ __launch_bounds__(WIDE_T, GSLNLS_WIDE_WAVES(PW)) void wide_step_kernel(WPassArgs a)
{
    wide_pass_body<M, JAC, PW, true>(a);
}

// ---- one workgroup = one complete fit -----------------------------------------------------------------------------------
// The per-point work of gsl_multistart_driver for 10 <= p <= 64 (src/nls_mstart.c:42-128, :245-255): det filter at the
// sampled point (det_eval_jtj, src/nls_utils.c:23-73), a short LM fit, det where it ended -- every sample point of a
// batch at once instead of one after the other: workgroup f owns point f for its whole fit.  A trial step is the pass
// over the n rows by the four wavefronts (wide_rows_to_sums: the workgroup's own sums ARE the totals), then the LM step by
// the first wavefront (wide_advance_pre / solve / post, its state in HBM, touched by that wavefront only), a workgroup
// barrier instead of a launch boundary.  Records are written position-addressed, (3 p + 8) doubles per point, as
// ms_fit_kernel writes them (batch_kernels.hpp): x, diag, x0, chisq0, chisq1, det0, det1, ssr at the start, niter,
// status, nevalf.
struct WFitArgs
{
    WPassArgs pass;        // x, y, sw, n, h_df, h_fvv, fvv_analytic, wf_only (the other fields are not read)
    LmParams prm;          // of the short fits (maxiter = mstart_p, gtol = 1e-3: src/nls_mstart.c:91, :254)
    int pivoted, nfit, always_fit, has_bounds;
    double dtol;
    const double *starts;  // [nfit][p]
    const double *lupars;  // [2 p] lower / upper pairs, or nullptr
    WState *states;        // [nfit]: one state per workgroup
    double *records;       // [nfit][3 p + 8]
};

template <class M, int JAC, int PW>
__global__ __launch_bounds__(WIDE_T, GSLNLS_WIDE_WAVES(PW)) void wide_fit_kernel(WFitArgs fa)
{
    constexpr int P = M::P, NX = M::NX, NW = WIDE_T / 64;
    constexpr int NA = P * (P + 1) / 2, NV = 2 + NA + P;
    constexpr int TILE_D = NW * PW * WIDE_LD, ADV_D = (int)(sizeof(WideLds) / sizeof(double)) + ((NV + 1) & ~1);
    constexpr int LDS_D = TILE_D > ADV_D ? TILE_D : ADV_D;
    __shared__ __attribute__((aligned(16))) double lds_raw[LDS_D];
    __shared__ double ftile[NW][64];
    __shared__ double th_s[P], vel_s[P], delta_s[P];
    __shared__ double pre_s[M::NPRE > 0 ? M::NPRE : 1];
    __shared__ double red_s[NW][2];
    __shared__ int phase_s;
    double(*const tile)[PW * WIDE_LD] = reinterpret_cast<double(*)[PW * WIDE_LD]>(lds_raw);
    WideLds &L = *reinterpret_cast<WideLds *>(lds_raw);
    double *const totals_s = lds_raw + sizeof(WideLds) / sizeof(double);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int f = blockIdx.x;
    if (f >= fa.nfit)
        return;
    const WPassArgs &a = fa.pass;
    WState *S = fa.states + f;
    double *rec = fa.records + (size_t)f * (3 * P + 8);
    const double *st0 = fa.starts + (size_t)f * P;
    // ---- lm_state_reset: the start state of the fit (by the wavefront that will run its steps) ----
    if (wave == 0)
    {
        if (lane < P)
        {
            const double v = st0[lane];
            S->x[lane] = S->xt[lane] = v;
            S->dx[lane] = S->vel[lane] = S->acc[lane] = S->g[lane] = 0.0;
            S->diag[lane] = 1.0;
            const double lo = fa.lupars ? fa.lupars[2 * lane] : -INFINITY, up = fa.lupars ? fa.lupars[2 * lane + 1] : INFINITY;
            S->lo[lane] = isfinite(lo) ? lo : -INFINITY;
            S->up[lane] = isfinite(up) ? up : INFINITY;
            rec[2 * P + lane] = v; // x0
        }
        for (int k = lane; k < NA; k += 64)
            S->A[k] = 0.0;
        if (lane == 0)
        {
            S->fnorm2 = INFINITY;
            S->mu = 0.0;
            S->nu = 2.0;
            S->delta = 0.0;
            S->avratio = 0.0;
            S->chisq0 = S->chisq1 = S->chisq_init = INFINITY;
            S->bad_steps = S->niter = 0;
            S->phase = PH_INIT;
            S->status = S->info = ST_CONTINUE;
            S->nevalf = S->nevaldf = S->nevalfvv = 0;
            S->p = P;
            S->end_launch = 0;
        }
    }
    if (tid < P)
    {
        const double v = st0[tid];
        th_s[tid] = v;
        vel_s[tid] = 0.0;
        double d = a.h_df * fabs(v);
        if (d == 0.0)
            d = a.h_df;
        delta_s[tid] = d;
    }
    if (tid == 0)
        phase_s = PH_INIT;
    __syncthreads();
    WAdvanceArgs adv;
    adv.state = S;
    adv.totals = totals_s;
    adv.prm = fa.prm;
    adv.ssrtrace = nullptr;
    adv.partrace = nullptr;
    adv.host_mirror = nullptr; // (no host copy, no completion word: the record is cut from the state below)
    adv.done_seq = nullptr;
    adv.seq = 0;
    adv.launch_idx = 0;
    adv.p = P;
    adv.pivoted = fa.pivoted;
    const long long ntile = (a.n + 63) / 64;
    bool first = true;
    for (int step = 0; step < (1 << 20); ++step)
    {
        const int phase = phase_s;
        // the padding columns of the tiles (the LM step used the region: zero them again)
        if constexpr (P < PW)
            for (int e = lane; e < (PW - P) * WIDE_LD; e += 64)
                tile[wave][P * WIDE_LD + e] = 0.0;
        double xr_n[NX], yy_n = 0.0, sw_n = 0.0;
        auto load_rows = [&](long long tt) {
            const long long i = tt * 64 + lane;
            const bool live = i < a.n;
            const long long ic = live ? i : a.n - 1;
#pragma unroll
            for (int c = 0; c < NX; ++c)
                xr_n[c] = a.x[(size_t)c * a.n + ic];
            yy_n = a.y[ic];
            sw_n = live ? (a.sw ? a.sw[ic] : 1.0) : 0.0;
        };
#pragma unroll
        for (int c = 0; c < NX; ++c)
            xr_n[c] = 0.0;
        if (wave < ntile)
            load_rows(wave);
        __syncthreads();
        wide_rows_to_sums<M, JAC, PW>(a, phase, (long long)wave, (long long)NW, ntile, xr_n, yy_n, sw_n, load_rows, tile, ftile, th_s,
                                      vel_s, delta_s, red_s, pre_s);
        // the workgroup's sums are the totals of the pass: ssr, non-finite flag, packed lower J^T J, J^T f
        {
            if (tid == 0)
            {
                double s0 = red_s[0][0], s1 = red_s[0][1];
                for (int w = 1; w < NW; ++w)
                {
                    s0 += red_s[w][0];
                    s1 += red_s[w][1];
                }
                totals_s[0] = s0;
                totals_s[1] = s1;
            }
            for (int e = tid; e < NA; e += WIDE_T)
            {
                int i = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
                while (i * (i + 1) / 2 > e)
                    --i;
                while ((i + 1) * (i + 2) / 2 <= e)
                    ++i;
                const int j = e - i * (i + 1) / 2;
                totals_s[2 + e] = wide_sum4<PW>(tile, i * PW + j);
            }
            for (int k = tid; k < P; k += WIDE_T)
                totals_s[2 + NA + k] = wide_sum4g(ftile, k);
        }
        __syncthreads();
        if (wave == 0)
        {
            const int lane = wide_lane(); // (the LDS addresses of this step are formed in this step: wide_core.hpp)
            bool go = true;
            if (first)
            {
                // det_eval_jtj at the sampled point; the fit is only run where the filter lets the point through
                double det0 = wide_det_reg<PW>(totals_s + 2, P, lane);
                if (JAC == JAC_ANALYTIC && !(totals_s[1] == 0.0))
                    det0 = 0.0; // eval_df failed (src/nls_utils.c:47-48)
                go = fa.always_fit || det0 > fa.dtol;
                if (lane == 0)
                {
                    rec[3 * P + 2] = det0;
                    rec[3 * P + 4] = totals_s[0];
                }
                if (!go)
                {
                    if (lane < P)
                    {
                        rec[lane] = st0[lane];
                        rec[P + lane] = 1.0;
                    }
                    if (lane == 0)
                    {
                        rec[3 * P + 0] = INFINITY;
                        rec[3 * P + 1] = totals_s[0];
                        rec[3 * P + 3] = 0.0;
                        rec[3 * P + 5] = 0.0;
                        rec[3 * P + 6] = (double)ST_CONTINUE;
                        rec[3 * P + 7] = 1.0;
                        phase_s = -1;
                    }
                }
            }
            if (go)
            {
                WideCtx ctx;
                wide_advance_pre<P>(adv, L, ctx);
                if (ctx.active && ctx.want)
                    wide_solve_pw<PW>(L, P, ctx.mu, L.rhs, ctx.want == 1 ? L.acc : L.vel, lane, adv.pivoted);
                wide_advance_post<P>(adv, L, ctx);
                // the next pass reads its point from LDS: the step's own copies of x, xt, vel are still there
                const int nphase = S->phase;
                if (lane < P)
                {
                    const double t0 = (nphase == PH_FVV) ? L.x[lane] : L.xt[lane];
                    th_s[lane] = t0;
                    vel_s[lane] = L.vel[lane];
                    double d = a.h_df * fabs(t0);
                    if (d == 0.0)
                        d = a.h_df;
                    delta_s[lane] = d;
                }
                if (lane == 0)
                    phase_s = nphase;
                if (nphase == PH_DONE)
                {
                    // the record of a fitted point
                    const double det1 = wide_det_reg<PW>(L.A, P, lane);
                    if (lane < P)
                    {
                        rec[lane] = L.x[lane];
                        rec[P + lane] = L.diag[lane];
                    }
                    if (lane == 0)
                    {
                        rec[3 * P + 0] = S->chisq0;
                        rec[3 * P + 1] = S->chisq1;
                        rec[3 * P + 3] = det1;
                        rec[3 * P + 5] = (double)S->niter;
                        rec[3 * P + 6] = (double)S->status;
                        rec[3 * P + 7] = (double)S->nevalf;
                    }
                }
            }
        }
        first = false;
        __syncthreads();
        if (phase_s == PH_DONE || phase_s < 0)
            return;
    }
}

// After the fit: weighted residual and Jacobian at the final point in the layout C_nls returns them (resid n; grad
// n x p column-major, src/nls.c:695-737)
template <class M, int JAC>
__global__ __launch_bounds__(256) void wide_finalize_kernel(WPassArgs a, double *resid, double *grad)
{
    constexpr int P = M::P, NX = M::NX;
    __shared__ double th_s[P], delta_s[P];
    __shared__ double pre_s[M::NPRE > 0 ? M::NPRE : 1];
    const WState *S = a.state;
    for (int k = threadIdx.x; k < P; k += 256)
    {
        const double t = S->x[k];
        th_s[k] = t;
        double d = a.h_df * fabs(t);
        if (d == 0.0)
            d = a.h_df;
        delta_s[k] = d;
    }
    __syncthreads();
    const WideTheta th{th_s};
    if constexpr (JAC == JAC_ANALYTIC && M::NPRE > 0)
    {
        if (grad)
        {
            WidePreOut po{pre_s}; // (every thread stores the same values)
            M::prologue(th, po);
        }
        __syncthreads();
    }
    const WidePre pre{pre_s};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (long long)gridDim.x * 256)
    {
        double xr[NX];
#pragma unroll
        for (int c = 0; c < NX; ++c)
            xr[c] = a.x[(size_t)c * a.n + i];
        const double yy = a.y[i], sw = a.sw ? a.sw[i] : 1.0;
        double f;
        if constexpr (JAC == JAC_ANALYTIC)
        {
            if (grad)
            {
                WideGradSink sink{grad + i, a.n, sw};
                const double m = M::value_grad_sink_pre(th, pre, xr, sink);
                f = (isfinite(m) ? m - yy : INFINITY) * sw;
            }
            else
                f = wide_resid<M>(th, xr, yy, sw);
        }
        else
        {
            f = wide_resid<M>(th, xr, yy, sw);
            if (grad)
                for (int j = 0; j < P; ++j)
                {
                    const double d = delta_s[j];
                    double col;
                    if constexpr (JAC == JAC_FORWARD)
                        col = (wide_resid<M>(WideThetaPert{th_s, j, d}, xr, yy, sw) - f) * (1.0 / d);
                    else
                        col = (wide_resid<M>(WideThetaPert{th_s, j, 0.5 * d}, xr, yy, sw) -
                               wide_resid<M>(WideThetaPert{th_s, j, -0.5 * d}, xr, yy, sw)) *
                              (1.0 / d);
                    grad[i + (size_t)a.n * j] = col;
                }
        }
        if (resid)
            resid[i] = f;
    }
}

} // namespace gslnls
