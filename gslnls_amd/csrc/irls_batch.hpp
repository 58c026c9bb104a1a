// irls_batch.hpp -- workgroup-per-dataset robust fits (BASELINE config C5: thousands of independent
// data sets, each n ~ 1e4 rows, p = 8, loss = "bisquare").
//
// The reference has no batched form: a user would call gsl_nls(loss = ...) once per data set, and each
// call runs gsl_multifit_nlinear_rho_driver (src/nls_irls.c:412-546): cold LM solve from the original
// start -> unweighted residuals -> sigma = 1.4826 median|r| (full sort) -> new weights -> repeat.
// Here ONE workgroup owns one data set for the whole procedure and never leaves the kernel:
//   * LM loop: every thread streams its rows (x, y, sqrt w from L2/Infinity Cache), workgroup
//     reduction through LDS (fixed order), wavefront 0 runs the same lm_advance() as every other path
//     and broadcasts the next trial point through LDS -- the workgroup barrier replaces the kernel
//     boundary of the grid-per-fit path;
//   * re-weighting: |r_i| bit patterns to a scratch array, 8-pass radix SELECT with an LDS histogram
//     for the median (two order statistics when n is even), psi(r/sigma)/(r/sigma) weights normalised
//     to sum n, sqrt into the weight array of the next solve;
//   * stopping rule test_delta_irls (src/nls_irls.c:343-362) by wavefront 0.
// Data sets are independent => sharding over GPUs is a contiguous split of the batch, no collective.
#pragma once
#include <hip/hip_runtime.h>
#include "dense_kernels.hpp"
#include "irls_core.hpp"

namespace gslnls
{

template <int P>
struct IrlsBatchArgs
{
    const double *x;      // [B][NX][n]
    const double *y;      // [B][n]
    const double *usw;    // [B][n] sqrt(user weights) or nullptr
    double *sw;           // [B][n] scratch: sqrt weights of the current solve
    unsigned long long *keys; // [B][n] scratch: bits of |r_i|
    int n, lo, hi;        // rows per data set; data sets [lo, hi) of the batch
    double start[P];
    double lu[2 * P];
    int has_lu;
    LmParams prm;
    LossCfg loss;
    int irls_maxiter;
    double irls_xtol;
    // outputs, indexed by data set
    double *par;          // [B][P]
    double *scal;         // [B][4]: sigma, ssr (weighted), irls_tol, chisq_init
    int *ints;            // [B][4]: conv, irls_status, irls_niter, niter
    unsigned long long *prof; // developer diagnostic (GSLNLS_BATCH_PROF): [B][8] cycle totals, or nullptr
    unsigned long long *pass_total; // [2]: passes over the rows by the LM solves / by the re-weightings, summed over the
                                    // data sets of the call (the roofline accounting of bench.py), or nullptr
};

template <class M, int JAC, int T>
__global__ __launch_bounds__(T, 2) void irls_batch_kernel(IrlsBatchArgs<M::P> a)
{
    constexpr int P = M::P, NX = M::NX;
    using Sums = PassSums<P>;
    constexpr int NV = Sums::NV, NW = T / 64;
    constexpr int GV = 3 * NW; // values per group of the workgroup reduction (12 for 256 threads: 24 KB of LDS)
    __shared__ double lds_red[NW * NV];
    __shared__ double lds_grp[GV * T];
    __shared__ double lds_tot[NV];
    __shared__ LmState<P> lds_state;
    __shared__ LmParams lds_prm;
    __shared__ unsigned int hist[256];
    __shared__ unsigned long long sel_prefix, sel_mask, sel_k, sel_lo, sel_hi;
    __shared__ unsigned int sel_cnt_le, sel_cand;
    __shared__ double sh_val[2], sh_sigma, sh_scale;
    __shared__ int sh_flag;
    // bracketed median (IRLS iteration >= 2): the keys that fall inside a narrow bracket around the previous median
    constexpr int BR_CAP = 2048;
    __shared__ unsigned long long br_list[BR_CAP];
    __shared__ unsigned int br_n, br_below;

    const int d = a.lo + blockIdx.x;
    if (d >= a.hi)
        return;
    unsigned long long pf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto now = [&]() -> unsigned long long { return a.prof ? (unsigned long long)__builtin_amdgcn_s_memtime() : 0ull; };
    const unsigned long long t_begin = now();
    const int n = a.n, tid = threadIdx.x;
    const double *xd = a.x + (size_t)d * NX * n;
    const double *yd = a.y + (size_t)d * n;
    const double *ud = a.usw ? a.usw + (size_t)d * n : nullptr;
    double *swd = a.sw + (size_t)d * n;
    unsigned long long *kd = a.keys + (size_t)d * n;

    for (int i = tid; i < n; i += T)
        swd[i] = ud ? ud[i] : 1.0;
    __syncthreads();

    double workp[P];
#pragma unroll
    for (int k = 0; k < P; ++k)
        workp[k] = a.start[k];
    int irls_iter = 0, irls_status = ST_FAILURE, status = ST_CONTINUE;
    unsigned int n_lm_pass = 0, n_rw = 0;
    double chisq_carry = NAN, chisq_init = NAN, sigma = 1.0, sigma_change = 1.0;
    LmParams prm = a.prm;
    prm.has_weights = 1;

    for (;;)
    {
        irls_iter += 1;
        // ---------------- cold LM solve from the original start with the current weights ----------------
        prm.chisq_in = (irls_iter > 1) ? chisq_carry : NAN;
        if (tid == 0)
        {
            lm_state_reset<P>(lds_state, a.start, a.has_lu ? a.lu : nullptr);
            lds_prm = prm; // what the out-of-line state machine reads (LDS, like the state)
        }
        __syncthreads();
        for (int guard = 0; guard < 1000000; ++guard)
        {
            const int phase = lds_state.phase;
            if (phase == PH_DONE)
                break;
            double th[P], vel[P], delta[P];
#pragma unroll
            for (int k = 0; k < P; ++k)
            {
                th[k] = (phase == PH_FVV) ? lds_state.x[k] : lds_state.xt[k];
                vel[k] = lds_state.vel[k];
            }
            fd_deltas<P>(th, prm.h_df, delta);
            const unsigned long long t0 = now();
            Sums acc;
            pass_zero<P>(acc);
            // the phase is uniform: one loop per kind of pass (both bodies inside one loop cost registers)
            if (phase == PH_FVV)
            {
                for (int i = tid; i < n; i += T)
                {
                    double xr[NX];
#pragma unroll
                    for (int c = 0; c < NX; ++c)
                        xr[c] = xd[(size_t)c * n + i];
                    double Jrow[P];
                    const double fv = row_fvv<M, JAC>(th, vel, delta, prm.h_fvv, prm.fvv_analytic != 0, xr, yd[i],
                                                      swd[i], Jrow, &acc.badj);
#pragma unroll
                    for (int k = 0; k < P; ++k)
                        acc.g[k] += Jrow[k] * fv;
                }
            }
            else
            {
                // software pipeline: the next row's x, y, sqrt(w) are requested before this row is evaluated
                double nx[NX], ny = 0.0, nw = 0.0;
                if (tid < n)
                {
#pragma unroll
                    for (int c = 0; c < NX; ++c)
                        nx[c] = xd[(size_t)c * n + tid];
                    ny = yd[tid];
                    nw = swd[tid];
                }
#pragma unroll 1
                for (int i = tid; i < n; i += T)
                {
                    double xr[NX];
#pragma unroll
                    for (int c = 0; c < NX; ++c)
                        xr[c] = nx[c];
                    const double yy = ny, ww = nw;
                    const int inext = i + T < n ? i + T : i;
#pragma unroll
                    for (int c = 0; c < NX; ++c)
                        nx[c] = xd[(size_t)c * n + inext];
                    ny = yd[inext];
                    nw = swd[inext];
                    double Jrow[P];
                    const double f = row_fj<M, JAC>(th, delta, xr, yy, ww, Jrow, &acc.badj);
                    acc_fj<P>(acc, f, Jrow);
                }
            }
            const unsigned long long t1 = now();
            // Workgroup sums, GV values at a time through an LDS transpose [value][thread]: wave w then owns values
            // w, w + NW, ... of the group, every lane adds the NW entries of its column and one DPP wave_sum finishes
            // a value -- GV / NW wave_sums per wave and group instead of one per value (46 DPP chains per wave were
            // 14 k cycles per pass); fixed order, so results do not depend on how the data sets are scheduled.
            {
                constexpr int NGRP = (NV + GV - 1) / GV;
                const int lane = tid & 63, wave = tid >> 6;
                const double *av = reinterpret_cast<const double *>(&acc);
#pragma unroll
                for (int gq = 0; gq < NGRP; ++gq)
                {
#pragma unroll
                    for (int v = 0; v < GV; ++v)
                        if (gq * GV + v < NV)
                            lds_grp[v * T + tid] = av[gq * GV + v];
                    __syncthreads();
                    double part[GV / NW];
#pragma unroll
                    for (int q = 0; q < GV / NW; ++q)
                    {
                        const int v = wave + q * NW;
                        part[q] = 0.0;
                        if (gq * GV + v < NV)
                        {
#pragma unroll
                            for (int w = 0; w < NW; ++w)
                                part[q] += lds_grp[v * T + w * 64 + lane];
                        }
                    }
#pragma unroll
                    for (int q = 0; q < GV / NW; ++q)
                        part[q] = wave_sum(part[q]);
#pragma unroll
                    for (int q = 0; q < GV / NW; ++q)
                    {
                        const int v = gq * GV + wave + q * NW;
                        if (lane == 0 && v < NV)
                            lds_tot[v] = part[q];
                    }
                    __syncthreads();
                }
            }
            const unsigned long long t2 = now();
            if (tid == 0)
            {
                // one lane advances the state IN PLACE in LDS: a register copy of LmState<8> (~115 doubles) plus
                // the 8 x 8 system of lm_solve pushed the kernel to 256 VGPRs + AGPR/scratch spills = one
                // wavefront per SIMD; p-sized algebra through LDS costs a few thousand cycles per LM iteration
                // but lets a second data set share the CU and hide them
                lm_advance_lds3<P>(lds_offset_of(&lds_state), lds_offset_of(lds_tot), lds_offset_of(&lds_prm));
            }
            __syncthreads();
            const unsigned long long t3 = now();
            pf[0] += t1 - t0;
            pf[1] += t2 - t1;
            pf[2] += t3 - t2;
            pf[5] += 1;
            n_lm_pass += 1;
        }
        status = lds_state.status;
        if (irls_iter == 1)
            chisq_init = lds_state.chisq_init;
        chisq_carry = lds_state.chisq1;
        if (status == ST_EBADFUNC || (status == ST_ENOPROG && irls_iter == 1))
            break;

        // ---------------- re-weighting ----------------
        const unsigned long long tw0 = now();
        n_rw += 1;
        double th[P];
#pragma unroll
        for (int k = 0; k < P; ++k)
            th[k] = lds_state.x[k];
        // From the second IRLS iteration on the median of |r| is close to the previous one: while the keys are produced,
        // the ones inside a bracket around the previous median are collected in LDS and the ones below it are counted.
        // If both middle order statistics fall inside the bracket the select runs on that short list in LDS -- the same
        // exact order statistics as a select over all keys; otherwise the select over the global keys runs.
        const bool have_br = irls_iter > 1 && sigma > 0.0 && sigma < INFINITY;
        const double med_prev = sigma * (1.0 / 1.482602218505602);
        // half-width: 12 % after the first re-weighting (the fit itself still moves), then three times the last
        // relative change of sigma (IRLS contracts), at least 0.5 %
        const double br_half = (irls_iter == 2) ? 0.12 : fmin(0.12, fmax(0.005, 3.0 * sigma_change));
        const unsigned long long b_lo = (unsigned long long)__double_as_longlong(med_prev * (1.0 - br_half)),
                                 b_hi = (unsigned long long)__double_as_longlong(med_prev * (1.0 + br_half));
        if (tid == 0)
        {
            br_n = 0;
            br_below = 0;
        }
        __syncthreads();
        unsigned int below = 0;
        // (four rows per trip, their loads first: with two wavefronts per SIMD a one-row loop exposes an L2 round trip
        // per row -- measured 1.5 k cycles per row against ~0.5 k of arithmetic)
        constexpr int RU = 4;
        for (int i0 = tid; i0 < n; i0 += RU * T)
        {
            double xr[RU][NX], yy[RU];
#pragma unroll
            for (int u = 0; u < RU; ++u)
            {
                const int i = i0 + u * T < n ? i0 + u * T : n - 1;
#pragma unroll
                for (int c = 0; c < NX; ++c)
                    xr[u][c] = xd[(size_t)c * n + i];
                yy[u] = yd[i];
            }
#pragma unroll
            for (int u = 0; u < RU; ++u)
            {
                const int i = i0 + u * T;
                if (i >= n)
                    break;
                const unsigned long long key =
                    (unsigned long long)__double_as_longlong(fabs(row_resid<M>(th, xr[u], yy[u], 1.0)));
                kd[i] = key;
                if (have_br)
                {
                    below += key < b_lo ? 1u : 0u;
                    if (key >= b_lo && key <= b_hi)
                    {
                        const unsigned int pos = atomicAdd(&br_n, 1u);
                        if (pos < (unsigned int)BR_CAP)
                            br_list[pos] = key;
                    }
                }
            }
        }
        if (have_br)
            atomicAdd(&br_below, below);
        __syncthreads();
        // median of |r| (src/nls_utils.c:162-186 sorts; here a radix SELECT on the bit patterns, which order like
        // the non-negative doubles they encode).  One select for the lower middle order statistic k_lo; for even
        // n the upper one is the smallest key above it unless the value repeats -- one extra sweep instead of a
        // second 8-pass select.  Bytes that are equal in the smallest and the largest key are skipped.  The
        // histogram of a byte is built with wave-aggregated LDS atomics: the keys of one wavefront mostly share
        // their high bytes, and 64 same-address atomics serialise.
        const unsigned long long k_lo = (unsigned long long)((n - 1) / 2), k_hi = (unsigned long long)(n / 2);
        const int nsel = (k_hi != k_lo) ? 2 : 1;
        const unsigned long long tw1 = now();
        // ---- exact order statistics by radix select (src/nls_utils.c:162-186 sorts) -------------------------------
        // select_rank(src, nk, rank): the key of that rank among src[0 .. nk) ends up in sel_prefix.  Byte-wise
        // histogram passes from the first byte in which the smallest and the largest key differ; keys order like the
        // non-negative doubles they encode.  When the source is the global key array and the keys still matching the
        // prefix are few enough, they are collected into LDS (br_list) and the remaining passes run there.
        auto select_rank = [&](const unsigned long long *src, int nk, unsigned long long rank, bool may_collect) {
            {
                unsigned long long kmin = ~0ull, kmax = 0ull;
#pragma unroll 8
                for (int i = tid; i < nk; i += T)
                {
                    const unsigned long long key = src[i];
                    kmin = key < kmin ? key : kmin;
                    kmax = key > kmax ? key : kmax;
                }
                if (tid == 0)
                {
                    sel_lo = ~0ull;
                    sel_hi = 0ull;
                }
                __syncthreads();
                atomicMin(&sel_lo, kmin);
                atomicMax(&sel_hi, kmax);
                __syncthreads();
            }
            int first_pass = 7;
            {
                const unsigned long long diff = sel_lo ^ sel_hi;
                while (first_pass > 0 && ((diff >> (8 * first_pass)) & 255ull) == 0ull)
                    --first_pass;
            }
            if (tid == 0)
            {
                const int sh = 8 * (first_pass + 1);
                sel_mask = sh >= 64 ? 0ull : ~((1ull << sh) - 1ull); // bytes above first_pass: common to all keys
                sel_prefix = sel_lo & sel_mask;
                sel_k = rank;
                sel_cand = (unsigned int)nk;
            }
            __syncthreads();
            for (int pass = first_pass; pass >= 0; --pass)
            {
                if (may_collect && sel_cand <= (unsigned int)BR_CAP)
                {
                    // few candidates left: gather them (one more sweep of the global keys), finish in LDS
                    const unsigned long long prefix = sel_prefix, mask = sel_mask;
                    if (tid == 0)
                        br_n = 0;
                    __syncthreads();
                    constexpr int KU = 8;
                    for (int i0 = 0; i0 < nk; i0 += KU * T)
                    {
                        unsigned long long key[KU];
#pragma unroll
                        for (int u = 0; u < KU; ++u)
                        {
                            const int i = i0 + u * T + tid;
                            key[u] = src[i < nk ? i : nk - 1];
                        }
#pragma unroll
                        for (int u = 0; u < KU; ++u)
                        {
                            const int i = i0 + u * T + tid;
                            if (i < nk && (key[u] & mask) == prefix)
                                br_list[atomicAdd(&br_n, 1u)] = key[u]; // at most sel_cand <= BR_CAP of them
                        }
                    }
                    __syncthreads();
                    src = br_list;
                    nk = (int)br_n;
                    may_collect = false;
                    // every listed key carries the prefix: the rank within the list is sel_k, the passes go on below
                }
                for (int bq = tid; bq < 256; bq += T)
                    hist[bq] = 0;
                __syncthreads();
                const unsigned long long prefix = sel_prefix, mask = sel_mask;
                // the first two bytes examined are shared by most keys of a wavefront: aggregate; later bytes are
                // spread over the 256 bins and plain atomics rarely collide
                const bool aggregate = pass > first_pass - 2;
                constexpr int KU = 8; // keys in flight per thread: the loads overlap instead of exposing 8 latencies
                for (int i0 = 0; i0 < nk; i0 += KU * T)
                {
                    unsigned long long key[KU];
#pragma unroll
                    for (int u = 0; u < KU; ++u)
                    {
                        const int i = i0 + u * T + tid;
                        key[u] = src[i < nk ? i : nk - 1];
                    }
#pragma unroll
                    for (int u = 0; u < KU; ++u)
                    {
                        const int i = i0 + u * T + tid;
                        int bin = -1;
                        if (i < nk && (key[u] & mask) == prefix)
                            bin = (int)((key[u] >> (8 * pass)) & 255ull);
                        if (aggregate)
                        {
                            // up to 3 rounds of "everyone with the first lane's bin adds once", then plain atomics
                            unsigned long long todo = __ballot(bin >= 0);
#pragma unroll 1
                            for (int round = 0; round < 3 && todo; ++round)
                            {
                                const int leader = __ffsll((long long)todo) - 1;
                                const int bb = __builtin_amdgcn_readlane(bin, leader);
                                const unsigned long long same = __ballot(bin == bb);
                                if ((int)(tid & 63) == leader)
                                    atomicAdd(&hist[bb], (unsigned int)__popcll(same));
                                todo &= ~same;
                                if (bin == bb)
                                    bin = -1;
                            }
                        }
                        if (bin >= 0)
                            atomicAdd(&hist[bin], 1u);
                    }
                }
                __syncthreads();
                // which bin holds rank sel_k: wavefront 0, lane l owns bins 4l .. 4l+3, exclusive scan over the lanes
                if (tid < 64)
                {
                    const unsigned int c0 = hist[4 * tid], c1 = hist[4 * tid + 1], c2 = hist[4 * tid + 2], c3 = hist[4 * tid + 3];
                    unsigned int incl = c0 + c1 + c2 + c3;
#pragma unroll
                    for (int dlt = 1; dlt < 64; dlt <<= 1)
                    {
                        const unsigned int up = __shfl_up(incl, dlt);
                        if (tid >= dlt)
                            incl += up;
                    }
                    const unsigned long long k = sel_k;
                    const unsigned long long before = incl - (c0 + c1 + c2 + c3);
                    if (k >= before && k < incl)
                    {
                        unsigned long long cum = before;
                        int bin = 4 * tid;
                        unsigned int cb = c0;
                        if (k >= cum + c0)
                        {
                            cum += c0;
                            bin += 1;
                            cb = c1;
                            if (k >= cum + c1)
                            {
                                cum += c1;
                                bin += 1;
                                cb = c2;
                                if (k >= cum + c2)
                                {
                                    cum += c2;
                                    bin += 1;
                                    cb = c3;
                                }
                            }
                        }
                        sel_k = k - cum;
                        sel_cand = cb;
                        sel_prefix = prefix | ((unsigned long long)bin << (8 * pass));
                        sel_mask = mask | (255ull << (8 * pass));
                    }
                }
                __syncthreads();
            }
        };
        // upper_middle(src, nk, v_lo, rank_hi): the key of rank rank_hi = rank(v_lo) + 1 among src[0 .. nk): v_lo again if
        // enough keys are <= it, else the smallest key above it -- one sweep instead of a second select
        auto upper_middle = [&](const unsigned long long *src, int nk, unsigned long long v_lo, unsigned long long rank_hi) {
            unsigned int cnt_le = 0;
            unsigned long long next = ~0ull;
#pragma unroll 8
            for (int i = tid; i < nk; i += T)
            {
                const unsigned long long key = src[i];
                cnt_le += key <= v_lo ? 1u : 0u;
                if (key > v_lo && key < next)
                    next = key;
            }
            if (tid == 0)
            {
                sel_lo = ~0ull;
                sel_cnt_le = 0;
            }
            __syncthreads();
            atomicAdd(&sel_cnt_le, cnt_le);
            atomicMin(&sel_lo, next);
            __syncthreads();
            if (tid == 0)
                sh_val[1] = __longlong_as_double((long long)(((unsigned long long)sel_cnt_le > rank_hi) ? v_lo : sel_lo));
        };
        const unsigned int br_m = br_n, br_b = br_below;
        const bool br_ok = have_br && br_m <= (unsigned int)BR_CAP && k_lo >= br_b && k_hi < (unsigned long long)br_b + br_m;
        if (br_ok)
        {
            // both middle order statistics are among the br_m listed keys: select inside LDS
            select_rank(br_list, (int)br_m, k_lo - br_b, false);
            const unsigned long long v_lo = sel_prefix;
            if (nsel == 2)
                upper_middle(br_list, (int)br_m, v_lo, k_hi - br_b);
            if (tid == 0)
                sh_val[0] = __longlong_as_double((long long)v_lo);
            __syncthreads();
        }
        else
        {
            select_rank(kd, n, k_lo, true);
            const unsigned long long v_lo = sel_prefix;
            if (nsel == 2)
                upper_middle(kd, n, v_lo, k_hi);
            if (tid == 0)
                sh_val[0] = __longlong_as_double((long long)v_lo);
            __syncthreads();
        }
        pf[br_ok ? 7 : 6] += now() - tw1;
        if (tid == 0)
            sh_sigma = 1.482602218505602 * (nsel == 1 ? sh_val[0] : (sh_val[0] + sh_val[1]) / 2.0);
        __syncthreads();
        sigma_change = (sigma > 0.0) ? fabs(sh_sigma - sigma) / sigma : 1.0;
        sigma = sh_sigma;
        // raw weights (even in r: only |r| is needed), their sum in a fixed order, then the normalised sqrt
        double wsum = 0.0;
        for (int i0 = tid; i0 < n; i0 += RU * T)
        {
            unsigned long long kk[RU];
#pragma unroll
            for (int u = 0; u < RU; ++u)
                kk[u] = kd[i0 + u * T < n ? i0 + u * T : n - 1];
#pragma unroll
            for (int u = 0; u < RU; ++u)
            {
                const int i = i0 + u * T;
                if (i >= n)
                    break;
                const double rs = __longlong_as_double((long long)kk[u]) / sigma;
                const double w = fmax(irls_psi(rs, a.loss) / rs, DBL_EPSILON);
                swd[i] = w;
                wsum += w; // same order of additions per thread as the one-row loop: rows tid, tid + T, ...
            }
        }
        wsum = wave_sum(wsum);
        if ((tid & 63) == 0)
            lds_red[tid >> 6] = wsum;
        __syncthreads();
        if (tid == 0)
        {
            double t = 0.0;
            for (int w = 0; w < NW; ++w)
                t += lds_red[w];
            sh_scale = (double)n / t;
            // test_delta_irls on wavefront 0's copy of the iterates
            int st = ST_CONTINUE;
            for (int k = 0; k < P; ++k)
            {
                const double xi = lds_state.x[k], dxi = fabs(workp[k] - xi);
                if (fmin(dxi / fabs(xi), dxi) < a.irls_xtol)
                    st = ST_SUCCESS;
                else
                {
                    st = ST_CONTINUE;
                    break;
                }
            }
            sh_flag = st;
        }
        __syncthreads();
        const double scale = sh_scale;
        irls_status = sh_flag;
        pf[3] += now() - tw0;
        if (irls_status == ST_SUCCESS || irls_iter >= a.irls_maxiter)
            break;
        for (int i0 = tid; i0 < n; i0 += RU * T)
        {
            double ww[RU], uu[RU];
#pragma unroll
            for (int u = 0; u < RU; ++u)
            {
                const int i = i0 + u * T < n ? i0 + u * T : n - 1;
                ww[u] = swd[i];
                uu[u] = ud ? ud[i] : 1.0;
            }
#pragma unroll
            for (int u = 0; u < RU; ++u)
            {
                const int i = i0 + u * T;
                if (i >= n)
                    break;
                double w = ww[u] * scale;
                if (ud)
                    w = (uu[u] * uu[u]) * w;
                swd[i] = sqrt(w);
            }
        }
#pragma unroll
        for (int k = 0; k < P; ++k)
            workp[k] = lds_state.x[k];
        __syncthreads();
    }

    if (tid == 0 && a.pass_total)
    {
        atomicAdd(a.pass_total, (unsigned long long)n_lm_pass);
        atomicAdd(a.pass_total + 1, (unsigned long long)n_rw);
    }
    if (tid == 0 && a.prof)
    {
        pf[4] = now() - t_begin;
        for (int k = 0; k < 8; ++k)
            a.prof[(size_t)d * 8 + k] = pf[k];
    }
    if (tid == 0)
    {
        int conv = status;
        double irls_tol = 0.0;
        if (!(status == ST_EBADFUNC || (status == ST_ENOPROG && irls_iter == 1)))
        {
            if (irls_iter >= a.irls_maxiter && irls_status != ST_SUCCESS)
            {
                irls_status = ST_EMAXITER;
                conv = ST_EMAXITER;
            }
        }
        const bool ok = (conv == ST_SUCCESS || conv == ST_EMAXITER);
        for (int k = 0; k < P; ++k)
        {
            a.par[(size_t)d * P + k] = ok ? lds_state.x[k] : a.start[k];
            irls_tol = fmax(irls_tol, fabs(workp[k] - lds_state.x[k]));
        }
        a.scal[(size_t)d * 4 + 0] = sigma;
        a.scal[(size_t)d * 4 + 1] = lds_state.chisq1;
        a.scal[(size_t)d * 4 + 2] = irls_tol;
        a.scal[(size_t)d * 4 + 3] = chisq_init;
        a.ints[(size_t)d * 4 + 0] = conv;
        a.ints[(size_t)d * 4 + 1] = irls_status;
        a.ints[(size_t)d * 4 + 2] = irls_iter;
        a.ints[(size_t)d * 4 + 3] = lds_state.niter;
    }
}

} // namespace gslnls
