// wide_core.hpp -- the Levenberg-Marquardt state machine for 10 <= p <= 64 parameters, run by ONE workgroup.
//
// Same algorithm, same order of operations as lm_advance<P>() in lm_core.hpp (which see for the reference's
// file:line of every step: trust_init_LD / trust_iterate_lu_LD / lm_step_LD / nielsen_* src/trust.c, driver2
// src/nls_fit.c:40-121, GSL scaling.c / cholesky.c (mcholesky) / convergence.c) -- but p is a run-time value,
// the state lives in LDS instead of registers, and the p x p algebra is spread over the lanes of a wavefront:
// the reference allocates its n x p workspace for any p (src/nls.c:266) and so does the formula front end
// (R/nls.R:588-599), while lm_core.hpp unrolls everything for p <= 9.
//
//   * ONE wavefront runs the whole step: lane k owns component k of every p-vector (p <= 64); the vectors and the packed
//     lower triangle of J^T J sit in LDS for the duration of the call, the scalars (mu, nu, delta, counters, phase) in
//     registers, identical in every lane; sums that lm_core.hpp takes sequentially (v^T J^T J v, ||D v||^2, ...) are
//     taken in the same index order, their terms travelling by ds_bpermute;
//   * modified Cholesky with diagonal pivoting (gsl_linalg_mcholesky, Gill-Murray-Wright) of J^T J + mu D^2: lane i owns
//     row i of the lower triangle in LDS (leading dimension p + 1: conflict-free column walks) and its diagonal entry
//     in a register, pivot search by a wavefront maximum + ballot, the rank-one update of column step j by all rows
//     at once, four elements per LDS round trip;
//   * triangular solves: lane i owns b_i, the pivot element travels by ds_bpermute.  The back substitution runs over
//     columns (j descending) where lm_solve<P> runs over rows (j ascending): same sums, different association --
//     the one place where the two state machines are not operation for operation the same.
#pragma once
#if !defined(__HIPCC_RTC__)
#include <hip/hip_runtime.h>
#endif
#include "lm_core.hpp"

namespace gslnls
{

constexpr int WP = 64;                 // capacity in parameters
constexpr int WNA = WP * (WP + 1) / 2; // packed lower triangle
constexpr int WT_ADV = 64;             // the advancing workgroup is one wavefront

struct WState
{
    double x[WP], xt[WP], dx[WP], vel[WP], acc[WP], g[WP], diag[WP], lo[WP], up[WP];
    double A[WNA]; // lower triangle of J^T J at x, packed row by row with the ACTUAL p: (i,j), j<=i -> i(i+1)/2+j
    double fnorm2, mu, nu, delta, avratio, chisq0, chisq1, chisq_init;
    int bad_steps, niter, phase, status, info, nevalf, nevaldf, nevalfvv;
    int p, end_launch;
};

struct WAdvanceArgs
{
    WState *state;
    const double *totals; // [2 + NA + p]: ssr, badj, packed J^T J, J^T f of the pass that state->phase asked for
    LmParams prm;
    double *ssrtrace, *partrace; // maxiter + 1, (maxiter + 1) x p column-major, or nullptr
    WState *host_mirror;         // pinned, mapped: the final state lands here
    unsigned int *done_seq;      // pinned word: sequence number of the last finished fit
    unsigned int seq;
    int launch_idx;
    int p; // (the same as state->p: the loads of a call are issued at once instead of after a first round trip for p)
    int pivoted; // 1: always the reference's pivoted, modified Cholesky (GSLNLS_WIDE_PIVOTED=1); 0: natural order first
};

#if defined(__HIPCC__) || defined(__HIPCC_RTC__)

// the lane of this thread, as a value the compiler cannot trace: kernels that run the step in a loop (wide_fit_kernel,
// wide_loop_kernel) otherwise have every lane-dependent LDS address of the loop body -- hundreds: tri(lane, k) for each
// literal k of each sum -- computed once before the loop, spilled, and fetched back from scratch memory one use at a time
// (an advance_pre of 11 us instead of 4).  Costs nothing where there is no loop.
__device__ __forceinline__ int wide_lane()
{
    int l = (int)(threadIdx.x & 63u);
    asm volatile("" : "+v"(l));
    return l;
}

__device__ __forceinline__ void wide_lds_sync()
{
    // LDS traffic between the lanes of ONE wavefront: make the stores visible before the loads that follow
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// value of lane `src_lane` (wave-uniform index) in every lane: two v_readlane_b32 (a few cycles; through
// ds_bpermute every broadcast was an LDS round trip on the critical path of the factorisation)
__device__ __forceinline__ double wide_bcast(double v, int src_lane)
{
    const long long bits = __double_as_longlong(v);
    const int sl = __builtin_amdgcn_readfirstlane(src_lane);
    const int lo = __builtin_amdgcn_readlane((int)(bits & 0xffffffffll), sl);
    const int hi = __builtin_amdgcn_readlane((int)(bits >> 32), sl);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// per-lane source index (ds_bpermute)
__device__ __forceinline__ double wide_shfl(double v, int src_lane)
{
    const long long bits = __double_as_longlong(v);
    const int lo = __shfl((int)(bits & 0xffffffffll), src_lane, 64), hi = __shfl((int)(bits >> 32), src_lane, 64);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

template <int CTRL>
__device__ __forceinline__ double wide_dpp(double v)
{
    const long long bits = __double_as_longlong(v);
    int lo = (int)(bits & 0xffffffffll), hi = (int)(bits >> 32);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// maximum over the 64 lanes, in every lane: DPP butterflies inside each row of 16 lanes, then the four row values by
// v_readlane.  fmax is exact and order independent: any reduction tree gives the same bits.
__device__ __forceinline__ double wide_wave_max(double v)
{
    v = fmax(v, wide_dpp<0xB1>(v));  // quad_perm [1,0,3,2]
    v = fmax(v, wide_dpp<0x4E>(v));  // quad_perm [2,3,0,1]
    v = fmax(v, wide_dpp<0x141>(v)); // row_half_mirror
    v = fmax(v, wide_dpp<0x140>(v)); // row_mirror
    return fmax(fmax(wide_bcast(v, 0), wide_bcast(v, 16)), fmax(wide_bcast(v, 32), wide_bcast(v, 48)));
}

// wavefront sum by xor butterflies: every lane performs the same tree (a + b and b + a are the same bits), so the
// result is identical in all 64 lanes and from run to run
__device__ __forceinline__ double wave_sum_wide(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
    {
        const long long bits = __double_as_longlong(v);
        const int lo = __shfl_xor((int)(bits & 0xffffffffll), m, 64), hi = __shfl_xor((int)(bits >> 32), m, 64);
        v += __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
    }
    return v;
}

// LDS working set of one call
struct WideLds
{
    double x[WP], xt[WP], dx[WP], vel[WP], acc[WP], g[WP], diag[WP], lo[WP], up[WP];
    double A[WNA];
    double M[WP * (WP + 1)]; // full symmetric p x p, leading dimension p + 1
    double rhs[WP], sol[WP], row[WP];
};

// sum of v_0 + v_1 + ... + v_{p-1} in index order (the order of the sequential loops of lm_core.hpp), every lane gets it;
// the p values travel by ds_bpermute, all requests in flight together, instead of p dependent LDS round trips
__device__ __forceinline__ double wide_seq_sum(double v, int p)
{
    double s = 0.0;
    for (int i = 0; i < p; ++i)
        s += wide_bcast(v, i);
    return s;
}

// m += (lane K of this lane's row of 16 lanes of vb) * t: the 64-bit DPP form of v_fmac_f64 (row_newbcast is the one DPP
// control the double-precision ALU takes), one instruction where v_readlane x 2 + v_fma_f64 were three.  All 64 lanes
// must be active.
template <int V>
struct WideInt
{
    static constexpr int value = V;
};
template <int K>
__device__ __forceinline__ void wide_fmac_rowbcast(double &m, double vb, double t)
{
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(m) : "v"(vb), "v"(t), "n"(K));
}

// maximum over the first 16 R lanes (the others hold values that cannot win), in every lane
template <int R>
__device__ __forceinline__ double wide_wave_max_rows(double v)
{
    v = fmax(v, wide_dpp<0xB1>(v));  // quad_perm [1,0,3,2]
    v = fmax(v, wide_dpp<0x4E>(v));  // quad_perm [2,3,0,1]
    v = fmax(v, wide_dpp<0x141>(v)); // row_half_mirror
    v = fmax(v, wide_dpp<0x140>(v)); // row_mirror
    double r = wide_bcast(v, 0);
    if constexpr (R > 1)
        r = fmax(r, wide_bcast(v, 16));
    if constexpr (R > 2)
        r = fmax(r, wide_bcast(v, 32));
    if constexpr (R > 3)
        r = fmax(r, wide_bcast(v, 48));
    return r;
}

// m[q] for a wave-uniform q < PW: a computed jump into a table of `v_mov_b64 v, m[k]; s_branch end` pairs, eight bytes
// each (a run-time register index would send the whole row to scratch memory, and the compiler's branch tree for a
// switch costs more than the whole rank-one update at PW = 16)
template <int PW>
__device__ __forceinline__ double wide_pick(const double (&m)[PW], int q)
{
    double v;
#define GSLNLS_WPICK_E(k) "v_mov_b64 %0, %" #k "\n s_branch 1f\n"
    asm("s_getpc_b64 s[96:97]\n"
        "s_lshl_b32 s98, %1, 3\n"
        "s_add_u32 s98, s98, 20\n" // from the instruction after s_getpc to the first table entry
        "s_add_u32 s96, s96, s98\n"
        "s_addc_u32 s97, s97, 0\n"
        "s_setpc_b64 s[96:97]\n"
        GSLNLS_WPICK_E(2) GSLNLS_WPICK_E(3) GSLNLS_WPICK_E(4) GSLNLS_WPICK_E(5) GSLNLS_WPICK_E(6) GSLNLS_WPICK_E(7) GSLNLS_WPICK_E(8) GSLNLS_WPICK_E(9)
        GSLNLS_WPICK_E(10) GSLNLS_WPICK_E(11) GSLNLS_WPICK_E(12) GSLNLS_WPICK_E(13) GSLNLS_WPICK_E(14) GSLNLS_WPICK_E(15) GSLNLS_WPICK_E(16) GSLNLS_WPICK_E(17)
        GSLNLS_WPICK_E(18) GSLNLS_WPICK_E(19) GSLNLS_WPICK_E(20) GSLNLS_WPICK_E(21) GSLNLS_WPICK_E(22) GSLNLS_WPICK_E(23) GSLNLS_WPICK_E(24) GSLNLS_WPICK_E(25)
        GSLNLS_WPICK_E(26) GSLNLS_WPICK_E(27) GSLNLS_WPICK_E(28) GSLNLS_WPICK_E(29) GSLNLS_WPICK_E(30) GSLNLS_WPICK_E(31) GSLNLS_WPICK_E(32) GSLNLS_WPICK_E(33)
        GSLNLS_WPICK_E(34) GSLNLS_WPICK_E(35) GSLNLS_WPICK_E(36) GSLNLS_WPICK_E(37) GSLNLS_WPICK_E(38) GSLNLS_WPICK_E(39) GSLNLS_WPICK_E(40) GSLNLS_WPICK_E(41)
        GSLNLS_WPICK_E(42) GSLNLS_WPICK_E(43) GSLNLS_WPICK_E(44) GSLNLS_WPICK_E(45) GSLNLS_WPICK_E(46) GSLNLS_WPICK_E(47) GSLNLS_WPICK_E(48) GSLNLS_WPICK_E(49)
        GSLNLS_WPICK_E(50) GSLNLS_WPICK_E(51) GSLNLS_WPICK_E(52) GSLNLS_WPICK_E(53) GSLNLS_WPICK_E(54) GSLNLS_WPICK_E(55) GSLNLS_WPICK_E(56) GSLNLS_WPICK_E(57)
        GSLNLS_WPICK_E(58) GSLNLS_WPICK_E(59) GSLNLS_WPICK_E(60) GSLNLS_WPICK_E(61) GSLNLS_WPICK_E(62) GSLNLS_WPICK_E(63) GSLNLS_WPICK_E(64) GSLNLS_WPICK_E(65)
        "1:\n"
        : "=&v"(v)
        : "s"(q), "v"(m[0 < PW ? 0 : 0]), "v"(m[1 < PW ? 1 : 0]), "v"(m[2 < PW ? 2 : 0]), "v"(m[3 < PW ? 3 : 0]),
          "v"(m[4 < PW ? 4 : 0]), "v"(m[5 < PW ? 5 : 0]), "v"(m[6 < PW ? 6 : 0]), "v"(m[7 < PW ? 7 : 0]),
          "v"(m[8 < PW ? 8 : 0]), "v"(m[9 < PW ? 9 : 0]), "v"(m[10 < PW ? 10 : 0]), "v"(m[11 < PW ? 11 : 0]),
          "v"(m[12 < PW ? 12 : 0]), "v"(m[13 < PW ? 13 : 0]), "v"(m[14 < PW ? 14 : 0]), "v"(m[15 < PW ? 15 : 0]),
          "v"(m[16 < PW ? 16 : 0]), "v"(m[17 < PW ? 17 : 0]), "v"(m[18 < PW ? 18 : 0]), "v"(m[19 < PW ? 19 : 0]),
          "v"(m[20 < PW ? 20 : 0]), "v"(m[21 < PW ? 21 : 0]), "v"(m[22 < PW ? 22 : 0]), "v"(m[23 < PW ? 23 : 0]),
          "v"(m[24 < PW ? 24 : 0]), "v"(m[25 < PW ? 25 : 0]), "v"(m[26 < PW ? 26 : 0]), "v"(m[27 < PW ? 27 : 0]),
          "v"(m[28 < PW ? 28 : 0]), "v"(m[29 < PW ? 29 : 0]), "v"(m[30 < PW ? 30 : 0]), "v"(m[31 < PW ? 31 : 0]),
          "v"(m[32 < PW ? 32 : 0]), "v"(m[33 < PW ? 33 : 0]), "v"(m[34 < PW ? 34 : 0]), "v"(m[35 < PW ? 35 : 0]),
          "v"(m[36 < PW ? 36 : 0]), "v"(m[37 < PW ? 37 : 0]), "v"(m[38 < PW ? 38 : 0]), "v"(m[39 < PW ? 39 : 0]),
          "v"(m[40 < PW ? 40 : 0]), "v"(m[41 < PW ? 41 : 0]), "v"(m[42 < PW ? 42 : 0]), "v"(m[43 < PW ? 43 : 0]),
          "v"(m[44 < PW ? 44 : 0]), "v"(m[45 < PW ? 45 : 0]), "v"(m[46 < PW ? 46 : 0]), "v"(m[47 < PW ? 47 : 0]),
          "v"(m[48 < PW ? 48 : 0]), "v"(m[49 < PW ? 49 : 0]), "v"(m[50 < PW ? 50 : 0]), "v"(m[51 < PW ? 51 : 0]),
          "v"(m[52 < PW ? 52 : 0]), "v"(m[53 < PW ? 53 : 0]), "v"(m[54 < PW ? 54 : 0]), "v"(m[55 < PW ? 55 : 0]),
          "v"(m[56 < PW ? 56 : 0]), "v"(m[57 < PW ? 57 : 0]), "v"(m[58 < PW ? 58 : 0]), "v"(m[59 < PW ? 59 : 0]),
          "v"(m[60 < PW ? 60 : 0]), "v"(m[61 < PW ? 61 : 0]), "v"(m[62 < PW ? 62 : 0]), "v"(m[63 < PW ? 63 : 0])
        : "s96", "s97", "s98", "scc");
#undef GSLNLS_WPICK_E
    return v;
}

// v_max_u32 with the DPP exchange folded into the instruction
template <int CTRL>
__device__ __forceinline__ unsigned int wide_umax_dpp(unsigned int x)
{
    const unsigned int y = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, true);
    return x > y ? x : y;
}

// (A + mu D^2) sol = rhs by ONE wavefront (all 64 lanes of it must call; lanes >= p idle along), p <= PW.
//
// gsl_linalg_mcholesky (Gill-Murray-Wright modified Cholesky with diagonal pivoting) with the matrix in REGISTERS: lane i
// holds row i of the symmetric matrix, m[k] = S[i][k], by ORIGINAL index -- nothing is ever interchanged.  The
// reference's permutation lives in `pos` (lane i: the position row i has in the permuted order -- rows at positions
// > j are the ones still to be eliminated; the tie rule of the pivot search, the first position wins, reads it), the
// pivot sequence in `ord` (lane j: the row eliminated at step j).
// One step: pivot row q, its column v_i = S[i][q] is each lane's own m[q], alpha = max(eps, |d_qq|, theta^2 / beta) with
// theta = max |v_i|, then the rank-one update S[i][k] -= (v_i / alpha) v_k of ALL columns in PW instructions (v_fmac_f64
// with a DPP row broadcast of v_k) -- columns and rows already eliminated see v = 0.  The forward substitution rides
// along (b_i -= l_i b_q); the multipliers l_i = v_i / alpha also go to LDS, T[q][i], so that the back substitution
// L^T w = z finds row q_j of L^T as T[i][q_j]: lane i's own row, conflict-free.
// A lone wavefront issues one instruction every ~5 clocks, ~10 when it depends on the one before (scripts/lane_probe),
// so the step is built to be SHORT rather than wide:
//   * the pivot search reduces the HIGH words of |d_ii| (sign, exponent, 20 mantissa bits; v_max_u32 with the DPP
//     exchange folded in); one lane holding the largest high word is the common case and ends the search, several go
//     to the full comparison (exact ties: the first position wins);
//   * theta is only needed when it raises alpha: (|v_i| / sqrt(beta))^2 > max(eps, |d_qq|) in ANY lane is the same
//     condition as for the maximum (rounding is monotone), one ballot instead of a reduction;
//   * 1 / alpha is the compiler's own correctly rounded sequence, spelled out so that the permutation bookkeeping issues
//     into its dependent chain, and the rank-one update of step j issues into the pivot search of step j + 1 (which
//     only needs the updated diagonal).
// Against lm_solve<P> / the reference: the same sums in the same order with one exception -- an entry S[i][k] whose rows
// were interchanged relative to each other is updated as (v_i / alpha) v_k here where the reference's single (lower)
// copy gets (v_k / alpha) v_i: one rounding apart.
template <int PW>
__device__ __forceinline__ void wide_solve_reg(WideLds &L, int p, double mu, const double *rhs, double *sol, int lane)
{
    constexpr int R = PW / 16;
    constexpr int LD = WP + 1;
    constexpr int NCH = PW / 4; // rank-one update: chunks of four columns
    constexpr int NSLOT = 8;    // gaps of the pivot search they are issued into
    static_assert(NCH <= 2 * NSLOT, "every chunk of the rank-one update needs a slot");
    double *T = L.M;
    const bool mine = lane < p;
    double m[PW];
    {
        const int base = lane * (lane + 1) / 2;
#pragma unroll
        for (int k = 0; k < PW; ++k)
        {
            const int idx = k < lane ? base + k : k * (k + 1) / 2 + lane;
            const double a = L.A[(mine && k < p) ? idx : 0];
            m[k] = (mine && k < p && k != lane) ? a : 0.0; // (the diagonal lives in dg)
        }
    }
    double dg = 0.0, b = 0.0;
    if (mine)
    {
        dg = L.A[tri(lane, lane)] + mu * L.diag[lane] * L.diag[lane];
        b = rhs[lane];
    }
    double xm = 0.0;
    {
        // (four interleaved chains: PW dependent maxima in a row are PW x 10 clocks)
        double x4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < PW; ++k)
            x4[k & 3] = fmax(x4[k & 3], fabs(m[k]));
        xm = fmax(fmax(x4[0], x4[1]), fmax(x4[2], x4[3]));
    }
    const double gamma = wide_wave_max_rows<R>(fabs(dg)), xi = wide_wave_max_rows<R>(xm);
    double beta;
    if (p == 1)
        beta = fmax(fmax(gamma, xi), DBL_EPSILON);
    else
        beta = fmax(fmax(gamma, xi / sqrt((double)p * p - 1.0)), DBL_EPSILON);
    const double betainv = 1.0 / sqrt(beta);
    double dinv = 0.0; // 1 / alpha of the step that eliminated this row
    int pos = mine ? lane : -1, ord = 0;

    // pivot of step jn among the rows `live` (positions >= jn): the first position holding the largest |diagonal|
    // (`if (d > maxd)` of the sequential scan from position jn: the first element wins ties, NaNs never win -- unless
    // one sits at position jn, where the scan starts: then nothing compares greater and the pivot stays there).
    // dq = |diagonal| of the pivot.
    auto pivot = [&](int jn, bool live, int &q, double &dq, auto &&fill) {
        double d;
        asm("v_max_f64 %0, |%1|, 0" : "=v"(d) : "v"(dg)); // |d_ii|, NaN -> 0
        d = live ? d : 0.0;
        const unsigned int hi = (unsigned int)((unsigned long long)__double_as_longlong(d) >> 32);
        fill(0);
        unsigned int x = wide_umax_dpp<0xB1>(hi); // quad_perm [1,0,3,2]
        fill(1);
        x = wide_umax_dpp<0x4E>(x); // quad_perm [2,3,0,1]
        fill(2);
        x = wide_umax_dpp<0x141>(x); // row_half_mirror
        fill(3);
        x = wide_umax_dpp<0x140>(x); // row_mirror
        fill(4);
        unsigned int mh = (unsigned int)__builtin_amdgcn_readlane((int)x, 0);
        if constexpr (R > 1)
        {
            const unsigned int t1 = (unsigned int)__builtin_amdgcn_readlane((int)x, 16);
            mh = mh > t1 ? mh : t1;
        }
        if constexpr (R > 2)
        {
            const unsigned int t2 = (unsigned int)__builtin_amdgcn_readlane((int)x, 32);
            mh = mh > t2 ? mh : t2;
        }
        if constexpr (R > 3)
        {
            const unsigned int t3 = (unsigned int)__builtin_amdgcn_readlane((int)x, 48);
            mh = mh > t3 ? mh : t3;
        }
        fill(5);
        unsigned long long hit = __builtin_amdgcn_ballot_w64(live && hi == mh);
        const unsigned long long nanr = __builtin_amdgcn_ballot_w64(live && pos == jn && dg != dg);
        fill(6);
        q = hit ? (int)__builtin_ctzll(hit) : 0;
        fill(7);
        if (hit & (hit - 1))
        {
            // several rows share the largest high word: the full comparison, then the first position among the equal
            const double maxd = wide_wave_max_rows<R>((hit >> lane) & 1 ? d : 0.0);
            hit = __builtin_amdgcn_ballot_w64(live && d == maxd);
            int best = 1 << 30;
            while (hit)
            {
                const int c = (int)__builtin_ctzll(hit);
                hit &= hit - 1;
                const int pc = __builtin_amdgcn_readlane(pos, c);
                if (pc < best)
                {
                    best = pc;
                    q = c;
                }
            }
        }
        q = __builtin_amdgcn_readfirstlane(q);
        dq = wide_bcast(d, q);
        if (nanr)
        {
            q = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(nanr));
            dq = __longlong_as_double(0x7ff8000000000000ll);
        }
    };

    int q = 0;
    double dq = 0.0;
    pivot(0, mine, q, dq, [](int) {});
    for (int j = 0; j < p; ++j)
    {
        // alpha = max(eps, |d_qq|, theta^2 / beta), theta = max |v_i|: 1 / max(eps, |d_qq|) is started at once
        const double a0 = fmax(DBL_EPSILON, dq);
        bool dv_f = false, dv_g = false;
        const double dv_s = __builtin_amdgcn_div_scale(1.0, a0, false, &dv_g);
        double dv_r = __builtin_amdgcn_rcp(dv_s);
        double v = wide_pick<PW>(m, q);
        __builtin_amdgcn_sched_barrier(0);
        dv_r = __builtin_fma(dv_r, __builtin_fma(-dv_s, dv_r, 1.0), dv_r);
        // rows at position j and q trade positions (nothing moves); afterwards the rows still to be eliminated are the
        // ones at positions > j
        const int pq = __builtin_amdgcn_readlane(pos, q);
        pos = pos == j ? pq : pos;
        pos = lane == q ? j : pos;
        const bool upd = pos > j;
        v = upd ? v : 0.0;
        __builtin_amdgcn_sched_barrier(0);
        dv_r = __builtin_fma(dv_r, __builtin_fma(-dv_s, dv_r, 1.0), dv_r);
        const double dv_n = __builtin_amdgcn_div_scale(1.0, a0, true, &dv_f);
        const double w = fabs(v) * betainv;
        const unsigned long long raise = __builtin_amdgcn_ballot_w64(w * w > a0);
        ord = lane == j ? q : ord;
        __builtin_amdgcn_sched_barrier(0);
        double dv_q = dv_n * dv_r;
        dv_q = __builtin_amdgcn_div_fmas(__builtin_fma(-dv_s, dv_q, dv_n), dv_r, dv_q, dv_f);
        const double bq = wide_bcast(b, q);
        __builtin_amdgcn_sched_barrier(0);
        double ainv = __builtin_amdgcn_div_fixup(dv_q, a0, 1.0);
        if (raise)
        {
            const double theta = wide_wave_max_rows<R>(fmax(fabs(v), 0.0));
            const double u = theta * betainv;
            ainv = 1.0 / (u * u);
            asm volatile("" : "+v"(ainv)); // (a branch, not a select in front of a second division)
        }
        const double t = ainv * v; // the multiplier l_i (0 in rows that take no part)
        dg -= t * v;
        b = upd ? b - t * bq : b;
        dinv = lane == q ? ainv : dinv;
        T[q * LD + lane] = t;
        const double tn = -t;
        // v of row 16 g + l % 16 in lane l, for the DPP row broadcasts
        double vb[R];
        vb[0] = v;
        if constexpr (R > 1)
        {
            const long long bits = __double_as_longlong(v);
            const unsigned int lo = (unsigned int)(bits & 0xffffffffll), hi = (unsigned int)(bits >> 32);
            const auto l16 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
            const auto h16 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
            // l16[0] = rows [0, 0, 2, 2], l16[1] = rows [1, 1, 3, 3]
            if constexpr (R == 2)
            {
                vb[0] = __longlong_as_double(((long long)h16[0] << 32) | l16[0]);
                vb[1] = __longlong_as_double(((long long)h16[1] << 32) | l16[1]);
            }
            else
            {
                const auto l0 = __builtin_amdgcn_permlane32_swap(l16[0], l16[0], false, false);
                const auto h0 = __builtin_amdgcn_permlane32_swap(h16[0], h16[0], false, false);
                const auto l1 = __builtin_amdgcn_permlane32_swap(l16[1], l16[1], false, false);
                const auto h1 = __builtin_amdgcn_permlane32_swap(h16[1], h16[1], false, false);
                vb[0] = __longlong_as_double(((long long)h0[0] << 32) | l0[0]); // rows [0, 0, 0, 0]
                vb[2] = __longlong_as_double(((long long)h0[1] << 32) | l0[1]); // rows [2, 2, 2, 2]
                vb[1] = __longlong_as_double(((long long)h1[0] << 32) | l1[0]);
                if constexpr (R > 3)
                    vb[3] = __longlong_as_double(((long long)h1[1] << 32) | l1[1]);
            }
        }
        // columns 4 c .. 4 c + 3
        auto chunk = [&](auto cc) {
            constexpr int c = decltype(cc)::value;
            if constexpr (c < NCH)
            {
                constexpr int g = (4 * c) / 16, k0 = (4 * c) % 16;
                wide_fmac_rowbcast<k0 + 0>(m[4 * c + 0], vb[g], tn);
                wide_fmac_rowbcast<k0 + 1>(m[4 * c + 1], vb[g], tn);
                wide_fmac_rowbcast<k0 + 2>(m[4 * c + 2], vb[g], tn);
                wide_fmac_rowbcast<k0 + 3>(m[4 * c + 3], vb[g], tn);
            }
        };
        auto fill = [&](int slot) {
            // (slot is a literal at every call site: the switch folds away)
            switch (slot)
            {
            case 0: chunk(WideInt<0>{}); chunk(WideInt<NSLOT>{}); break;
            case 1: chunk(WideInt<1>{}); chunk(WideInt<NSLOT + 1>{}); break;
            case 2: chunk(WideInt<2>{}); chunk(WideInt<NSLOT + 2>{}); break;
            case 3: chunk(WideInt<3>{}); chunk(WideInt<NSLOT + 3>{}); break;
            case 4: chunk(WideInt<4>{}); chunk(WideInt<NSLOT + 4>{}); break;
            case 5: chunk(WideInt<5>{}); chunk(WideInt<NSLOT + 5>{}); break;
            case 6: chunk(WideInt<6>{}); chunk(WideInt<NSLOT + 6>{}); break;
            case 7: chunk(WideInt<7>{}); chunk(WideInt<NSLOT + 7>{}); break;
            default: break;
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        // (a DPP operand written by the instruction before: 2 wait states)
        if constexpr (R == 1)
            asm volatile("s_nop 1" : "+v"(vb[0]));
        else if constexpr (R == 2)
            asm volatile("s_nop 1" : "+v"(vb[0]), "+v"(vb[1]));
        else if constexpr (R == 3)
            asm volatile("s_nop 1" : "+v"(vb[0]), "+v"(vb[1]), "+v"(vb[2]));
        else
            asm volatile("s_nop 1" : "+v"(vb[0]), "+v"(vb[1]), "+v"(vb[2]), "+v"(vb[3]));
        if (j + 1 < p) // (nothing reads the matrix after the last step)
            pivot(j + 1, upd, q, dq, fill);
    }
    b *= dinv;
    wide_lds_sync();
    // L^T w = z: column sweep from the last pivot (j descending per element; lm_solve<P> adds them ascending).  `pos` is
    // now the step that eliminated the row; the element of L^T that row i needs from pivot row q_j is T[i][q_j].
    {
        const double *Trow = T + lane * LD;
        int j = p - 1;
        for (; j >= 4; j -= 4)
        {
            const int q0 = __builtin_amdgcn_readlane(ord, j), q1 = __builtin_amdgcn_readlane(ord, j - 1),
                      q2 = __builtin_amdgcn_readlane(ord, j - 2), q3 = __builtin_amdgcn_readlane(ord, j - 3);
            const double t0 = Trow[q0], t1 = Trow[q1], t2 = Trow[q2], t3 = Trow[q3];
            double w = wide_bcast(b, q0);
            if (mine && pos < j)
                b -= t0 * w;
            w = wide_bcast(b, q1);
            if (mine && pos < j - 1)
                b -= t1 * w;
            w = wide_bcast(b, q2);
            if (mine && pos < j - 2)
                b -= t2 * w;
            w = wide_bcast(b, q3);
            if (mine && pos < j - 3)
                b -= t3 * w;
        }
        for (; j >= 1; --j)
        {
            const int qj = __builtin_amdgcn_readlane(ord, j);
            const double w = wide_bcast(b, qj);
            if (mine && pos < j)
                b -= Trow[qj] * w;
        }
    }
    if (mine)
        sol[lane] = b;
    wide_lds_sync();
}

// v of row 16 g + l % 16 in lane l, g = 0 .. R - 1: the operands of the DPP row broadcasts of a rank-one update
template <int R>
__device__ __forceinline__ void wide_row_copies(double v, double (&vb)[R])
{
    vb[0] = v;
    if constexpr (R > 1)
    {
        const long long bits = __double_as_longlong(v);
        const unsigned int lo = (unsigned int)(bits & 0xffffffffll), hi = (unsigned int)(bits >> 32);
        const auto l16 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto h16 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        // l16[0] = rows [0, 0, 2, 2], l16[1] = rows [1, 1, 3, 3]
        if constexpr (R == 2)
        {
            vb[0] = __longlong_as_double(((long long)h16[0] << 32) | l16[0]);
            vb[1] = __longlong_as_double(((long long)h16[1] << 32) | l16[1]);
        }
        else
        {
            const auto l0 = __builtin_amdgcn_permlane32_swap(l16[0], l16[0], false, false);
            const auto h0 = __builtin_amdgcn_permlane32_swap(h16[0], h16[0], false, false);
            const auto l1 = __builtin_amdgcn_permlane32_swap(l16[1], l16[1], false, false);
            const auto h1 = __builtin_amdgcn_permlane32_swap(h16[1], h16[1], false, false);
            vb[0] = __longlong_as_double(((long long)h0[0] << 32) | l0[0]); // rows [0, 0, 0, 0]
            vb[2] = __longlong_as_double(((long long)h0[1] << 32) | l0[1]); // rows [2, 2, 2, 2]
            vb[1] = __longlong_as_double(((long long)h1[0] << 32) | l1[0]);
            if constexpr (R > 3)
                vb[3] = __longlong_as_double(((long long)h1[1] << 32) | l1[1]);
        }
    }
    // (a DPP operand written by the instruction before: 2 wait states)
    if constexpr (R == 1)
        asm volatile("s_nop 1" : "+v"(vb[0]));
    else if constexpr (R == 2)
        asm volatile("s_nop 1" : "+v"(vb[0]), "+v"(vb[1]));
    else if constexpr (R == 3)
        asm volatile("s_nop 1" : "+v"(vb[0]), "+v"(vb[1]), "+v"(vb[2]));
    else
        asm volatile("s_nop 1" : "+v"(vb[0]), "+v"(vb[1]), "+v"(vb[2]), "+v"(vb[3]));
}

// columns K .. KEND - 1 of the rank-one update (a piece of it: the caller interleaves the pieces with other work)
template <int K, int KEND, int PW, int R>
__device__ __forceinline__ void wide_chol_update_range(double (&m)[PW], const double (&vb)[R], double tn)
{
    if constexpr (K < KEND)
    {
        wide_fmac_rowbcast<(K & 15)>(m[K], vb[K >> 4], tn);
        wide_chol_update_range<K + 1, KEND, PW, R>(m, vb, tn);
    }
}

// columns K .. PW - 1 of the rank-one update S -= t v^T (K a literal: the DPP control and the register are immediates)
template <int K, int PW, int R>
__device__ __forceinline__ void wide_chol_update(double (&m)[PW], const double (&vb)[R], double tn)
{
    if constexpr (K < PW)
    {
        wide_fmac_rowbcast<(K & 15)>(m[K], vb[K >> 4], tn);
        wide_chol_update<K + 1, PW, R>(m, vb, tn);
    }
}

// The same system WITHOUT the pivot search, for the matrices an LM step actually meets: J^T J + mu D^2 with mu > 0 is
// positive definite, and on a numerically positive definite matrix gsl_linalg_mcholesky never modifies anything (with the
// largest diagonal as pivot, theta^2 / beta <= d_jj always) -- its pivoting only decides the ORDER in which the same L D L^T
// is rounded.  Taking the rows in their natural order makes every register index a literal: no search (25 instructions
// of the ~145 a pivoted step issues), no computed jump for the pivot column, no permutation to keep, and the rank-one
// update of step j touches only the columns behind j -- ~42 instructions per step instead of ~145 on the lone wavefront
// that is the critical path of every trial step.  Same arithmetic per element as wide_solve_reg (t = v / alpha as
// (1 / alpha) v, S -= t v^T, the forward substitution riding along, D^-1, then L^T by columns): when the largest diagonal
// happens to come first at every step the two give the same bits.
// Returns false -- nothing written -- as soon as a pivot is not safely positive (d_jj <= 1e-12 of the entry it started
// as, d_jj <= DBL_EPSILON -- where the reference's alpha = max(eps, |d|, ..) would replace it --, or NaN): the matrix is not numerically positive definite, the caller runs the modified, pivoted factorisation.
template <int PW>
__device__ __forceinline__ bool wide_chol_reg(WideLds &L, int p, double mu, const double *rhs, double *sol, int lane)
{
    constexpr int R = PW / 16;
    constexpr int LD = WP + 1;
    double *T = L.M;
    const bool mine = lane < p;
    double m[PW];
    {
        const int base = lane * (lane + 1) / 2;
#pragma unroll
        for (int k = 0; k < PW; ++k)
        {
            const int idx = k < lane ? base + k : k * (k + 1) / 2 + lane;
            const double a = L.A[(mine && k < p) ? idx : 0];
            m[k] = (mine && k < p && k != lane) ? a : 0.0; // (the diagonal lives in dg)
        }
    }
    double dg = 0.0, b = 0.0;
    if (mine)
    {
        dg = L.A[tri(lane, lane)] + mu * L.diag[lane] * L.diag[lane];
        b = rhs[lane];
    }
    // (DBL_EPSILON: below it gsl_linalg_mcholesky replaces the pivot, alpha = max(eps, |d|, theta^2 / beta) -- the pivoted routine's case)
    const double thr = fmax(1e-12 * dg, DBL_EPSILON);
    double dinv = 0.0;
    // step j with j a LITERAL (a recursion over WideInt<j>, not a loop the compiler may or may not unroll: with a run-time
    // j the rows m[] are indexed dynamically and live in scratch memory)
    auto step = [&](auto self, auto jj) __attribute__((always_inline)) -> bool {
        constexpr int j = decltype(jj)::value;
        if constexpr (j >= PW)
            return true;
        else
        {
            if (j >= p)
                return true;
            const double dj = wide_bcast(dg, j), tj = wide_bcast(thr, j);
            if (!(dj > tj))
                return false;
            const double ainv = 1.0 / dj;
            const bool below = lane > j;
            const double v = below ? m[j] : 0.0; // S[i][j] of the rows still to be eliminated
            const double t = ainv * v;           // the multiplier l_i
            dg -= t * v;
            const double bj = wide_bcast(b, j);
            b = below ? b - t * bj : b;
            dinv = lane == j ? ainv : dinv;
            T[j * LD + lane] = t;
            if constexpr (j + 1 < PW)
            {
                double vb[R];
                wide_row_copies<R>(v, vb);
                wide_chol_update<j + 1, PW, R>(m, vb, -t);
            }
            return self(self, WideInt<j + 1>{});
        }
    };
    const bool ok = step(step, WideInt<0>{});
    if (!ok)
        return false;
    b *= dinv;
    wide_lds_sync();
    // L^T w = z by columns, the last one first: row i needs L[j][i] = the multiplier lane j stored at step i, T[i][j]
    {
        const double *Trow = T + lane * LD;
        int j = p - 1;
        for (; j >= 4; j -= 4)
        {
            const double t0 = Trow[j], t1 = Trow[j - 1], t2 = Trow[j - 2], t3 = Trow[j - 3];
            double w = wide_bcast(b, j);
            if (lane < j)
                b -= t0 * w;
            w = wide_bcast(b, j - 1);
            if (lane < j - 1)
                b -= t1 * w;
            w = wide_bcast(b, j - 2);
            if (lane < j - 2)
                b -= t2 * w;
            w = wide_bcast(b, j - 3);
            if (lane < j - 3)
                b -= t3 * w;
        }
        for (; j >= 1; --j)
        {
            const double w = wide_bcast(b, j);
            if (lane < j)
                b -= Trow[j] * w;
        }
    }
    if (mine)
        sol[lane] = b;
    wide_lds_sync();
    return true;
}

// det(A) of the packed lower triangle in L.A by the same natural-order elimination (no damping, no right-hand side):
// the product of the pivots d_0 d_1 ... -- what det_cholesky_jtj returns as (prod L_ii)^2 (src/nls_utils.c:55-73) -- or 0
// as soon as a pivot is not positive (the reference: gsl_linalg_cholesky_decomp1 fails -> 0).  All 64 lanes call.
template <int PW>
__device__ __forceinline__ double wide_det_reg(const double *Ap, int p, int lane)
{
    constexpr int R = PW / 16;
    const bool mine = lane < p;
    double m[PW];
    {
        const int base = lane * (lane + 1) / 2;
#pragma unroll
        for (int k = 0; k < PW; ++k)
        {
            const int idx = k < lane ? base + k : k * (k + 1) / 2 + lane;
            const double a = Ap[(mine && k < p) ? idx : 0];
            m[k] = (mine && k < p && k != lane) ? a : 0.0;
        }
    }
    double dg = mine ? Ap[tri(lane, lane)] : 0.0, det = 1.0;
    auto step = [&](auto self, auto jj) __attribute__((always_inline)) -> bool {
        constexpr int j = decltype(jj)::value;
        if constexpr (j >= PW)
            return true;
        else
        {
            if (j >= p)
                return true;
            const double dj = wide_bcast(dg, j);
            if (!(dj > 0.0))
                return false;
            det *= dj;
            const double ainv = 1.0 / dj;
            const double v = lane > j ? m[j] : 0.0;
            const double t = ainv * v;
            dg -= t * v;
            if constexpr (j + 1 < PW)
            {
                double vb[R];
                wide_row_copies<R>(v, vb);
                wide_chol_update<j + 1, PW, R>(m, vb, -t);
            }
            return self(self, WideInt<j + 1>{});
        }
    };
    return step(step, WideInt<0>{}) ? det : 0.0;
}

// the damped solve of one LM step: natural-order L D L^T first, gsl_linalg_mcholesky (modified, pivoted) when the matrix
// is not numerically positive definite -- or always, when `pivoted` asks for the reference's order of operations
template <int PW>
__device__ __forceinline__ void wide_solve_pw(WideLds &L, int p, double mu, const double *rhs, double *sol, int lane, int pivoted)
{
    if (!pivoted && wide_chol_reg<PW>(L, p, mu, rhs, sol, lane))
        return;
    wide_solve_reg<PW>(L, p, mu, rhs, sol, lane);
}

__device__ __forceinline__ void wide_solve(WideLds &L, int p, double mu, const double *rhs, double *sol, int lane, int pivoted = 0)
{
    // (noinline-sized bodies, one per register-file width; the caller has ONE call site)
    if (p <= 16)
        wide_solve_pw<16>(L, p, mu, rhs, sol, lane, pivoted);
    else if (p <= 32)
        wide_solve_pw<32>(L, p, mu, rhs, sol, lane, pivoted);
    else if (p <= 48)
        wide_solve_pw<48>(L, p, mu, rhs, sol, lane, pivoted);
    else
        wide_solve_pw<64>(L, p, mu, rhs, sol, lane, pivoted);
}

// the scalars of the state while a call works on them (registers, identical in every lane) + what the call decided
struct WideCtx
{
    double fnorm2, mu, nu, delta, avratio, chisq0, chisq1, chisq_init;
    int bad_steps, niter, phase, status, info, nevalf, nevaldf, nevalfvv;
    int niter_before, phase_before;
    int want;      // the damped solve this call needs: 0 none, 1 acceleration (rhs = -J^T fvv), 2 velocity (rhs = -g)
    bool rejected; // the velocity solve follows a rejected trial: same J^T J, D, g as before the pass, mu <- mu nu
    bool active;   // false: a launch enqueued past the end of the fit (nothing to do, nothing written)
};

// ONE wavefront (64 lanes >= p: lane k owns component k of every p-vector); all 64 lanes call.  Scalars of the state live
// in registers, identical in every lane; LDS traffic between lanes is ordered by wide_lds_sync().
// A call is three pieces: wide_advance_pre (state -> LDS, the decision of trust_iterate_lu_LD, right-hand side of the
// damped system in L.rhs), the solve (wide_solve / wide_solve_reg<PW>: into L.acc when c.want == 1, L.vel when 2), and
// wide_advance_post (trial point, state -> memory).  Kernels that carry a second user of the solve (the speculative
// solve of wide_kernels.hpp) put ONE call site of it between the two.
// PFIX > 0: p is known when the kernel is compiled (the formula's kernels, wide_kernels.hpp): the staging loops and the
// sums are cut to size and the other sizes are not compiled at all.
// The part of the state that wide_advance_pre reads from memory, as registers of the calling wavefront: a kernel that
// knows the LM step may be its to run (the group reducers of wide_step_kernel) asks for it BEFORE it finds out -- the
// state was written by the previous launch, on another XCD, and the loads are most of two microseconds.
template <int IT>
struct WideStateRegs
{
    double sv[9], d[8], ba[IT];
    int i[8];
};
template <int P>
__device__ __forceinline__ void wide_state_load(const WState *S, int lane, WideStateRegs<(P * (P + 1) / 2 + 63) / 64> &r)
{
    constexpr int NA = P * (P + 1) / 2, IT = (NA + 63) / 64;
    const bool mine = lane < P;
    const double *vec[9] = {S->x, S->xt, S->dx, S->vel, S->acc, S->g, S->diag, S->lo, S->up};
#pragma unroll
    for (int k = 0; k < 9; ++k)
        r.sv[k] = mine ? vec[k][lane] : 0.0;
    r.d[0] = S->fnorm2, r.d[1] = S->mu, r.d[2] = S->nu, r.d[3] = S->delta, r.d[4] = S->avratio, r.d[5] = S->chisq0;
    r.d[6] = S->chisq1, r.d[7] = S->chisq_init;
    r.i[0] = S->bad_steps, r.i[1] = S->niter, r.i[2] = S->phase, r.i[3] = S->status, r.i[4] = S->info, r.i[5] = S->nevalf;
    r.i[6] = S->nevaldf, r.i[7] = S->nevalfvv;
#pragma unroll
    for (int i = 0; i < IT; ++i)
        r.ba[i] = lane + 64 * i < NA ? S->A[lane + 64 * i] : 0.0;
}

template <int PFIX = 0, class REGS = void>
__device__ __forceinline__ void wide_advance_pre(const WAdvanceArgs &a, WideLds &L, WideCtx &c, const REGS *regs = nullptr)
{
    const int lane = wide_lane();
    WState *S = a.state;
    const int p = PFIX > 0 ? PFIX : a.p, NA = p * (p + 1) / 2;
    const LmParams prm = a.prm;
    c.want = 0;
    c.rejected = false;
    c.active = true;
    const double *tot = a.totals;
    const double *rA = tot + 2;
    const bool mine = lane < p;
    // ---- state -> LDS (vectors, packed matrix) and registers (scalars) ----
    // Every global load of the call is issued here, before the first of them is waited for: a load is most of a
    // microsecond for this lone wavefront (the data was written by other XCDs), and a copy loop of NA / 64 trips that
    // waits for each trip's load in turn cost 5 us at p = 32.  J^T J of the pass that just ran is parked in L.M (free
    // until the solve) for lm_take_point.  (regs: the caller has them already.)
    const double r_ssr = tot[0], r_badj = tot[1];
    const double r_g = mine ? tot[2 + NA + lane] : 0.0;
    double sv[9];
    double fnorm2, mu, nu, delta, avratio, chisq0, chisq1, chisq_init;
    int bad_steps, niter, phase, status, info, nevalf, nevaldf, nevalfvv;
    if constexpr (!__is_same(REGS, void))
    {
#pragma unroll
        for (int k = 0; k < 9; ++k)
            sv[k] = regs->sv[k];
        fnorm2 = regs->d[0], mu = regs->d[1], nu = regs->d[2], delta = regs->d[3], avratio = regs->d[4], chisq0 = regs->d[5];
        chisq1 = regs->d[6], chisq_init = regs->d[7];
        bad_steps = regs->i[0], niter = regs->i[1], phase = regs->i[2], status = regs->i[3], info = regs->i[4];
        nevalf = regs->i[5], nevaldf = regs->i[6], nevalfvv = regs->i[7];
    }
    else
    {
        const double *vec[9] = {S->x, S->xt, S->dx, S->vel, S->acc, S->g, S->diag, S->lo, S->up};
#pragma unroll
        for (int k = 0; k < 9; ++k)
            sv[k] = mine ? vec[k][lane] : 0.0;
        fnorm2 = S->fnorm2, mu = S->mu, nu = S->nu, delta = S->delta, avratio = S->avratio, chisq0 = S->chisq0;
        chisq1 = S->chisq1, chisq_init = S->chisq_init;
        bad_steps = S->bad_steps, niter = S->niter, phase = S->phase, status = S->status, info = S->info;
        nevalf = S->nevalf, nevaldf = S->nevaldf, nevalfvv = S->nevalfvv;
    }
    const int phase_in = phase; // (checked below, after every load of the call is in flight)
    auto stage_matrices = [&](auto trips) {
        constexpr int IT = decltype(trips)::value;
        double ba[IT], br[IT];
#pragma unroll
        for (int i = 0; i < IT; ++i)
        {
            const int k = lane + 64 * i;
            if constexpr (!__is_same(REGS, void))
                ba[i] = regs->ba[i];
            else
                ba[i] = k < NA ? S->A[k] : 0.0;
            br[i] = k < NA ? rA[k] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < IT; ++i)
        {
            const int k = lane + 64 * i;
            if (k < NA)
            {
                L.A[k] = ba[i];
                L.M[k] = br[i];
            }
        }
    };
    if constexpr (PFIX > 0)
        stage_matrices(WideInt<(PFIX * (PFIX + 1) / 2 + 63) / 64>{});
    else if (NA <= 64 * 3)
        stage_matrices(WideInt<3>{});
    else if (NA <= 64 * 9)
        stage_matrices(WideInt<9>{});
    else if (NA <= 64 * 19)
        stage_matrices(WideInt<19>{});
    else
        stage_matrices(WideInt<(WNA + 63) / 64>{});
    if (phase_in == PH_DONE)
    {
        c.active = false;
        return; // (a launch enqueued past the end of the fit: nothing is written)
    }
    if (mine)
    {
        L.x[lane] = sv[0];
        L.xt[lane] = sv[1];
        L.dx[lane] = sv[2];
        L.vel[lane] = sv[3];
        L.acc[lane] = sv[4];
        L.g[lane] = sv[5];
        L.diag[lane] = sv[6];
        L.lo[lane] = sv[7];
        L.up[lane] = sv[8];
    }
    const int niter_before = niter, phase_before = phase;
    wide_lds_sync();

    auto take_point = [&]() { // lm_take_point: x <- xt, g, A, fnorm2 from the pass
        if (mine)
        {
            L.x[lane] = L.xt[lane];
            L.g[lane] = r_g;
        }
        for (int k = lane; k < NA; k += 64)
            L.A[k] = L.M[k];
        fnorm2 = r_ssr;
        wide_lds_sync();
    };
    auto scale = [&](bool init) { // GSL scaling.c on the diagonal of J^T J
        if (mine)
        {
            if (prm.scale == 1)
            {
                if (init)
                    L.diag[lane] = 1.0;
            }
            else
            {
                double norm = sqrt(L.A[tri(lane, lane)]);
                if (norm == 0.0)
                    norm = 1.0;
                if (init || prm.scale == 2)
                    L.diag[lane] = norm;
                else
                    L.diag[lane] = fmax(L.diag[lane], norm);
            }
        }
        wide_lds_sync();
    };
    auto test = [&](int *inf) -> int { // gsl_multifit_nlinear_test
        const bool fail = mine && !(fabs(L.dx[lane]) < prm.xtol * prm.xtol + prm.xtol * fabs(L.x[lane]));
        if (__ballot(fail) == 0)
        {
            *inf = 1;
            return ST_SUCCESS;
        }
        const double gnorm = wide_wave_max(mine ? fabs(fmax(L.x[lane], 1.0) * L.g[lane]) : 0.0);
        const double phi = 0.5 * fnorm2;
        if (gnorm <= prm.gtol * fmax(phi, 1.0))
        {
            *inf = 2;
            return ST_SUCCESS;
        }
        *inf = 0;
        return ST_CONTINUE;
    };
    auto end_iteration = [&](int itstatus) -> bool { // lm_end_iteration
        const int iter = niter;
        niter += 1;
        chisq1 = fnorm2;
        if (itstatus == ST_EBADFUNC || (itstatus == ST_ENOPROG && iter == 0))
        {
            info = itstatus;
            status = itstatus;
            phase = PH_DONE;
            return false;
        }
        int inf = 0;
        const int t = test(&inf);
        info = inf;
        if (t == ST_SUCCESS)
        {
            status = ST_SUCCESS;
            phase = PH_DONE;
            return false;
        }
        if (niter >= prm.maxiter)
        {
            status = ST_EMAXITER;
            phase = PH_DONE;
            return false;
        }
        chisq0 = chisq1;
        bad_steps = 0;
        return true;
    };
    bool step = false;
    int want = 0; // which damped solve this call ends with: 1 acceleration, 2 velocity
    if (phase == PH_INIT)
    {
        nevalf += 1;
        bool ok = true;
        if (prm.jac_analytic)
        {
            nevaldf += 1;
            if (!(r_badj == 0.0))
                ok = false;
        }
        else
            nevalf += lm_fd_cost(prm, p);
        take_point();
        if (!ok)
        {
            chisq_init = chisq0 = chisq1 = r_ssr;
            status = ST_EBADFUNC;
            info = ST_EBADFUNC;
            phase = PH_DONE;
        }
        else
        {
            scale(true);
            const double u = mine ? L.diag[lane] * L.x[lane] : 0.0;
            const double Dx2 = wide_seq_sum(u * u, p);
            const double mx = wide_wave_max(mine ? sqrt(L.A[tri(lane, lane)]) / L.diag[lane] : -1.0);
            delta = 0.3 * fmax(1.0, sqrt(Dx2));
            mu = 1.0e-3 * mx * mx;
            nu = 2.0;
            avratio = 0.0;
            chisq_init = r_ssr;
            chisq0 = chisq1 = (prm.chisq_in == prm.chisq_in) ? prm.chisq_in : r_ssr;
            niter = 0;
            bad_steps = 0;
            step = true;
        }
    }
    else if (phase == PH_FVV)
    {
        if (prm.fvv_analytic)
            nevalfvv += 1;
        else
            nevalf += 1;
        if (prm.fvv_analytic && !(r_badj == 0.0))
        {
            // a failed fvv counts as a rejected step (src/trust.c:452-483, :530-545)
            delta /= prm.factor_down;
            lmd_nielsen_reject(mu, nu);
            const int itstatus = (++bad_steps > LMD_MAX_REJECTS) ? ST_ENOPROG : ST_CONTINUE;
            step = (itstatus == ST_CONTINUE) ? true : end_iteration(itstatus);
        }
        else
            want = 1;
    }
    else
    {
        // PH_TRIAL: trust_eval_step + radius / mu updates (src/trust.c:474-545)
        nevalf += 1;
        double rho;
        if (!(r_ssr < fnorm2))
            rho = -1.0;
        else
        {
            // lm_preduction: v^T (J^T J) v, row i of the product by lane i (j ascending), the outer sums in index order
            double row = 0.0;
            if (mine)
                for (int j = 0; j < p; ++j)
                    row += L.A[j <= lane ? tri(lane, j) : tri(j, lane)] * L.vel[j];
            const double vl = mine ? L.vel[lane] : 0.0;
            const double vAv = wide_seq_sum(row * vl, p);
            const double ud = mine ? L.diag[lane] * vl : 0.0;
            const double Dv2 = wide_seq_sum(ud * ud, p);
            rho = lmd_rho_of(r_ssr, fnorm2, vAv, Dv2, mu);
        }
        const bool found = lmd_step_found(rho, prm.trs, avratio, prm.avmax);
        lmd_radius(rho, prm.factor_up, prm.factor_down, delta);
        int itstatus = ST_CONTINUE;
        if (found)
        {
            itstatus = ST_SUCCESS;
            if (prm.jac_analytic)
            {
                nevaldf += 1;
                if (!(r_badj == 0.0))
                    itstatus = ST_EBADFUNC;
            }
            else
                nevalf += lm_fd_cost(prm, p);
            if (itstatus == ST_SUCCESS)
            {
                take_point();
                scale(false);
                lmd_nielsen_accept(rho, mu, nu);
                bad_steps = 0;
            }
        }
        else
        {
            lmd_nielsen_reject(mu, nu);
            if (++bad_steps > LMD_MAX_REJECTS)
                itstatus = ST_ENOPROG;
            c.rejected = itstatus == ST_CONTINUE;
        }
        step = (itstatus == ST_CONTINUE) ? true : end_iteration(itstatus);
    }
    if (step)
        want = 2;
    if (want)
    {
        // the one damped solve of this call: the acceleration (rhs = -J^T fvv) or the velocity of lm_begin_step (rhs = -g)
        if (mine)
            L.rhs[lane] = want == 1 ? -r_g : -L.g[lane];
        wide_lds_sync();
    }
    c.fnorm2 = fnorm2, c.mu = mu, c.nu = nu, c.delta = delta, c.avratio = avratio, c.chisq0 = chisq0, c.chisq1 = chisq1,
    c.chisq_init = chisq_init;
    c.bad_steps = bad_steps, c.niter = niter, c.phase = phase, c.status = status, c.info = info, c.nevalf = nevalf,
    c.nevaldf = nevaldf, c.nevalfvv = nevalfvv;
    c.niter_before = niter_before, c.phase_before = phase_before;
    c.want = want;
}

template <int PFIX = 0>
__device__ __forceinline__ int wide_advance_post(const WAdvanceArgs &a, WideLds &L, const WideCtx &c) // -> the phase it leaves
{
    if (!c.active)
        return PH_DONE;
    const int lane = wide_lane();
    WState *S = a.state;
    const int p = PFIX > 0 ? PFIX : a.p, NA = p * (p + 1) / 2;
    const LmParams prm = a.prm;
    const bool mine = lane < p;
    double fnorm2 = c.fnorm2, mu = c.mu, nu = c.nu, delta = c.delta, avratio = c.avratio, chisq0 = c.chisq0, chisq1 = c.chisq1,
           chisq_init = c.chisq_init;
    int bad_steps = c.bad_steps, niter = c.niter, phase = c.phase, status = c.status, info = c.info, nevalf = c.nevalf,
        nevaldf = c.nevaldf, nevalfvv = c.nevalfvv;
    const int niter_before = c.niter_before, phase_before = c.phase_before, want = c.want;
    auto set_trial = [&]() { // trust_trial_step_lu
        if (mine)
        {
            const double dxi = L.dx[lane], xi = L.x[lane];
            double xt = xi + dxi;
            if (prm.has_bounds)
            {
                if (xt < L.lo[lane])
                    xt = xi + (dxi / fmax(fabs(dxi), delta) * fabs(xi - L.lo[lane]));
                else if (xt > L.up[lane])
                    xt = xi + (dxi / fmax(fabs(dxi), delta) * fabs(xi - L.up[lane]));
            }
            L.xt[lane] = xt;
        }
        wide_lds_sync();
    };
    if (want == 1)
    {
        const double ai = mine ? L.acc[lane] : 0.0, vi = mine ? L.vel[lane] : 0.0;
        const double an = wide_seq_sum(ai * ai, p), vn = wide_seq_sum(vi * vi, p);
        avratio = sqrt(an) / sqrt(vn);
        if (mine)
            L.dx[lane] = vi + 0.5 * ai;
        wide_lds_sync();
        set_trial();
        phase = PH_TRIAL;
    }
    else if (want == 2)
    {
        if (prm.trs == 1)
            phase = PH_FVV;
        else
        {
            if (mine)
            {
                L.acc[lane] = 0.0;
                L.dx[lane] = L.vel[lane];
            }
            wide_lds_sync();
            set_trial();
            phase = PH_TRIAL;
        }
    }
    if (prm.bench_hold && phase == PH_DONE)
    {
        // timing mode: never finish, so that every step pays a full pass
        phase = PH_TRIAL;
        status = ST_CONTINUE;
        mu = 1.0;
        nu = 2.0;
        bad_steps = 0;
    }
    // ---- LDS / registers -> state (+ trace rows, + the host's copy when the fit has ended) ----
    const bool done = phase == PH_DONE;
    for (int rep = 0; rep < (done && a.host_mirror ? 2 : 1); ++rep)
    {
        WState *D = rep == 0 ? S : a.host_mirror;
        // (the state in memory: what this call cannot have changed is not written again -- the bounds never, and x, g, D,
        // J^T J not by a rejected trial, the most common step of a fit near its end: 4.5 KB at p = 32 that the end of the
        // launch would wait for)
        const bool all = rep != 0, moved = all || !c.rejected;
        if (mine)
        {
            D->xt[lane] = L.xt[lane];
            D->dx[lane] = L.dx[lane];
            D->vel[lane] = L.vel[lane];
            D->acc[lane] = L.acc[lane];
            if (moved)
            {
                D->x[lane] = L.x[lane];
                D->g[lane] = L.g[lane];
                D->diag[lane] = L.diag[lane];
            }
            if (all)
            {
                D->lo[lane] = L.lo[lane];
                D->up[lane] = L.up[lane];
            }
        }
        if (moved)
            for (int k = lane; k < NA; k += 64)
                D->A[k] = L.A[k];
        if (lane == 0)
        {
            D->fnorm2 = fnorm2;
            D->mu = mu;
            D->nu = nu;
            D->delta = delta;
            D->avratio = avratio;
            D->chisq0 = chisq0;
            D->chisq1 = chisq1;
            D->chisq_init = chisq_init;
            D->bad_steps = bad_steps;
            D->niter = niter;
            D->phase = phase;
            D->status = status;
            D->info = info;
            D->nevalf = nevalf;
            D->nevaldf = nevaldf;
            D->nevalfvv = nevalfvv;
            D->p = p;
            D->end_launch = a.launch_idx;
        }
    }
    if (a.ssrtrace)
    {
        // callback (src/nls.c:980-995): trace row 0 after init, row niter after each iteration
        if (phase_before == PH_INIT)
        {
            if (lane == 0)
                a.ssrtrace[0] = chisq_init;
            if (mine)
                a.partrace[(size_t)(prm.maxiter + 1) * lane] = L.x[lane];
        }
        else if (niter != niter_before && status != ST_EBADFUNC && !(status == ST_ENOPROG && niter_before == 0))
        {
            if (lane == 0)
                a.ssrtrace[niter] = chisq1;
            if (mine)
                a.partrace[niter + (size_t)(prm.maxiter + 1) * lane] = L.x[lane];
        }
    }
    if (done && a.done_seq)
    {
        __threadfence_system(); // every lane: its own stores to the host's copy are out before the completion word
        wide_lds_sync();
        if (lane == 0)
            __hip_atomic_store(a.done_seq, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return phase;
}

__device__ __forceinline__ void wide_advance(const WAdvanceArgs &a, WideLds &L)
{
    WideCtx c;
    wide_advance_pre(a, L, c);
    if (c.active && c.want)
        wide_solve(L, a.p, c.mu, L.rhs, c.want == 1 ? L.acc : L.vel, wide_lane(), a.pivoted);
    wide_advance_post(a, L, c);
}

#endif // __HIPCC__

} // namespace gslnls
