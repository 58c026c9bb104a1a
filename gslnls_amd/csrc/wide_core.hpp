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
};

#if defined(__HIPCC__) || defined(__HIPCC_RTC__)

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
template <int K>
__device__ __forceinline__ void wide_fmac_rowbcast(double &m, double vb, double t)
{
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(m) : "v"(vb), "v"(t), "n"(K));
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

// m[q] for a wave-uniform q: a scalar branch tree over the register file (a run-time register index would send the whole
// row to scratch memory)
template <int PW>
__device__ __forceinline__ double wide_pick(const double (&m)[PW], int q)
{
    double v = 0.0;
#define GSLNLS_WPICK(k)                                                                                                      \
    case k:                                                                                                                  \
        if constexpr (k < PW)                                                                                                \
        {                                                                                                                    \
            v = m[k < PW ? k : 0];                                                                                           \
            asm volatile("" : "+v"(v));                                                                                      \
        }                                                                                                                    \
        break;
#define GSLNLS_WPICK8(b)                                                                                                     \
    GSLNLS_WPICK(b + 0)                                                                                                      \
    GSLNLS_WPICK(b + 1) GSLNLS_WPICK(b + 2) GSLNLS_WPICK(b + 3) GSLNLS_WPICK(b + 4) GSLNLS_WPICK(b + 5) GSLNLS_WPICK(b + 6)  \
        GSLNLS_WPICK(b + 7)
    switch (__builtin_amdgcn_readfirstlane(q))
    {
        GSLNLS_WPICK8(0)
        GSLNLS_WPICK8(8)
        GSLNLS_WPICK8(16)
        GSLNLS_WPICK8(24)
        GSLNLS_WPICK8(32)
        GSLNLS_WPICK8(40)
        GSLNLS_WPICK8(48)
        GSLNLS_WPICK8(56)
    default:
        break;
    }
#undef GSLNLS_WPICK8
#undef GSLNLS_WPICK
    return v;
}

// (A + mu D^2) sol = rhs by ONE wavefront (all 64 lanes of it must call; lanes >= p idle along), p <= PW.
//
// gsl_linalg_mcholesky (Gill-Murray-Wright modified Cholesky with diagonal pivoting) with the matrix in REGISTERS: lane i
// holds row i of the symmetric matrix, m[k] = S[i][k], by ORIGINAL index -- nothing is ever interchanged.  The
// reference's permutation lives in `pos` (lane i: the position row i has in the permuted order; the tie rule of the pivot
// search -- the first position wins -- reads it), the pivot sequence in `ord` (lane j: the row eliminated at step j).
// One step: pivot row q (wavefront maximum of the diagonal, which every lane keeps in `dg`), its column v_i = S[i][q] is
// each lane's own m[q], theta = max |v_i|, alpha, then the rank-one update S[i][k] -= (v_i / alpha) v_k of ALL columns in
// PW instructions (v_fmac_f64 with a DPP row broadcast of v_k) -- columns and rows already eliminated see v = 0.  The
// forward substitution rides along (b_i -= l_i b_q); the multipliers l_i = v_i / alpha also go to LDS, T[q][i], so that
// the back substitution L^T w = z finds row q_j of L^T as T[i][q_j]: lane i's own row, conflict-free.
// Against lm_solve<P> / the reference: the same sums in the same order with one exception -- an entry S[i][k] whose rows
// were interchanged relative to each other is updated as (v_i / alpha) v_k here where the reference's single (lower)
// copy gets (v_k / alpha) v_i: one rounding apart.
template <int PW>
__device__ __forceinline__ void wide_solve_reg(WideLds &L, int p, double mu, const double *rhs, double *sol, int lane)
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
    double xm = 0.0;
#pragma unroll
    for (int k = 0; k < PW; ++k)
        xm = fmax(xm, fabs(m[k]));
    const double gamma = wide_wave_max_rows<R>(fabs(dg)), xi = wide_wave_max_rows<R>(xm);
    double beta;
    if (p == 1)
        beta = fmax(fmax(gamma, xi), DBL_EPSILON);
    else
        beta = fmax(fmax(gamma, xi / sqrt((double)p * p - 1.0)), DBL_EPSILON);
    const double betainv = 1.0 / sqrt(beta);
    double dinv = 0.0; // 1 / alpha of the step that eliminated this row
    int pos = lane, ord = 0;
    bool act = mine;
    const int sub = lane & 15;
    for (int j = 0; j < p; ++j)
    {
        // pivot: the first position holding the largest |diagonal| among positions j..p-1 (`if (d > maxd)` of the
        // sequential scan from position j: the first element wins ties, NaNs never win)
        const double d = act ? fabs(dg) : -1.0;
        const double maxd = wide_wave_max_rows<R>(d);
        const int r = (int)__builtin_ctzll(__ballot(act && pos == j) | (1ull << 63)); // the row at position j
        int q = r;
        if (maxd > fabs(wide_bcast(dg, r)))
        {
            unsigned long long hit = __ballot(act && d == maxd);
            q = hit ? (int)__builtin_ctzll(hit) : r;
            if (hit & (hit - 1))
            {
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
        }
        q = __builtin_amdgcn_readfirstlane(q);
        // rows r and q trade positions (nothing moves)
        const int pq = __builtin_amdgcn_readlane(pos, q);
        if (lane == r)
            pos = pq;
        if (lane == q)
            pos = j;
        ord = lane == j ? q : ord;
        const bool upd = act && lane != q;
        double v = wide_pick<PW>(m, q);
        v = upd ? v : 0.0;
        const double theta = wide_wave_max_rows<R>(fabs(v));
        const double u = theta * betainv;
        const double alpha = fmax(fmax(DBL_EPSILON, fabs(wide_bcast(dg, q))), u * u);
        const double ainv = 1.0 / alpha;
        if (lane == q)
        {
            dinv = ainv;
            act = false;
        }
        const double t = ainv * v; // the multiplier l_i (0 in rows that take no part)
        const double bq = wide_bcast(b, q);
        if (upd)
        {
            dg -= t * v;
            b -= t * bq;
        }
        T[q * LD + lane] = t;
        const double tn = -t;
#pragma unroll
        for (int g = 0; g < R; ++g)
        {
            double vb = wide_shfl(v, 16 * g + sub); // lane l: v of row 16 g + l % 16
            asm volatile("s_nop 1" : "+v"(vb));     // (a DPP operand written by the instruction before: 2 wait states)
            wide_fmac_rowbcast<0>(m[16 * g + 0], vb, tn);
            wide_fmac_rowbcast<1>(m[16 * g + 1], vb, tn);
            wide_fmac_rowbcast<2>(m[16 * g + 2], vb, tn);
            wide_fmac_rowbcast<3>(m[16 * g + 3], vb, tn);
            wide_fmac_rowbcast<4>(m[16 * g + 4], vb, tn);
            wide_fmac_rowbcast<5>(m[16 * g + 5], vb, tn);
            wide_fmac_rowbcast<6>(m[16 * g + 6], vb, tn);
            wide_fmac_rowbcast<7>(m[16 * g + 7], vb, tn);
            wide_fmac_rowbcast<8>(m[16 * g + 8], vb, tn);
            wide_fmac_rowbcast<9>(m[16 * g + 9], vb, tn);
            wide_fmac_rowbcast<10>(m[16 * g + 10], vb, tn);
            wide_fmac_rowbcast<11>(m[16 * g + 11], vb, tn);
            wide_fmac_rowbcast<12>(m[16 * g + 12], vb, tn);
            wide_fmac_rowbcast<13>(m[16 * g + 13], vb, tn);
            wide_fmac_rowbcast<14>(m[16 * g + 14], vb, tn);
            wide_fmac_rowbcast<15>(m[16 * g + 15], vb, tn);
        }
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

__device__ __forceinline__ void wide_solve(WideLds &L, int p, double mu, const double *rhs, double *sol, int lane)
{
    // (noinline-sized bodies, one per register-file width; the caller has ONE call site)
    if (p <= 16)
        wide_solve_reg<16>(L, p, mu, rhs, sol, lane);
    else if (p <= 32)
        wide_solve_reg<32>(L, p, mu, rhs, sol, lane);
    else if (p <= 48)
        wide_solve_reg<48>(L, p, mu, rhs, sol, lane);
    else
        wide_solve_reg<64>(L, p, mu, rhs, sol, lane);
}

// ONE wavefront (64 lanes >= p: lane k owns component k of every p-vector); all 64 lanes call.  Scalars of the state live
// in registers, identical in every lane; LDS traffic between lanes is ordered by wide_lds_sync().
__device__ __forceinline__ void wide_advance(const WAdvanceArgs &a, WideLds &L)
{
    const int lane = threadIdx.x & 63;
    WState *S = a.state;
    const int p = S->p, NA = p * (p + 1) / 2;
    const LmParams prm = a.prm;
    if (S->phase == PH_DONE)
        return;
    const double *tot = a.totals;
    const double r_ssr = tot[0], r_badj = tot[1];
    const double *rA = tot + 2, *rg = tot + 2 + NA;
    const bool mine = lane < p;
    // ---- state -> LDS (vectors, packed matrix) and registers (scalars) ----
    if (mine)
    {
        L.x[lane] = S->x[lane];
        L.xt[lane] = S->xt[lane];
        L.dx[lane] = S->dx[lane];
        L.vel[lane] = S->vel[lane];
        L.acc[lane] = S->acc[lane];
        L.g[lane] = S->g[lane];
        L.diag[lane] = S->diag[lane];
        L.lo[lane] = S->lo[lane];
        L.up[lane] = S->up[lane];
    }
    for (int k = lane; k < NA; k += 64)
        L.A[k] = S->A[k];
    double fnorm2 = S->fnorm2, mu = S->mu, nu = S->nu, delta = S->delta, avratio = S->avratio, chisq0 = S->chisq0,
           chisq1 = S->chisq1, chisq_init = S->chisq_init;
    int bad_steps = S->bad_steps, niter = S->niter, phase = S->phase, status = S->status, info = S->info, nevalf = S->nevalf,
        nevaldf = S->nevaldf, nevalfvv = S->nevalfvv;
    const int niter_before = niter, phase_before = phase;
    wide_lds_sync();

    auto take_point = [&]() { // lm_take_point: x <- xt, g, A, fnorm2 from the pass
        if (mine)
        {
            L.x[lane] = L.xt[lane];
            L.g[lane] = rg[lane];
        }
        for (int k = lane; k < NA; k += 64)
            L.A[k] = rA[k];
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
            mu *= nu;
            nu *= 2.0;
            const int itstatus = (++bad_steps > 15) ? ST_ENOPROG : ST_CONTINUE;
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
            const double finv = 1.0 / fnorm2;
            const double ared = 1.0 - r_ssr * finv;
            // lm_preduction: v^T (J^T J) v, row i of the product by lane i (j ascending), the outer sums in index order
            double row = 0.0;
            if (mine)
                for (int j = 0; j < p; ++j)
                    row += L.A[j <= lane ? tri(lane, j) : tri(j, lane)] * L.vel[j];
            const double vl = mine ? L.vel[lane] : 0.0;
            const double vAv = wide_seq_sum(row * vl, p);
            const double ud = mine ? L.diag[lane] * vl : 0.0;
            const double Dv2 = wide_seq_sum(ud * ud, p);
            const double pred = vAv * finv + 2.0 * mu * (Dv2 * finv);
            rho = (pred > 0.0) ? ared / pred : -1.0;
        }
        bool found = rho > 0.0;
        if (prm.trs == 1 && avratio > prm.avmax)
            found = false;
        if (rho > 0.75)
            delta *= prm.factor_up;
        else if (rho < 0.25)
            delta /= prm.factor_down;
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
                double b = 2.0 * rho - 1.0;
                b = 1.0 - b * b * b;
                nu = 2.0;
                mu *= fmax(0.333333333333333, b);
                bad_steps = 0;
            }
        }
        else
        {
            mu *= nu;
            nu *= 2.0;
            if (++bad_steps > 15)
                itstatus = ST_ENOPROG;
        }
        step = (itstatus == ST_CONTINUE) ? true : end_iteration(itstatus);
    }
    if (step)
        want = 2;
    if (want)
    {
        // the one damped solve of this call: the acceleration (rhs = -J^T fvv) or the velocity of lm_begin_step (rhs = -g)
        if (mine)
            L.rhs[lane] = want == 1 ? -rg[lane] : -L.g[lane];
        wide_lds_sync();
        wide_solve(L, p, mu, L.rhs, want == 1 ? L.acc : L.vel, lane);
    }
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
    for (int rep = 0; rep < (done ? 2 : 1); ++rep)
    {
        WState *D = rep == 0 ? S : a.host_mirror;
        if (mine)
        {
            D->x[lane] = L.x[lane];
            D->xt[lane] = L.xt[lane];
            D->dx[lane] = L.dx[lane];
            D->vel[lane] = L.vel[lane];
            D->acc[lane] = L.acc[lane];
            D->g[lane] = L.g[lane];
            D->diag[lane] = L.diag[lane];
            D->lo[lane] = L.lo[lane];
            D->up[lane] = L.up[lane];
        }
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
    if (done)
    {
        __threadfence_system(); // every lane: its own stores to the host's copy are out before the completion word
        wide_lds_sync();
        if (lane == 0)
            __hip_atomic_store(a.done_seq, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

#endif // __HIPCC__

} // namespace gslnls
