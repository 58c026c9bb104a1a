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

// (A + mu D^2) sol = rhs by wave 0 (all 64 lanes of it must call; lanes >= p idle along).  Only the lower triangle
// M[i][k], k <= i, is kept (leading dimension p + 1), as in lm_solve<P>.
__device__ __forceinline__ void wide_solve(WideLds &L, int p, double mu, const double *rhs, double *sol, int lane)
{
    const int LD = p + 1;
    double *M = L.M;
    double *colj = L.row; // column j of the current step, contiguous (L.row is free while a solve runs)
    if (lane < p)
    {
        for (int j = 0; j <= lane; ++j)
        {
            double v = L.A[tri(lane, j)];
            if (j == lane)
                v += mu * L.diag[lane] * L.diag[lane];
            M[lane * LD + j] = v;
        }
    }
    wide_lds_sync();
    double b = lane < p ? rhs[lane] : 0.0;
    int perm = lane;
    double gm = 0.0, xm = 0.0;
    if (lane < p)
    {
        gm = fabs(M[lane * LD + lane]);
        for (int j = 0; j < lane; ++j)
            xm = fmax(xm, fabs(M[lane * LD + j]));
    }
    const double gamma = wide_wave_max(gm), xi = wide_wave_max(xm);
    double beta;
    if (p == 1)
        beta = fmax(fmax(gamma, xi), DBL_EPSILON);
    else
        beta = fmax(fmax(gamma, xi / sqrt((double)p * p - 1.0)), DBL_EPSILON);
    const double betainv = 1.0 / sqrt(beta);
    double dinv = 0.0;                                  // lane j keeps 1 / alpha_j
    double dg = lane < p ? M[lane * LD + lane] : 0.0;   // lane i keeps the current diagonal entry M[i][i]
    for (int j = 0; j < p; ++j)
    {
        // pivot: first index of the largest |diagonal| among j..p-1 (`if (d > maxd)` of the sequential scan: the first
        // element wins ties, NaNs never win)
        const double d = (lane >= j && lane < p) ? fabs(dg) : -1.0;
        const double maxd = wide_wave_max(d);
        const unsigned long long hit = __ballot(lane >= j && lane < p && d == maxd);
        int q = hit ? (int)__builtin_ctzll(hit) : j;
        const double djj = wide_bcast(dg, j);
        if (!(maxd > fabs(djj)))
            q = j;
        if (q != j)
        {
            // symmetric interchange of rows / columns j and q in the lower triangle, one element pair per lane:
            //   k < j: (j,k) <-> (q,k);  j < k < q: (k,j) <-> (q,k);  k > q: (k,j) <-> (k,q);  the diagonal entries swap
            if (lane < p && lane != j && lane != q)
            {
                const int a = lane < j ? j * LD + lane : lane * LD + j;
                const int c = lane < q ? q * LD + lane : lane * LD + q;
                const double t = M[a];
                M[a] = M[c];
                M[c] = t;
            }
            const double dq = wide_bcast(dg, q);
            const double bj = wide_bcast(b, j), bq = wide_bcast(b, q);
            const int pj = __builtin_amdgcn_readlane(perm, __builtin_amdgcn_readfirstlane(j)),
                      pq = __builtin_amdgcn_readlane(perm, __builtin_amdgcn_readfirstlane(q));
            if (lane == j)
            {
                dg = dq;
                b = bq;
                perm = pq;
            }
            if (lane == q)
            {
                dg = djj;
                b = bj;
                perm = pj;
            }
            wide_lds_sync();
        }
        const double vi = (lane > j && lane < p) ? M[lane * LD + j] : 0.0;
        const double theta = wide_wave_max(fabs(vi));
        const double u = theta * betainv;
        const double alpha = fmax(fmax(DBL_EPSILON, fabs(wide_bcast(dg, j))), u * u);
        const double ainv = 1.0 / alpha;
        if (lane == j)
        {
            dinv = ainv;
            dg = alpha;
        }
        colj[lane] = vi;
        wide_lds_sync();
        if (lane > j && lane < p)
        {
            // M[i][k] -= ainv * vi * M[k][j], k = j+1..i: eight at a time, all loads of a group ahead of its stores
            // (the compiler cannot tell that rows and the column copy never overlap and would serialise every element)
            int k = j + 1;
            for (; k + 7 < lane; k += 8)
            {
                double m[8], c[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                {
                    m[u] = M[lane * LD + k + u];
                    c[u] = colj[k + u];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    m[u] -= ainv * vi * c[u];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    M[lane * LD + k + u] = m[u];
            }
            for (; k + 3 < lane; k += 4)
            {
                const double m0 = M[lane * LD + k], m1 = M[lane * LD + k + 1], m2 = M[lane * LD + k + 2], m3 = M[lane * LD + k + 3];
                const double c0 = colj[k], c1 = colj[k + 1], c2 = colj[k + 2], c3 = colj[k + 3];
                double r0 = m0, r1 = m1, r2 = m2, r3 = m3;
                r0 -= ainv * vi * c0;
                r1 -= ainv * vi * c1;
                r2 -= ainv * vi * c2;
                r3 -= ainv * vi * c3;
                M[lane * LD + k] = r0;
                M[lane * LD + k + 1] = r1;
                M[lane * LD + k + 2] = r2;
                M[lane * LD + k + 3] = r3;
            }
            for (; k < lane; ++k)
            {
                double m = M[lane * LD + k];
                m -= ainv * vi * colj[k];
                M[lane * LD + k] = m;
            }
            dg -= ainv * vi * vi; // k = i: the diagonal entry lives in a register
            M[lane * LD + j] = vi * ainv;
        }
        wide_lds_sync();
    }
    // L z = P b: column sweep, every b_i collects its terms in ascending j like the row form
    for (int j = 0; j < p; ++j)
    {
        const double bj = wide_bcast(b, j);
        if (lane > j && lane < p)
            b -= M[lane * LD + j] * bj;
    }
    b *= dinv;
    // L^T w = z: column sweep from the last column (j descending per element; lm_solve<P> adds them ascending)
    for (int j = p - 1; j >= 1; --j)
    {
        const double bj = wide_bcast(b, j);
        if (lane < j)
            b -= M[j * LD + lane] * bj;
    }
    if (lane < p)
        sol[perm] = b;
    wide_lds_sync();
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
        {
            if (mine)
                L.rhs[lane] = -rg[lane];
            wide_lds_sync();
            wide_solve(L, p, mu, L.rhs, L.acc, lane);
            const double ai = mine ? L.acc[lane] : 0.0, vi = mine ? L.vel[lane] : 0.0;
            const double an = wide_seq_sum(ai * ai, p), vn = wide_seq_sum(vi * vi, p);
            avratio = sqrt(an) / sqrt(vn);
            if (mine)
                L.dx[lane] = vi + 0.5 * ai;
            wide_lds_sync();
            set_trial();
            phase = PH_TRIAL;
        }
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
    {
        // lm_begin_step
        if (mine)
            L.rhs[lane] = -L.g[lane];
        wide_lds_sync();
        wide_solve(L, p, mu, L.rhs, L.vel, lane);
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
