// wide_core.hpp -- the Levenberg-Marquardt state machine for 10 <= p <= 64 parameters, run by ONE workgroup.
//
// Same algorithm, same order of operations as lm_advance<P>() in lm_core.hpp (which see for the reference's
// file:line of every step: trust_init_LD / trust_iterate_lu_LD / lm_step_LD / nielsen_* src/trust.c, driver2
// src/nls_fit.c:40-121, GSL scaling.c / cholesky.c (mcholesky) / convergence.c) -- but p is a run-time value,
// the state lives in LDS instead of registers, and the p x p algebra is spread over the lanes of a wavefront:
// the reference allocates its n x p workspace for any p (src/nls.c:266) and so does the formula front end
// (R/nls.R:588-599), while lm_core.hpp unrolls everything for p <= 9.
//
//   * p-sized vectors and the packed lower triangle of J^T J: LDS copies for the duration of the call;
//   * scalar control flow (rho, accept / reject, mu, delta, stopping rule): thread 0, decisions published through LDS;
//   * modified Cholesky with diagonal pivoting (gsl_linalg_mcholesky, Gill-Murray-Wright) of J^T J + mu D^2: wave 0,
//     lane i owns row i of a full symmetric copy in LDS (leading dimension p + 1: conflict-free column walks), pivot
//     search by a DPP maximum + ballot, the rank-one update of column step j by all rows at once;
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
constexpr int WT_ADV = 256;            // threads of the advancing workgroup

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

__device__ __forceinline__ double wide_bcast(double v, int src_lane)
{
    const long long bits = __double_as_longlong(v);
    const int lo = __shfl((int)(bits & 0xffffffffll), src_lane, 64), hi = __shfl((int)(bits >> 32), src_lane, 64);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

__device__ __forceinline__ double wide_wave_max(double v)
{
    // fmax is exact and order independent: any reduction tree gives the same bits
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
    {
        const long long bits = __double_as_longlong(v);
        const int lo = __shfl_xor((int)(bits & 0xffffffffll), m, 64), hi = __shfl_xor((int)(bits >> 32), m, 64);
        v = fmax(v, __longlong_as_double(((long long)hi << 32) | (unsigned int)lo));
    }
    return v;
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
    double fnorm2, mu, nu, delta, avratio, chisq0, chisq1, chisq_init;
    int bad_steps, niter, phase, status, info, nevalf, nevaldf, nevalfvv;
    int do_step, do_take, do_solve_acc, advanced, niter_before, phase_before;
};

// (A + mu D^2) sol = rhs by wave 0 (all 64 lanes of it must call; lanes >= p idle along)
__device__ __forceinline__ void wide_solve(WideLds &L, int p, double mu, const double *rhs, double *sol, int lane)
{
    const int LD = p + 1;
    double *M = L.M;
    // build the full symmetric matrix: lane i fills row i
    if (lane < p)
    {
        for (int j = 0; j < p; ++j)
        {
            double v = L.A[j <= lane ? tri(lane, j) : tri(j, lane)];
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
    double dinv = 0.0; // lane j keeps 1 / alpha_j
    for (int j = 0; j < p; ++j)
    {
        // pivot: first index of the largest |diagonal| among j..p-1
        const double d = (lane >= j && lane < p) ? fabs(M[lane * LD + lane]) : -1.0;
        const double maxd = wide_wave_max(d);
        const unsigned long long hit = __ballot(lane >= j && lane < p && d == maxd);
        int q = hit ? (int)__builtin_ctzll(hit) : j;
        if (!(maxd > fabs(M[j * LD + j])))
            q = j; // `if (d > maxd)` of the sequential scan: the first element wins ties (and NaNs never win)
        if (q != j)
        {
            // symmetric interchange of rows / columns j and q on the full matrix
            if (lane < p)
            {
                const double t = M[j * LD + lane];
                M[j * LD + lane] = M[q * LD + lane];
                M[q * LD + lane] = t;
            }
            wide_lds_sync();
            if (lane < p)
            {
                const double t = M[lane * LD + j];
                M[lane * LD + j] = M[lane * LD + q];
                M[lane * LD + q] = t;
            }
            wide_lds_sync();
            const double bj = wide_bcast(b, j), bq = wide_bcast(b, q);
            const int pj = __shfl(perm, j, 64), pq = __shfl(perm, q, 64);
            if (lane == j)
            {
                b = bq;
                perm = pq;
            }
            if (lane == q)
            {
                b = bj;
                perm = pj;
            }
        }
        const double vi = (lane > j && lane < p) ? M[lane * LD + j] : 0.0;
        const double theta = wide_wave_max(fabs(vi));
        const double u = theta * betainv;
        const double alpha = fmax(fmax(DBL_EPSILON, fabs(M[j * LD + j])), u * u);
        const double ainv = 1.0 / alpha;
        if (lane == j)
            dinv = ainv;
        if (lane > j && lane < p)
        {
            for (int k = j + 1; k <= lane; ++k)
            {
                double m = M[lane * LD + k];
                m -= ainv * vi * M[k * LD + j];
                M[lane * LD + k] = m;
                M[k * LD + lane] = m; // keep the mirror: later interchanges walk rows across the diagonal
            }
        }
        wide_lds_sync();
        if (lane > j && lane < p)
        {
            const double l = vi * ainv;
            M[lane * LD + j] = l;
            M[j * LD + lane] = l;
        }
        if (lane == j)
            M[j * LD + j] = alpha;
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

// one workgroup of WT_ADV threads; all threads call
__device__ __forceinline__ void wide_advance(const WAdvanceArgs &a, WideLds &L)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    WState *S = a.state;
    const int p = S->p, NA = p * (p + 1) / 2;
    const LmParams prm = a.prm;
    if (S->phase == PH_DONE)
        return;
    const double *tot = a.totals;
    const double r_ssr = tot[0], r_badj = tot[1];
    const double *rA = tot + 2, *rg = tot + 2 + NA;
    // ---- state -> LDS ----
    for (int k = tid; k < p; k += WT_ADV)
    {
        L.x[k] = S->x[k];
        L.xt[k] = S->xt[k];
        L.dx[k] = S->dx[k];
        L.vel[k] = S->vel[k];
        L.acc[k] = S->acc[k];
        L.g[k] = S->g[k];
        L.diag[k] = S->diag[k];
        L.lo[k] = S->lo[k];
        L.up[k] = S->up[k];
    }
    for (int k = tid; k < NA; k += WT_ADV)
        L.A[k] = S->A[k];
    if (tid == 0)
    {
        L.fnorm2 = S->fnorm2;
        L.mu = S->mu;
        L.nu = S->nu;
        L.delta = S->delta;
        L.avratio = S->avratio;
        L.chisq0 = S->chisq0;
        L.chisq1 = S->chisq1;
        L.chisq_init = S->chisq_init;
        L.bad_steps = S->bad_steps;
        L.niter = S->niter;
        L.phase = S->phase;
        L.status = S->status;
        L.info = S->info;
        L.nevalf = S->nevalf;
        L.nevaldf = S->nevaldf;
        L.nevalfvv = S->nevalfvv;
        L.do_step = 0;
        L.do_take = 0;
        L.do_solve_acc = 0;
        L.advanced = 1;
        L.niter_before = S->niter;
        L.phase_before = S->phase;
    }
    __syncthreads();

    auto take_point = [&]() { // lm_take_point: x <- xt, g, A, fnorm2 from the pass (all threads)
        for (int k = tid; k < p; k += WT_ADV)
        {
            L.x[k] = L.xt[k];
            L.g[k] = rg[k];
        }
        for (int k = tid; k < NA; k += WT_ADV)
            L.A[k] = rA[k];
        if (tid == 0)
            L.fnorm2 = r_ssr;
        __syncthreads();
    };
    auto scale = [&](bool init) { // GSL scaling.c on the diagonal of J^T J (all threads)
        for (int j = tid; j < p; j += WT_ADV)
        {
            if (prm.scale == 1)
            {
                if (init)
                    L.diag[j] = 1.0;
            }
            else
            {
                double norm = sqrt(L.A[tri(j, j)]);
                if (norm == 0.0)
                    norm = 1.0;
                if (init || prm.scale == 2)
                    L.diag[j] = norm;
                else
                    L.diag[j] = fmax(L.diag[j], norm);
            }
        }
        __syncthreads();
    };
    auto test = [&](int *info) -> int { // gsl_multifit_nlinear_test (thread 0)
        bool ok = true;
        for (int i = 0; i < p; ++i)
        {
            const double tol = prm.xtol * prm.xtol + prm.xtol * fabs(L.x[i]);
            if (ok && !(fabs(L.dx[i]) < tol))
                ok = false;
        }
        if (ok)
        {
            *info = 1;
            return ST_SUCCESS;
        }
        double gnorm = 0.0;
        for (int i = 0; i < p; ++i)
        {
            const double t = fabs(fmax(L.x[i], 1.0) * L.g[i]);
            if (t > gnorm)
                gnorm = t;
        }
        const double phi = 0.5 * L.fnorm2;
        if (gnorm <= prm.gtol * fmax(phi, 1.0))
        {
            *info = 2;
            return ST_SUCCESS;
        }
        *info = 0;
        return ST_CONTINUE;
    };
    auto end_iteration = [&](int itstatus) -> bool { // lm_end_iteration (thread 0)
        const int iter = L.niter;
        L.niter += 1;
        L.chisq1 = L.fnorm2;
        if (itstatus == ST_EBADFUNC || (itstatus == ST_ENOPROG && iter == 0))
        {
            L.info = itstatus;
            L.status = itstatus;
            L.phase = PH_DONE;
            return false;
        }
        int info = 0;
        const int t = test(&info);
        L.info = info;
        if (t == ST_SUCCESS)
        {
            L.status = ST_SUCCESS;
            L.phase = PH_DONE;
            return false;
        }
        if (L.niter >= prm.maxiter)
        {
            L.status = ST_EMAXITER;
            L.phase = PH_DONE;
            return false;
        }
        L.chisq0 = L.chisq1;
        L.bad_steps = 0;
        return true;
    };
    auto set_trial = [&]() { // trust_trial_step_lu (threads < p)
        for (int i = tid; i < p; i += WT_ADV)
        {
            const double dxi = L.dx[i], xi = L.x[i];
            double xt = xi + dxi;
            if (prm.has_bounds)
            {
                if (xt < L.lo[i])
                    xt = xi + (dxi / fmax(fabs(dxi), L.delta) * fabs(xi - L.lo[i]));
                else if (xt > L.up[i])
                    xt = xi + (dxi / fmax(fabs(dxi), L.delta) * fabs(xi - L.up[i]));
            }
            L.xt[i] = xt;
        }
    };

    const int phase = L.phase;
    if (phase == PH_INIT)
    {
        if (tid == 0)
        {
            L.nevalf += 1;
            bool ok = true;
            if (prm.jac_analytic)
            {
                L.nevaldf += 1;
                if (!(r_badj == 0.0))
                    ok = false;
            }
            else
                L.nevalf += lm_fd_cost(prm, p);
            L.do_take = ok ? 1 : 2;
        }
        __syncthreads();
        const int how = L.do_take;
        take_point();
        if (how == 2)
        {
            if (tid == 0)
            {
                L.chisq_init = L.chisq0 = L.chisq1 = r_ssr;
                L.status = ST_EBADFUNC;
                L.info = ST_EBADFUNC;
                L.phase = PH_DONE;
            }
        }
        else
        {
            scale(true);
            if (tid == 0)
            {
                double Dx2 = 0.0, mx = -1.0;
                for (int j = 0; j < p; ++j)
                {
                    const double u = L.diag[j] * L.x[j];
                    Dx2 += u * u;
                    mx = fmax(mx, sqrt(L.A[tri(j, j)]) / L.diag[j]);
                }
                L.delta = 0.3 * fmax(1.0, sqrt(Dx2));
                L.mu = 1.0e-3 * mx * mx;
                L.nu = 2.0;
                L.avratio = 0.0;
                L.chisq_init = r_ssr;
                L.chisq0 = L.chisq1 = (prm.chisq_in == prm.chisq_in) ? prm.chisq_in : r_ssr;
                L.niter = 0;
                L.bad_steps = 0;
                L.do_step = 1;
            }
        }
        __syncthreads();
    }
    else if (phase == PH_FVV)
    {
        if (tid == 0)
        {
            if (prm.fvv_analytic)
                L.nevalfvv += 1;
            else
                L.nevalf += 1;
            if (prm.fvv_analytic && !(r_badj == 0.0))
            {
                // a failed fvv counts as a rejected step (src/trust.c:452-483, :530-545)
                L.delta /= prm.factor_down;
                L.mu *= L.nu;
                L.nu *= 2.0;
                const int itstatus = (++L.bad_steps > 15) ? ST_ENOPROG : ST_CONTINUE;
                L.do_step = ((itstatus == ST_CONTINUE) ? true : end_iteration(itstatus)) ? 1 : 0;
            }
            else
                L.do_solve_acc = 1;
        }
        __syncthreads();
        if (L.do_solve_acc)
        {
            for (int i = tid; i < p; i += WT_ADV)
                L.rhs[i] = -rg[i];
            __syncthreads();
            if (wave == 0)
                wide_solve(L, p, L.mu, L.rhs, L.acc, lane);
            __syncthreads();
            if (tid == 0)
            {
                double an = 0.0, vn = 0.0;
                for (int i = 0; i < p; ++i)
                {
                    an += L.acc[i] * L.acc[i];
                    vn += L.vel[i] * L.vel[i];
                }
                L.avratio = sqrt(an) / sqrt(vn);
                L.phase = PH_TRIAL;
            }
            for (int i = tid; i < p; i += WT_ADV)
                L.dx[i] = L.vel[i] + 0.5 * L.acc[i];
            __syncthreads();
            set_trial();
            __syncthreads();
        }
    }
    else
    {
        // PH_TRIAL: trust_eval_step + radius / mu updates (src/trust.c:474-545)
        // lm_preduction needs v^T (J^T J) v: row i of the product by thread i, the outer sum in index order by thread 0
        for (int i = tid; i < p; i += WT_ADV)
        {
            double row = 0.0;
            for (int j = 0; j < p; ++j)
                row += L.A[j <= i ? tri(i, j) : tri(j, i)] * L.vel[j];
            L.row[i] = row;
        }
        __syncthreads();
        if (tid == 0)
        {
            L.nevalf += 1;
            double rho;
            if (!(r_ssr < L.fnorm2))
                rho = -1.0;
            else
            {
                const double finv = 1.0 / L.fnorm2;
                const double ared = 1.0 - r_ssr * finv;
                double vAv = 0.0, Dv2 = 0.0;
                for (int i = 0; i < p; ++i)
                {
                    vAv += L.row[i] * L.vel[i];
                    const double u = L.diag[i] * L.vel[i];
                    Dv2 += u * u;
                }
                const double pred = vAv * finv + 2.0 * L.mu * (Dv2 * finv);
                rho = (pred > 0.0) ? ared / pred : -1.0;
            }
            bool found = rho > 0.0;
            if (prm.trs == 1 && L.avratio > prm.avmax)
                found = false;
            if (rho > 0.75)
                L.delta *= prm.factor_up;
            else if (rho < 0.25)
                L.delta /= prm.factor_down;
            int itstatus = ST_CONTINUE;
            L.do_take = 0;
            if (found)
            {
                itstatus = ST_SUCCESS;
                if (prm.jac_analytic)
                {
                    L.nevaldf += 1;
                    if (!(r_badj == 0.0))
                        itstatus = ST_EBADFUNC;
                }
                else
                    L.nevalf += lm_fd_cost(prm, p);
                if (itstatus == ST_SUCCESS)
                {
                    L.do_take = 1;
                    double b = 2.0 * rho - 1.0;
                    b = 1.0 - b * b * b;
                    L.nu = 2.0;
                    L.mu *= fmax(0.333333333333333, b);
                    L.bad_steps = 0;
                }
            }
            else
            {
                L.mu *= L.nu;
                L.nu *= 2.0;
                if (++L.bad_steps > 15)
                    itstatus = ST_ENOPROG;
            }
            L.do_solve_acc = itstatus; // parked for the second half below (after the accepted point has been taken)
        }
        __syncthreads();
        if (L.do_take)
        {
            take_point();
            scale(false);
        }
        if (tid == 0)
        {
            const int itstatus = L.do_solve_acc;
            L.do_solve_acc = 0;
            L.do_step = ((itstatus == ST_CONTINUE) ? true : end_iteration(itstatus)) ? 1 : 0;
        }
        __syncthreads();
    }
    // ---- lm_begin_step ----
    if (L.do_step)
    {
        for (int i = tid; i < p; i += WT_ADV)
            L.rhs[i] = -L.g[i];
        __syncthreads();
        if (wave == 0)
            wide_solve(L, p, L.mu, L.rhs, L.vel, lane);
        __syncthreads();
        if (prm.trs == 1)
        {
            if (tid == 0)
                L.phase = PH_FVV;
        }
        else
        {
            for (int i = tid; i < p; i += WT_ADV)
            {
                L.acc[i] = 0.0;
                L.dx[i] = L.vel[i];
            }
            __syncthreads();
            set_trial();
            if (tid == 0)
                L.phase = PH_TRIAL;
        }
        __syncthreads();
    }
    if (tid == 0 && prm.bench_hold && L.phase == PH_DONE)
    {
        // timing mode: never finish, so that every step pays a full pass
        L.phase = PH_TRIAL;
        L.status = ST_CONTINUE;
        L.mu = 1.0;
        L.nu = 2.0;
        L.bad_steps = 0;
    }
    __syncthreads();
    // ---- LDS -> state (+ trace rows, + the host's copy when the fit has ended) ----
    const bool done = L.phase == PH_DONE;
    for (int rep = 0; rep < (done ? 2 : 1); ++rep)
    {
        WState *D = rep == 0 ? S : a.host_mirror;
        for (int k = tid; k < p; k += WT_ADV)
        {
            D->x[k] = L.x[k];
            D->xt[k] = L.xt[k];
            D->dx[k] = L.dx[k];
            D->vel[k] = L.vel[k];
            D->acc[k] = L.acc[k];
            D->g[k] = L.g[k];
            D->diag[k] = L.diag[k];
            D->lo[k] = L.lo[k];
            D->up[k] = L.up[k];
        }
        for (int k = tid; k < NA; k += WT_ADV)
            D->A[k] = L.A[k];
        if (tid == 0)
        {
            D->fnorm2 = L.fnorm2;
            D->mu = L.mu;
            D->nu = L.nu;
            D->delta = L.delta;
            D->avratio = L.avratio;
            D->chisq0 = L.chisq0;
            D->chisq1 = L.chisq1;
            D->chisq_init = L.chisq_init;
            D->bad_steps = L.bad_steps;
            D->niter = L.niter;
            D->phase = L.phase;
            D->status = L.status;
            D->info = L.info;
            D->nevalf = L.nevalf;
            D->nevaldf = L.nevaldf;
            D->nevalfvv = L.nevalfvv;
            D->p = p;
            D->end_launch = a.launch_idx;
        }
    }
    if (a.ssrtrace)
    {
        // callback (src/nls.c:980-995): trace row 0 after init, row niter after each iteration
        if (L.phase_before == PH_INIT)
        {
            if (tid == 0)
                a.ssrtrace[0] = L.chisq_init;
            for (int k = tid; k < p; k += WT_ADV)
                a.partrace[(size_t)(prm.maxiter + 1) * k] = L.x[k];
        }
        else if (L.phase_before != PH_INIT && L.niter != L.niter_before && L.status != ST_EBADFUNC &&
                 !(L.status == ST_ENOPROG && L.niter_before == 0))
        {
            if (tid == 0)
                a.ssrtrace[L.niter] = L.chisq1;
            for (int k = tid; k < p; k += WT_ADV)
                a.partrace[L.niter + (size_t)(prm.maxiter + 1) * k] = L.x[k];
        }
    }
    if (done)
    {
        __syncthreads();
        if (tid == 0)
        {
            __threadfence_system();
            __hip_atomic_store(a.done_seq, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

#endif // __HIPCC__

} // namespace gslnls
