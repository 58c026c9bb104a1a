// lm_core.hpp -- the Levenberg-Marquardt trust-region state machine that runs ON DEVICE.
//
// One function, lm_advance(), consumes the sums produced by one pass over the n
// residual rows and moves the fit to its next trial point.  It is the device
// twin of (reference file:line)
//   gsl_multifit_nlinear_driver2      src/nls_fit.c:40-121   (iterate / test loop)
//   trust_init_LD                     src/trust.c:311-372    (D, delta, mu0)
//   trust_iterate_lu_LD               src/trust.c:408-549    (trial, rho, accept/reject)
//   lm_step_LD                        src/trust.c:223-292    (velocity, acceleration)
//   nielsen_init/accept/reject        src/trust.c:149-199
//   trust_trial_step_lu               src/trust.c:9-32       (bound projection)
// plus the GSL-upstream pieces those call through vtables:
// scaling.c (more/levenberg/marquardt), cholesky.c solver = modified Cholesky of
// J^T J + mu D^2, lm.c predicted reduction, convergence.c stopping rule.
//
// Everything n-sized lives in the pass kernels; this file only touches p-sized
// quantities, so J is never stored: ||J v||^2 = v^T (J^T J) v, column norms are
// sqrt(diag(J^T J)).
//
// The same header compiles for the three execution shapes of the library:
//   grid-per-fit (dense_kernels.hip), workgroup-per-fit and lane-per-fit
//   (batch_kernels.hip).  It is host-compilable only so that tests/ can drive the
//   state machine on a CPU-only box; the shipped library never runs it on the host.
#pragma once
#if defined(__HIPCC_RTC__)
#include "rtc_prelude.hpp"
#else
#include <math.h>
#include <float.h>
#include <stdint.h>
#endif

#if defined(__HIPCC__)
#define GSLNLS_HD __host__ __device__ __forceinline__
#else
#define GSLNLS_HD inline
#endif
#include "lm_decide.hpp" // the scalar decisions of a trial, shared by every variant of the state machine

namespace gslnls
{

#if defined(GSLNLS_STAMPS) && defined(__HIPCC__)
__device__ unsigned long long g_adv_stamps[8];
#endif
#if defined(GSLNLS_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
#define GSLNLS_ADV_STAMP(k)                                                                  \
    do                                                                                       \
    {                                                                                        \
        unsigned long long t__;                                                              \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");          \
        if (blockIdx.x == 0 && threadIdx.x == 0)                                             \
            g_adv_stamps[k] = t__;                                                           \
    } while (0)
#else
#define GSLNLS_ADV_STAMP(k) \
    do                      \
    {                       \
    } while (0)
#endif

// GSL errno values placed in `conv`
enum
{
    ST_SUCCESS = 0,
    ST_FAILURE = -1,
    ST_CONTINUE = -2,
    ST_EBADFUNC = 9,
    ST_EMAXITER = 11,
    ST_ENOPROG = 27
};

// what the NEXT pass over the rows has to compute
enum
{
    PH_INIT = 0,  // f, J^T J, J^T f at xt (= start)
    PH_TRIAL = 1, // same sums at the trial point xt
    PH_FVV = 2,   // J(x)^T fvv(x; vel)  (lmaccel only)
    PH_DONE = 3
};

struct LmParams
{
    int maxiter;
    int trs;          // 0 lm, 1 lmaccel (control_int[2])
    int scale;        // 0 more, 1 levenberg, 2 marquardt (control_int[3])
    int fdtype;       // 0 forward, 1 center (control_int[5])
    int jac_analytic; // !Rf_isNull(jac)
    int fvv_analytic; // !Rf_isNull(fvv)
    int has_bounds;   // Rf_isMatrix(lupars)
    int has_weights;  // !Rf_isNull(swts)
    int bench_hold;   // timing mode of the library's own benchmark hook; 0 in every fit
    double factor_up, factor_down, avmax, h_df, h_fvv, xtol, ftol, gtol;
    // driver2 starts every iteration with chisq0 <- chisq1; callers that re-enter it with a stale chisq1
    // (IRLS restarts, src/nls_irls.c:447-464) pass that value here; NaN = use the ssr at the start point
    double chisq_in;
};

template <int P>
struct PassSums
{
    static constexpr int NA = P * (P + 1) / 2;
    static constexpr int NV = 2 + NA + P;
    double ssr;   // sum f_i^2
    double badj;  // 0, or NaN when a non-finite analytic Jacobian entry was seen (src/nls.c:899-907)
    double A[NA]; // lower triangle of J^T J, packed row by row: (i,j), j<=i -> i(i+1)/2+j
    double g[P];  // J^T f   (PH_FVV: J^T fvv)
};

template <int P>
struct LmState
{
    static constexpr int NA = P * (P + 1) / 2;
    double x[P], xt[P], dx[P], vel[P], acc[P], g[P], diag[P], lo[P], up[P];
    double A[NA];
    double fnorm2, mu, nu, delta, avratio, chisq0, chisq1, chisq_init;
    int bad_steps, niter, phase, status, info, nevalf, nevaldf, nevalfvv;
};

GSLNLS_HD int tri(int i, int j) { return i * (i + 1) / 2 + j; }

// ---------------------------------------------------------------------------------
// (J^T J + mu D^2) sol = rhs  through GSL's modified Cholesky with diagonal pivoting
// (gsl_linalg_mcholesky_decomp/_solve as used by multifit_nlinear/cholesky.c:
// Gill-Murray-Wright, P (A+E) P^T = L D L^T).  All loops have compile-time bounds and
// static indices so the p x p system stays in registers; the data-dependent pivot is
// applied by predicated row/column swaps and the permutation is undone by selects.
template <int P>
GSLNLS_HD void lm_solve(const double *Ap, const double *diag, double mu, const double *rhs, double *sol)
{
    // only the lower triangle M[i][j], j <= i, is kept (the matrix is symmetric and stays so): no mirrored copies
    // to maintain, and a symmetric interchange touches each stored element once
    double M[P][P];
    int perm[P];
#pragma unroll
    for (int i = 0; i < P; ++i)
    {
        perm[i] = i;
#pragma unroll
        for (int j = 0; j <= i; ++j)
        {
            double v = Ap[tri(i, j)];
            if (i == j)
                v += mu * diag[i] * diag[i];
            M[i][j] = v;
        }
    }
    double b[P];
#pragma unroll
    for (int i = 0; i < P; ++i)
        b[i] = rhs[i];
    double gamma = 0.0, xi = 0.0;
#pragma unroll
    for (int i = 0; i < P; ++i)
    {
        gamma = fmax(gamma, fabs(M[i][i]));
#pragma unroll
        for (int j = 0; j < i; ++j)
            xi = fmax(xi, fabs(M[i][j]));
    }
    double beta;
    if (P == 1)
        beta = fmax(fmax(gamma, xi), DBL_EPSILON);
    else
        beta = fmax(fmax(gamma, xi / sqrt((double)P * P - 1.0)), DBL_EPSILON);
    const double betainv = 1.0 / sqrt(beta);
    double dinv[P];
#pragma unroll
    for (int j = 0; j < P; ++j)
    {
        int q = j;
        double maxd = fabs(M[j][j]);
#pragma unroll
        for (int i = j + 1; i < P; ++i)
        {
            const double d = fabs(M[i][i]);
            if (d > maxd)
            {
                maxd = d;
                q = i;
            }
        }
        // symmetric interchange of rows/columns j and q (q > j), written for each candidate q = i with static
        // indices; in the lower triangle:  (j,j) <-> (i,i);  (j,k) <-> (i,k) for k < j;  (k,j) <-> (i,k) for
        // j < k < i;  (k,j) <-> (k,i) for k > i;  (i,j) stays
#pragma unroll
        for (int i = j + 1; i < P; ++i)
        {
            if (i == q)
            {
                {
                    const double t = M[j][j];
                    M[j][j] = M[i][i];
                    M[i][i] = t;
                }
#pragma unroll
                for (int k = 0; k < j; ++k)
                {
                    const double t = M[j][k];
                    M[j][k] = M[i][k];
                    M[i][k] = t;
                }
#pragma unroll
                for (int k = j + 1; k < i; ++k)
                {
                    const double t = M[k][j];
                    M[k][j] = M[i][k];
                    M[i][k] = t;
                }
#pragma unroll
                for (int k = i + 1; k < P; ++k)
                {
                    const double t = M[k][j];
                    M[k][j] = M[k][i];
                    M[k][i] = t;
                }
                const double tb = b[j];
                b[j] = b[i];
                b[i] = tb;
                const int tp = perm[j];
                perm[j] = perm[i];
                perm[i] = tp;
            }
        }
        double theta = 0.0;
#pragma unroll
        for (int i = j + 1; i < P; ++i)
            theta = fmax(theta, fabs(M[i][j]));
        const double u = theta * betainv;
        const double alpha = fmax(fmax(DBL_EPSILON, fabs(M[j][j])), u * u);
        const double ainv = 1.0 / alpha;
        dinv[j] = ainv;
#pragma unroll
        for (int i = j + 1; i < P; ++i)
        {
            const double vi = M[i][j];
#pragma unroll
            for (int k = j + 1; k <= i; ++k)
                M[i][k] -= ainv * vi * M[k][j];
        }
#pragma unroll
        for (int i = j + 1; i < P; ++i)
            M[i][j] *= ainv;
        M[j][j] = alpha;
    }
#pragma unroll
    for (int i = 0; i < P; ++i)
    {
#pragma unroll
        for (int j = 0; j < i; ++j)
            b[i] -= M[i][j] * b[j];
    }
#pragma unroll
    for (int i = 0; i < P; ++i)
        b[i] *= dinv[i];
#pragma unroll
    for (int i = P - 1; i >= 0; --i)
    {
#pragma unroll
        for (int j = i + 1; j < P; ++j)
            b[i] -= M[j][i] * b[j];
    }
    // sol[perm[i]] = b[i] without dynamic indexing
#pragma unroll
    for (int i = 0; i < P; ++i)
    {
#pragma unroll
        for (int k = 0; k < P; ++k)
            if (perm[i] == k)
                sol[k] = b[i];
    }
}

// plain Cholesky of the packed lower triangle; returns det(A) = (prod L_ii)^2, 0 when not
// positive definite: det_cholesky_jtj, src/nls_utils.c:55-73
template <int P>
GSLNLS_HD double det_cholesky(const double *Ap)
{
    double L[P][P];
#pragma unroll
    for (int i = 0; i < P; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j)
            L[i][j] = Ap[tri(i, j)];
    double det = 1.0;
    bool ok = true;
#pragma unroll
    for (int j = 0; j < P; ++j)
    {
        double ajj = L[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k)
            ajj -= L[j][k] * L[j][k];
        if (!(ajj > 0.0))
            ok = false;
        ajj = sqrt(ajj);
        L[j][j] = ajj;
        det *= ajj;
#pragma unroll
        for (int i = j + 1; i < P; ++i)
        {
            double s = L[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k)
                s -= L[i][k] * L[j][k];
            L[i][j] = s / ajj;
        }
    }
    return ok ? det * det : 0.0;
}

// (J^T J)^{-1} from the packed lower triangle (column-major p x p output, symmetric);
// fills NaN when A is not positive definite.  Stands in for gsl_multifit_nlinear_covar
// (src/nls.c:603-608) on the normal equations.
template <int P>
GSLNLS_HD void covar_from_jtj(const double *Ap, double *cov)
{
    double L[P][P], Li[P][P];
#pragma unroll
    for (int i = 0; i < P; ++i)
#pragma unroll
        for (int j = 0; j < P; ++j)
        {
            L[i][j] = (j <= i) ? Ap[tri(i, j)] : 0.0;
            Li[i][j] = 0.0;
        }
    bool ok = true;
#pragma unroll
    for (int j = 0; j < P; ++j)
    {
        double ajj = L[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k)
            ajj -= L[j][k] * L[j][k];
        if (!(ajj > 0.0))
            ok = false;
        ajj = sqrt(ajj);
        L[j][j] = ajj;
#pragma unroll
        for (int i = j + 1; i < P; ++i)
        {
            double s = L[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k)
                s -= L[i][k] * L[j][k];
            L[i][j] = s / ajj;
        }
    }
#pragma unroll
    for (int j = 0; j < P; ++j)
    {
        Li[j][j] = 1.0 / L[j][j];
#pragma unroll
        for (int i = j + 1; i < P; ++i)
        {
            double s = 0.0;
#pragma unroll
            for (int k = j; k < i; ++k)
                s -= L[i][k] * Li[k][j];
            Li[i][j] = s / L[i][i];
        }
    }
#pragma unroll
    for (int i = 0; i < P; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j)
        {
            double s = 0.0;
#pragma unroll
            for (int k = i; k < P; ++k)
                s += Li[k][i] * Li[k][j];
            if (!ok)
                s = NAN;
            cov[i + P * j] = s;
            cov[j + P * i] = s;
        }
}

// ---------------------------------------------------------------------------------
template <int P>
GSLNLS_HD void lm_state_reset(LmState<P> &s, const double *start, const double *lupars)
{
#pragma unroll
    for (int k = 0; k < P; ++k)
    {
        s.x[k] = start[k];
        s.xt[k] = start[k];
        s.dx[k] = 0.0;
        s.vel[k] = 0.0;
        s.acc[k] = 0.0;
        s.g[k] = 0.0;
        s.diag[k] = 1.0;
        // lupars is 2 x p column-major [lower, upper] pairs (src/nls.c:248-263)
        const double lo = lupars ? lupars[2 * k] : -INFINITY;
        const double up = lupars ? lupars[2 * k + 1] : INFINITY;
        s.lo[k] = isfinite(lo) ? lo : -INFINITY;
        s.up[k] = isfinite(up) ? up : INFINITY;
    }
#pragma unroll
    for (int k = 0; k < LmState<P>::NA; ++k)
        s.A[k] = 0.0;
    s.fnorm2 = INFINITY;
    s.mu = 0.0;
    s.nu = 2.0;
    s.delta = 0.0;
    s.avratio = 0.0;
    s.chisq0 = s.chisq1 = s.chisq_init = INFINITY;
    s.bad_steps = 0;
    s.niter = 0;
    s.phase = PH_INIT;
    s.status = ST_CONTINUE;
    s.info = ST_CONTINUE;
    s.nevalf = s.nevaldf = s.nevalfvv = 0;
}

// x_trial = x + dx, shrunk toward a violated bound (trust_trial_step_lu, src/trust.c:9-32)
template <int P>
GSLNLS_HD void lm_set_trial(LmState<P> &s, const LmParams &prm)
{
#pragma unroll
    for (int i = 0; i < P; ++i)
    {
        const double dxi = s.dx[i], xi = s.x[i];
        double xt = xi + dxi;
        if (prm.has_bounds)
        {
            if (xt < s.lo[i])
                xt = xi + (dxi / fmax(fabs(dxi), s.delta) * fabs(xi - s.lo[i]));
            else if (xt > s.up[i])
                xt = xi + (dxi / fmax(fabs(dxi), s.delta) * fabs(xi - s.up[i]));
        }
        s.xt[i] = xt;
    }
}

// lm_step (src/trust.c:223-250): velocity from the damped normal equations; without
// acceleration the trial point follows immediately, with it the next pass is PH_FVV.
template <int P>
GSLNLS_HD void lm_begin_step(LmState<P> &s, const LmParams &prm)
{
    double rhs[P];
#pragma unroll
    for (int i = 0; i < P; ++i)
        rhs[i] = -s.g[i];
    lm_solve<P>(s.A, s.diag, s.mu, rhs, s.vel);
    if (prm.trs == 1)
    {
        s.phase = PH_FVV;
        return;
    }
#pragma unroll
    for (int i = 0; i < P; ++i)
    {
        s.acc[i] = 0.0;
        s.dx[i] = s.vel[i];
    }
    lm_set_trial(s, prm);
    s.phase = PH_TRIAL;
}

// gsl_multifit_nlinear_test (GSL convergence.c; App. A.7), called from src/nls_fit.c:102
template <int P>
GSLNLS_HD int lm_test(const LmState<P> &s, const LmParams &prm, int *info)
{
    bool ok = true;
#pragma unroll
    for (int i = 0; i < P; ++i)
    {
        const double tol = prm.xtol * prm.xtol + prm.xtol * fabs(s.x[i]);
        if (ok && !(fabs(s.dx[i]) < tol))
            ok = false;
    }
    if (ok)
    {
        *info = 1;
        return ST_SUCCESS;
    }
    double gnorm = 0.0;
#pragma unroll
    for (int i = 0; i < P; ++i)
    {
        const double t = fabs(fmax(s.x[i], 1.0) * s.g[i]);
        if (t > gnorm)
            gnorm = t;
    }
    const double phi = 0.5 * s.fnorm2;
    if (gnorm <= prm.gtol * fmax(phi, 1.0))
    {
        *info = 2;
        return ST_SUCCESS;
    }
    *info = 0;
    return ST_CONTINUE;
}

// bookkeeping at the end of one driver2 iteration (src/nls_fit.c:73-103).
// Returns true when the driver goes round again, i.e. the caller has to start the next step.
template <int P>
GSLNLS_HD bool lm_end_iteration(LmState<P> &s, const LmParams &prm, int itstatus)
{
    const int iter = s.niter; // driver2's `iter` before ++
    s.niter += 1;
    s.chisq1 = s.fnorm2;
    if (itstatus == ST_EBADFUNC || (itstatus == ST_ENOPROG && iter == 0))
    {
        s.info = itstatus;
        s.status = itstatus;
        s.phase = PH_DONE;
        return false;
    }
    int info = 0;
    const int t = lm_test(s, prm, &info);
    s.info = info;
    if (t == ST_SUCCESS)
    {
        s.status = ST_SUCCESS;
        s.phase = PH_DONE;
        return false;
    }
    if (s.niter >= prm.maxiter)
    {
        s.status = ST_EMAXITER;
        s.phase = PH_DONE;
        return false;
    }
    // next driver2 iteration: chisq0 <- chisq1, fresh trust_iterate call
    s.chisq0 = s.chisq1;
    s.bad_steps = 0;
    return true;
}

// GSL scaling.c on the diagonal of J^T J
template <int P>
GSLNLS_HD void lm_scale(LmState<P> &s, const LmParams &prm, bool init)
{
#pragma unroll
    for (int j = 0; j < P; ++j)
    {
        if (prm.scale == 1)
        {
            if (init)
                s.diag[j] = 1.0;
        }
        else
        {
            double norm = sqrt(s.A[tri(j, j)]);
            if (norm == 0.0)
                norm = 1.0;
            if (init || prm.scale == 2)
                s.diag[j] = norm;
            else
                s.diag[j] = fmax(s.diag[j], norm);
        }
    }
}

template <int P>
GSLNLS_HD void lm_take_point(LmState<P> &s, const PassSums<P> &r)
{
#pragma unroll
    for (int k = 0; k < P; ++k)
    {
        s.x[k] = s.xt[k];
        s.g[k] = r.g[k];
    }
#pragma unroll
    for (int k = 0; k < LmState<P>::NA; ++k)
        s.A[k] = r.A[k];
    s.fnorm2 = r.ssr;
}

// number of f-evaluations the reference would have charged for one Jacobian (App. A.8)
GSLNLS_HD int lm_fd_cost(const LmParams &prm, int p) { return prm.fdtype ? 2 * p : p; }

// The state machine.  `r` holds the sums of the pass that `s.phase` asked for.
// Every path that continues with a new step (initialisation, accepted step, rejected step) only decides so and
// falls through to ONE lm_begin_step at the end: the damped solve is the bulk of this function, and where lanes
// of a wavefront hold different fits (multi-start) they would otherwise run one copy of it per path, one after
// the other.
template <int P>
GSLNLS_HD void lm_advance(LmState<P> &s, const PassSums<P> &r, const LmParams &prm)
{
    if (s.phase == PH_DONE)
        return;

    bool step = false;
    if (s.phase == PH_INIT)
    {
        // trust_init_LD (src/trust.c:311-372): f, J, g, D, delta, mu0
        s.nevalf += 1;
        bool ok = true;
        if (prm.jac_analytic)
        {
            s.nevaldf += 1;
            if (!(r.badj == 0.0))
            {
                // gsl_df returned GSL_EBADFUNC (src/nls.c:899-907): init fails, driver sees it at once
                lm_take_point(s, r);
                s.chisq_init = s.chisq0 = s.chisq1 = r.ssr;
                s.status = ST_EBADFUNC;
                s.info = ST_EBADFUNC;
                s.phase = PH_DONE;
                ok = false;
            }
        }
        else
            s.nevalf += lm_fd_cost(prm, P);
        if (ok)
        {
            lm_take_point(s, r);
            lm_scale(s, prm, true);
            double Dx2 = 0.0, mx = -1.0;
#pragma unroll
            for (int j = 0; j < P; ++j)
            {
                const double u = s.diag[j] * s.x[j];
                Dx2 += u * u;
                mx = fmax(mx, sqrt(s.A[tri(j, j)]) / s.diag[j]);
            }
            s.delta = 0.3 * fmax(1.0, sqrt(Dx2));
            s.mu = 1.0e-3 * mx * mx;
            s.nu = 2.0;
            s.avratio = 0.0;
            s.chisq_init = r.ssr;
            s.chisq0 = s.chisq1 = (prm.chisq_in == prm.chisq_in) ? prm.chisq_in : r.ssr;
            s.niter = 0;
            s.bad_steps = 0;
            step = true;
        }
    }
    else if (s.phase == PH_FVV)
    {
        // acceleration solve (src/trust.c:252-289): rhs = -J^T fvv, same damped matrix
        if (prm.fvv_analytic)
            s.nevalfvv += 1;
        else
            s.nevalf += 1;
        if (prm.fvv_analytic && !(r.badj == 0.0))
        {
            // gsl_fvv returned GSL_EBADFUNC (src/nls.c:963-970): lm_step fails, and trust_iterate_lu_LD counts a
            // failed step as a rejected one (src/trust.c:452-483, :530-545): rho = -1, so the radius shrinks, mu grows,
            // and the loop tries again with the shorter velocity (which may make fvv finite) -- up to 15 times
            s.delta /= prm.factor_down;
            lmd_nielsen_reject(s.mu, s.nu);
            const int itstatus = (++s.bad_steps > LMD_MAX_REJECTS) ? ST_ENOPROG : ST_CONTINUE;
            if ((itstatus == ST_CONTINUE) ? true : lm_end_iteration(s, prm, itstatus))
                lm_begin_step(s, prm);
            return;
        }
        double rhs[P];
#pragma unroll
        for (int i = 0; i < P; ++i)
            rhs[i] = -r.g[i];
        lm_solve<P>(s.A, s.diag, s.mu, rhs, s.acc);
        double an = 0.0, vn = 0.0;
#pragma unroll
        for (int i = 0; i < P; ++i)
        {
            an += s.acc[i] * s.acc[i];
            vn += s.vel[i] * s.vel[i];
        }
        s.avratio = sqrt(an) / sqrt(vn);
#pragma unroll
        for (int i = 0; i < P; ++i)
            s.dx[i] = s.vel[i] + 0.5 * s.acc[i];
        lm_set_trial(s, prm);
        s.phase = PH_TRIAL;
        return;
    }
    else
    {
        // PH_TRIAL: trust_eval_step + radius/mu updates (src/trust.c:474-545)
        GSLNLS_ADV_STAMP(0);
        s.nevalf += 1;
        double rho;
        if (!(r.ssr < s.fnorm2))
            rho = -1.0; // ||f_trial|| >= ||f|| (also catches +Inf residuals and NaN)
        else
        {
            // lm_preduction: (||J v||/||f||)^2 + 2 mu (||D v||/||f||)^2 with v the velocity
            double vAv = 0.0, Dv2 = 0.0;
#pragma unroll
            for (int i = 0; i < P; ++i)
            {
                double row = 0.0;
#pragma unroll
                for (int j = 0; j < P; ++j)
                    row += s.A[j <= i ? tri(i, j) : tri(j, i)] * s.vel[j];
                vAv += row * s.vel[i];
                const double u = s.diag[i] * s.vel[i];
                Dv2 += u * u;
            }
            rho = lmd_rho_of(r.ssr, s.fnorm2, vAv, Dv2, s.mu);
        }
        GSLNLS_ADV_STAMP(1);
        const bool found = lmd_step_found(rho, prm.trs, s.avratio, prm.avmax);
        lmd_radius(rho, prm.factor_up, prm.factor_down, s.delta);

        int itstatus = ST_CONTINUE; // CONTINUE = the iteration is not over (rejected step, another trial follows)
        if (found)
        {
            itstatus = ST_SUCCESS;
            if (prm.jac_analytic)
            {
                s.nevaldf += 1;
                if (!(r.badj == 0.0))
                    itstatus = ST_EBADFUNC;
            }
            else
                s.nevalf += lm_fd_cost(prm, P);
            if (itstatus == ST_SUCCESS)
            {
                lm_take_point(s, r);
                lm_scale(s, prm, false);
                lmd_nielsen_accept(rho, s.mu, s.nu); // (src/trust.c:175-188)
                s.bad_steps = 0;
            }
        }
        else
        {
            lmd_nielsen_reject(s.mu, s.nu); // (src/trust.c:190-199)
            if (++s.bad_steps > LMD_MAX_REJECTS)
                itstatus = ST_ENOPROG;
        }
        GSLNLS_ADV_STAMP(2);
        step = (itstatus == ST_CONTINUE) ? true : lm_end_iteration(s, prm, itstatus);
        GSLNLS_ADV_STAMP(3);
    }
    if (step)
        lm_begin_step(s, prm);
    GSLNLS_ADV_STAMP(4);
}

} // namespace gslnls
