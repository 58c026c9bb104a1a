/*
 * gslref_multifit.c -- ORACLE (test infrastructure, never shipped).
 * Restates the dense trust-region solver: GSL gsl_multifit_nlinear "trust"
 * with the lm / lmaccel subproblem, as driven and extended by the reference:
 *   src/trust.c   (trust_init_LD :311-372, trust_iterate_lu_LD :408-549,
 *                  lm_step_LD :223-292, nielsen_* :149-199, rho :67-147,
 *                  bound projection :9-32)
 *   src/fdf.c     (winit_LD :23-77, eval_f_LD :94-113, eval_df_LD :135-177,
 *                  eval_fvv_LD :200-233)
 *   src/fdjac.c   (forward :24-64, center :81-128)
 *   src/fdfvv.c   (:35-77)
 *   src/nls_fit.c (gsl_multifit_nlinear_driver2 :40-121)
 * and the GSL-upstream pieces reached through vtables:
 * scaling.c, lm.c (step, preduction), cholesky.c / qr.c solvers, convergence.c.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include "gslref_internal.h"

static double dnrm2(int n, const double *x)
{
    /* scaled 2-norm like reference-BLAS dnrm2 */
    double scale = 0.0, ssq = 1.0;
    int i;
    for (i = 0; i < n; ++i)
    {
        if (x[i] != 0.0)
        {
            double a = fabs(x[i]);
            if (isinf(a))
                return INFINITY;
            if (isnan(a))
                return NAN;
            if (scale < a)
            {
                ssq = 1.0 + ssq * (scale / a) * (scale / a);
                scale = a;
            }
            else
                ssq += (a / scale) * (a / scale);
        }
    }
    return scale * sqrt(ssq);
}

static double ddot(int n, const double *a, const double *b)
{
    double s = 0.0;
    int i;
    for (i = 0; i < n; ++i)
        s += a[i] * b[i];
    return s;
}

static double colnorm(const gslref_ws *w, int j)
{
    double scale = 0.0, ssq = 1.0;
    int i;
    for (i = 0; i < w->n; ++i)
    {
        double v = w->J[(size_t)i * w->p + j];
        if (v != 0.0)
        {
            double a = fabs(v);
            if (scale < a)
            {
                ssq = 1.0 + ssq * (scale / a) * (scale / a);
                scale = a;
            }
            else
                ssq += (a / scale) * (a / scale);
        }
    }
    return scale * sqrt(ssq);
}

gslref_ws *gslref_ws_alloc(int n, int p)
{
    gslref_ws *w = (gslref_ws *)calloc(1, sizeof(gslref_ws));
    w->n = n;
    w->p = p;
    w->x = (double *)calloc(p, sizeof(double));
    w->f_ = (double *)calloc(n, sizeof(double));
    w->J = (double *)calloc((size_t)n * p, sizeof(double));
    w->g = (double *)calloc(p, sizeof(double));
    w->dx = (double *)calloc(p, sizeof(double));
    w->sqrt_wts_work = (double *)calloc(n, sizeof(double));
    w->diag = (double *)calloc(p, sizeof(double));
    w->x_trial = (double *)calloc(p, sizeof(double));
    w->f_trial = (double *)calloc(n, sizeof(double));
    w->workp = (double *)calloc(p, sizeof(double));
    w->workn = (double *)calloc(n, sizeof(double));
    w->vel = (double *)calloc(p, sizeof(double));
    w->acc = (double *)calloc(p, sizeof(double));
    w->fvvv = (double *)calloc(n, sizeof(double));
    w->JTJ = (double *)calloc((size_t)p * p, sizeof(double));
    w->work_JTJ = (double *)calloc((size_t)p * p, sizeof(double));
    w->rhs = (double *)calloc(p, sizeof(double));
    w->perm = (int *)calloc(p, sizeof(int));
    w->aug = NULL;
    w->augrhs = NULL;
    /* gsl_multifit_nlinear_default_parameters() */
    w->trs = 0;
    w->scale = 0;
    w->solver = 0;
    w->fdtype = 0;
    w->factor_up = 3.0;
    w->factor_down = 2.0;
    w->avmax = 0.75;
    w->h_df = sqrt(DBL_EPSILON);
    w->h_fvv = 0.02;
    return w;
}

void gslref_ws_free(gslref_ws *w)
{
    if (!w)
        return;
    free(w->x); free(w->f_); free(w->J); free(w->g); free(w->dx);
    free(w->sqrt_wts_work); free(w->diag); free(w->x_trial); free(w->f_trial);
    free(w->workp); free(w->workn); free(w->vel); free(w->acc); free(w->fvvv);
    free(w->JTJ); free(w->work_JTJ); free(w->rhs); free(w->perm);
    free(w->aug); free(w->augrhs);
    free(w);
}

/* y <- L y for unit-lower L (dtrmv Lower NoTrans Unit), src/fdf.c:107 */
static void trmv_unit_lower(int n, const double *L, double *y)
{
    int i, j;
    for (i = n - 1; i >= 0; --i)
    {
        double s = y[i];
        for (j = 0; j < i; ++j)
            s += L[(size_t)i * n + j] * y[j];
        y[i] = s;
    }
}

/* src/fdf.c:94-113 (GLS) and GSL gsl_multifit_nlinear_eval_f (diagonal weights) */
int gslref_eval_f(gslref_ws *w, const double *x, double *y)
{
    int s = w->f(x, w->params, y);
    int i;
    ++w->nevalf;
    if (w->Lw)
    {
        trmv_unit_lower(w->n, w->Lw, y);
        if (w->sqrt_wts)
            for (i = 0; i < w->n; ++i)
                y[i] *= w->sqrt_wts[i];
    }
    else if (w->sqrt_wts)
    {
        for (i = 0; i < w->n; ++i)
            y[i] *= w->sqrt_wts[i];
    }
    return s;
}

/* src/fdjac.c:24-64 */
static int forward_jac(gslref_ws *w, double *x, const double *f, double *J)
{
    const int n = w->n, p = w->p;
    int i, j, status = 0;
    double *col = w->workn;
    for (j = 0; j < p; ++j)
    {
        double xj = x[j];
        double delta = w->h_df * fabs(xj);
        if (delta == 0.0)
            delta = w->h_df;
        x[j] = xj + delta;
        status += gslref_eval_f(w, x, col);
        if (status)
        {
            x[j] = xj;
            return status;
        }
        x[j] = xj;
        delta = 1.0 / delta;
        for (i = 0; i < n; ++i)
            J[(size_t)i * p + j] = (col[i] - f[i]) * delta;
    }
    return status;
}

/* src/fdjac.c:81-128 */
static int center_jac(gslref_ws *w, double *x, double *J)
{
    const int n = w->n, p = w->p;
    int i, j, status = 0;
    double *fp = w->workn;
    double *fm = (double *)malloc(sizeof(double) * n);
    for (j = 0; j < p; ++j)
    {
        double xj = x[j];
        double delta = w->h_df * fabs(xj);
        if (delta == 0.0)
            delta = w->h_df;
        x[j] = xj + 0.5 * delta;
        status += gslref_eval_f(w, x, fp);
        if (status)
        {
            x[j] = xj;
            free(fm);
            return status;
        }
        x[j] = xj - 0.5 * delta;
        status += gslref_eval_f(w, x, fm);
        if (status)
        {
            x[j] = xj;
            free(fm);
            return status;
        }
        x[j] = xj;
        delta = 1.0 / delta;
        for (i = 0; i < n; ++i)
            J[(size_t)i * p + j] = (fp[i] - fm[i]) * delta;
    }
    free(fm);
    return status;
}

/* src/fdf.c:135-177 and GSL gsl_multifit_nlinear_eval_df */
int gslref_eval_df(gslref_ws *w, const double *x, const double *f, double *J)
{
    const int n = w->n, p = w->p;
    int status, i, j;
    if (w->df)
    {
        status = w->df(x, w->params, J);
        ++w->nevaldf;
        if (w->Lw)
        {
            /* J <- L J (dtrmm Left Lower NoTrans Unit), row by row from the bottom */
            int k;
            for (i = n - 1; i >= 0; --i)
                for (j = 0; j < p; ++j)
                {
                    double s = J[(size_t)i * p + j];
                    for (k = 0; k < i; ++k)
                        s += w->Lw[(size_t)i * n + k] * J[(size_t)k * p + j];
                    J[(size_t)i * p + j] = s;
                }
        }
        if (w->sqrt_wts)
            for (i = 0; i < n; ++i)
                for (j = 0; j < p; ++j)
                    J[(size_t)i * p + j] *= w->sqrt_wts[i];
    }
    else
    {
        double *xx = (double *)malloc(sizeof(double) * p);
        memcpy(xx, x, sizeof(double) * p);
        if (w->fdtype == 0)
            status = forward_jac(w, xx, f, J);
        else
            status = center_jac(w, xx, J);
        free(xx);
    }
    return status;
}

/* src/fdf.c:200-233 + src/fdfvv.c:35-77 */
static int eval_fvv(gslref_ws *w, const double *x, const double *v, const double *f,
                    const double *J, double *yvv)
{
    const int n = w->n, p = w->p;
    int status, i, j;
    if (w->fvv)
    {
        status = w->fvv(x, v, w->params, yvv);
        ++w->nevalfvv;
        if (w->Lw)
            trmv_unit_lower(n, w->Lw, yvv);
        if (w->sqrt_wts)
            for (i = 0; i < n; ++i)
                yvv[i] *= w->sqrt_wts[i];
    }
    else
    {
        const double h = w->h_fvv, hinv = 1.0 / h;
        for (i = 0; i < p; ++i)
            w->workp[i] = x[i] + h * v[i];
        status = gslref_eval_f(w, w->workp, yvv);
        if (status)
            return status;
        for (i = 0; i < n; ++i)
        {
            double u = 0.0;
            for (j = 0; j < p; ++j)
                u += J[(size_t)i * p + j] * v[j];
            yvv[i] = (2.0 * hinv) * ((yvv[i] - f[i]) * hinv - u);
        }
    }
    return status;
}

/* GSL scaling.c (App. A.2) */
static void scale_init(gslref_ws *w)
{
    int j;
    for (j = 0; j < w->p; ++j)
    {
        if (w->scale == 1)
            w->diag[j] = 1.0;
        else
        {
            double norm = colnorm(w, j);
            if (norm == 0.0)
                norm = 1.0;
            w->diag[j] = norm;
        }
    }
}

static void scale_update(gslref_ws *w)
{
    int j;
    if (w->scale == 1)
        return;
    for (j = 0; j < w->p; ++j)
    {
        double norm = colnorm(w, j);
        if (norm == 0.0)
            norm = 1.0;
        if (w->scale == 2)
            w->diag[j] = norm;
        else
            w->diag[j] = fmax(w->diag[j], norm);
    }
}

static double scaled_norm(int p, const double *D, const double *a)
{
    double e2 = 0.0;
    int i;
    for (i = 0; i < p; ++i)
    {
        double u = D[i] * a[i];
        e2 += u * u;
    }
    return sqrt(e2);
}

/* solver init (GSL cholesky.c cholesky_init: JTJ <- J^T J lower; qr.c: factor J) */
static void solver_init(gslref_ws *w)
{
    const int n = w->n, p = w->p;
    int i, j, k;
    if (w->solver == 1)
    {
        memset(w->JTJ, 0, sizeof(double) * (size_t)p * p);
        for (k = 0; k < n; ++k)
            for (i = 0; i < p; ++i)
            {
                double jki = w->J[(size_t)k * p + i];
                for (j = 0; j <= i; ++j)
                    w->JTJ[i * p + j] += jki * w->J[(size_t)k * p + j];
            }
    }
}

/* presolve(mu)+solve(f): v = argmin ||J v + f||^2 + mu ||D v||^2  (App. A.3) */
static int solver_solve(gslref_ws *w, double mu, const double *f, double *xout, int refactor)
{
    const int n = w->n, p = w->p;
    int i, j;
    if (w->solver == 1)
    {
        if (refactor)
        {
            for (i = 0; i < p; ++i)
                for (j = 0; j <= i; ++j)
                    w->work_JTJ[i * p + j] = w->JTJ[i * p + j];
            for (i = 0; i < p; ++i)
                w->work_JTJ[i * p + i] += mu * w->diag[i] * w->diag[i];
            gslref_mcholesky_decomp(p, w->work_JTJ, w->perm);
        }
        for (j = 0; j < p; ++j)
        {
            double s = 0.0;
            for (i = 0; i < n; ++i)
                s += w->J[(size_t)i * p + j] * f[i];
            w->rhs[j] = -s;
        }
        return gslref_mcholesky_solve(p, w->work_JTJ, w->perm, w->rhs, xout);
    }
    else
    {
        /* qr / svd: least squares of the augmented system, rebuilt per solve */
        const double sq = sqrt(mu);
        if (!w->aug)
        {
            w->aug = (double *)malloc(sizeof(double) * (size_t)(n + p) * p);
            w->augrhs = (double *)malloc(sizeof(double) * (size_t)(n + p));
        }
        memcpy(w->aug, w->J, sizeof(double) * (size_t)n * p);
        for (i = 0; i < p; ++i)
            for (j = 0; j < p; ++j)
                w->aug[(size_t)(n + i) * p + j] = (i == j) ? sq * w->diag[i] : 0.0;
        for (i = 0; i < n; ++i)
            w->augrhs[i] = -f[i];
        for (i = 0; i < p; ++i)
            w->augrhs[n + i] = 0.0;
        return gslref_lstsq(n + p, p, w->aug, w->augrhs, xout);
    }
}

/* trust_init (src/trust.c:311-372) */
static int trust_init(gslref_ws *w)
{
    const int n = w->n, p = w->p;
    int status, i, j;
    double Dx, max = -1.0;
    status = gslref_eval_f(w, w->x, w->f_);
    if (status)
        return status;
    status = gslref_eval_df(w, w->x, w->f_, w->J);
    if (status)
        return status;
    for (j = 0; j < p; ++j)
    {
        double s = 0.0;
        for (i = 0; i < n; ++i)
            s += w->J[(size_t)i * p + j] * w->f_[i];
        w->g[j] = s;
    }
    scale_init(w);
    Dx = scaled_norm(p, w->diag, w->x);
    w->delta = 0.3 * fmax(1.0, Dx);
    /* nielsen_init (src/trust.c:149-173) */
    w->nu = 2;
    for (j = 0; j < p; ++j)
    {
        double norm = colnorm(w, j) / w->diag[j];
        max = fmax(max, norm);
    }
    w->mu = 1.0e-3 * max * max;
    w->avratio = 0.0;
    memset(w->acc, 0, sizeof(double) * p);
    memset(w->vel, 0, sizeof(double) * p);
    return GSLREF_SUCCESS;
}

/* gsl_multifit_nlinear_winit(_LD) (src/fdf.c:23-77) */
int gslref_winit(gslref_ws *w, const double *x, const double *wts)
{
    int i;
    w->nevalf = w->nevaldf = w->nevalfvv = 0;
    memcpy(w->x, x, sizeof(double) * w->p);
    w->niter = 0;
    if (wts)
    {
        w->sqrt_wts = w->sqrt_wts_work;
        for (i = 0; i < w->n; ++i)
            w->sqrt_wts[i] = sqrt(wts[i]);
    }
    else
        w->sqrt_wts = NULL;
    return trust_init(w);
}

/* lm_step / lm_step_LD (src/trust.c:223-292) */
static int lm_step(gslref_ws *w, double *dx)
{
    int status, i;
    status = solver_solve(w, w->mu, w->f_, w->vel, 1);
    if (status)
        return status;
    if (w->trs == 1)
    {
        double anorm, vnorm;
        status = eval_fvv(w, w->x, w->vel, w->f_, w->J, w->fvvv);
        if (status)
            return status;
        status = solver_solve(w, w->mu, w->fvvv, w->acc, 0);
        if (status)
            return status;
        anorm = dnrm2(w->p, w->acc);
        vnorm = dnrm2(w->p, w->vel);
        w->avratio = anorm / vnorm;
    }
    for (i = 0; i < w->p; ++i)
        dx[i] = w->vel[i] + 0.5 * w->acc[i];
    return GSLREF_SUCCESS;
}

/* trust_calc_rho + lm_preduction (src/trust.c:67-118; App. A.4) */
static double calc_rho(gslref_ws *w)
{
    const int n = w->n, p = w->p;
    const double normf = dnrm2(n, w->f_);
    const double normf_trial = dnrm2(n, w->f_trial);
    double u, ared, pred, norm_Dp, norm_Jp, v;
    int i, j;
    if (normf_trial >= normf)
        return -1.0;
    u = normf_trial / normf;
    ared = 1.0 - u * u;
    norm_Dp = scaled_norm(p, w->diag, w->vel);
    for (i = 0; i < n; ++i)
    {
        double s = 0.0;
        for (j = 0; j < p; ++j)
            s += w->J[(size_t)i * p + j] * w->vel[j];
        w->workn[i] = s;
    }
    norm_Jp = dnrm2(n, w->workn);
    u = norm_Jp / normf;
    v = norm_Dp / normf;
    pred = u * u + 2.0 * w->mu * v * v;
    if (pred > 0.0)
        return ared / pred;
    return -1.0;
}

/* trust_iterate_lu_LD (src/trust.c:408-549) */
int gslref_iterate(gslref_ws *w)
{
    const int n = w->n, p = w->p;
    int status, foundstep = 0, bad_steps = 0, i, j;
    double rho;

    solver_init(w); /* trs->preloop */

    while (!foundstep)
    {
        status = lm_step(w, w->dx);
        if (status == GSLREF_SUCCESS)
        {
            if (w->lu)
            {
                /* trust_trial_step_lu (src/trust.c:9-32) */
                for (i = 0; i < p; ++i)
                {
                    double dxi = w->dx[i], xi = w->x[i], xt = xi + dxi;
                    double lo = w->lu[i], up = w->lu[p + i];
                    if (xt < lo)
                        xt = xi + (dxi / fmax(fabs(dxi), w->delta) * fabs(xi - lo));
                    else if (xt > up)
                        xt = xi + (dxi / fmax(fabs(dxi), w->delta) * fabs(xi - up));
                    w->x_trial[i] = xt;
                }
            }
            else
                for (i = 0; i < p; ++i)
                    w->x_trial[i] = w->x[i] + w->dx[i];

            status = gslref_eval_f(w, w->x_trial, w->f_trial);
            if (status)
                return status;

            /* trust_eval_step (src/trust.c:126-147) */
            status = GSLREF_SUCCESS;
            if (w->trs == 1 && w->avratio > w->avmax)
                status = GSLREF_FAILURE;
            rho = calc_rho(w);
            if (rho <= 0.0)
                status = GSLREF_FAILURE;
            if (status == GSLREF_SUCCESS)
                foundstep = 1;
        }
        else
            rho = -1.0;

        if (rho > 0.75)
            w->delta *= w->factor_up;
        else if (rho < 0.25)
            w->delta /= w->factor_down;

        if (foundstep)
        {
            status = gslref_eval_df(w, w->x_trial, w->f_trial, w->J);
            if (status)
                return status;
            memcpy(w->x, w->x_trial, sizeof(double) * p);
            memcpy(w->f_, w->f_trial, sizeof(double) * n);
            for (j = 0; j < p; ++j)
            {
                double s = 0.0;
                for (i = 0; i < n; ++i)
                    s += w->J[(size_t)i * p + j] * w->f_[i];
                w->g[j] = s;
            }
            scale_update(w);
            /* nielsen_accept (src/trust.c:175-188) */
            {
                double b = 2.0 * rho - 1.0;
                b = 1.0 - b * b * b;
                w->nu = 2;
                w->mu *= fmax(0.333333333333333, b);
            }
            bad_steps = 0;
        }
        else
        {
            /* nielsen_reject (src/trust.c:190-199) */
            w->mu *= (double)w->nu;
            w->nu <<= 1;
            if (++bad_steps > 15)
                return GSLREF_ENOPROG;
        }
    }
    return GSLREF_SUCCESS;
}

/* GSL convergence.c gsl_multifit_nlinear_test (App. A.7) */
int gslref_test(const gslref_ws *w, double xtol, double gtol, double ftol, int *info)
{
    const int p = w->p;
    int i, ok = 1;
    double gnorm = 0.0, fnorm, phi;
    (void)ftol;
    *info = 0;
    for (i = 0; i < p; ++i)
    {
        double tol = xtol * xtol + xtol * fabs(w->x[i]);
        if (fabs(w->dx[i]) < tol)
            ok = 1;
        else
        {
            ok = 0;
            break;
        }
    }
    if (ok)
    {
        *info = 1;
        return GSLREF_SUCCESS;
    }
    for (i = 0; i < p; ++i)
    {
        double xi = fmax(w->x[i], 1.0);
        double t = fabs(xi * w->g[i]);
        if (t > gnorm)
            gnorm = t;
    }
    fnorm = dnrm2(w->n, w->f_);
    phi = 0.5 * fnorm * fnorm;
    if (gnorm <= gtol * fmax(phi, 1.0))
    {
        *info = 2;
        return GSLREF_SUCCESS;
    }
    return GSLREF_CONTINUE;
}

/* src/nls_fit.c:40-121 */
int gslref_driver2(gslref_ws *w, int maxiter, double xtol, double gtol, double ftol,
                   gslref_cb_t cb, void *cbp, int *info, double *chisq0, double *chisq1)
{
    int status = GSLREF_CONTINUE;
    int iter = 0;
    do
    {
        chisq0[0] = chisq1[0];
        status = gslref_iterate(w);
        w->niter++;
        chisq1[0] = ddot(w->n, w->f_, w->f_);
        if (status == GSLREF_EBADFUNC || (status == GSLREF_ENOPROG && iter == 0))
        {
            *info = status;
            return status;
        }
        ++iter;
        if (cb)
            cb(iter, cbp, w, chisq1[0]);
        status = gslref_test(w, xtol, gtol, ftol, info);
    } while (status == GSLREF_CONTINUE && iter < maxiter);

    if (status == GSLREF_ETOLF || status == GSLREF_ETOLX || status == GSLREF_ETOLG)
    {
        *info = status;
        status = GSLREF_SUCCESS;
    }
    if (iter >= maxiter && status != GSLREF_SUCCESS)
        status = GSLREF_EMAXITER;
    return status;
}

/* src/nls_utils.c:23-53 */
double gslref_det_eval_jtj(gslref_ws *w)
{
    int status = gslref_eval_f(w, w->x, w->f_);
    if (status)
        return 0.0;
    status = gslref_eval_df(w, w->x, w->f_, w->J);
    if (status)
        return 0.0;
    return gslref_det_cholesky_jtj(w->n, w->p, w->J);
}

const char *gslref_strerror(int code)
{
    switch (code)
    {
    case GSLREF_SUCCESS: return "success";
    case GSLREF_FAILURE: return "failure";
    case GSLREF_CONTINUE: return "the iteration has not converged yet";
    case GSLREF_EINVAL: return "invalid argument supplied by user";
    case GSLREF_EBADFUNC: return "problem with user-supplied function";
    case GSLREF_EMAXITER: return "exceeded max number of iterations";
    case GSLREF_ENOPROG: return "iteration is not making progress towards solution";
    case GSLREF_ETOLF: return "cannot reach the specified tolerance in F";
    case GSLREF_ETOLX: return "cannot reach the specified tolerance in X";
    case GSLREF_ETOLG: return "cannot reach the specified tolerance in gradient";
    default: return "unknown error code";
    }
}
