/*
 * gslref_multilarge.c -- ORACLE (test infrastructure, never shipped).
 * Restates what runs behind .Call(C_nls_large) (src/nls_large.c:77-424):
 * gsl_multilarge_nlinear "trust" on the normal equations with the
 * Steihaug-Toint CG subproblem ("cgst", control_int[2] == 5) and the LM
 * subproblem (control_int[2] == 0), iterated by
 * gsl_multilarge_nlinear_driver2 (src/nls_fit.c:153-224).
 * GSL multilarge_nlinear {trust.c, cgst.c, lm.c, cholesky.c, scaling.c,
 * convergence.c, covar} are un-vendored upstream code restated from their
 * published algorithm.  The user callback contract is
 * gsl_df_large's (src/nls_large.c:474-653): v = J u / J^T u and/or J^T J.
 * Quirk preserved: weights scale f only (GSL multilarge eval_f), the reference's
 * callback never scales J (src/nls_large.c:629-633).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include "gslref_internal.h"

typedef struct
{
    int n, p;
    int trs, scale;
    double factor_up, factor_down;
    gslref_f_t f;
    gslref_dfl_t df;
    void *params;
    long nevalf, nevaldfu, nevaldf2;
    double *x, *f_, *g, *dx, *JTJ, *sqrt_wts;
    double *diag, *x_trial, *f_trial, *workp, *workn;
    double *z, *r, *d, *vel, *work_JTJ, *rhs;
    int *perm;
    double delta, mu;
    long nu;
    int niter;
    double cgtol;
    long cgmaxit;
} lws;

static double nrm2(int n, const double *x)
{
    double scale = 0.0, ssq = 1.0;
    int i;
    for (i = 0; i < n; ++i)
        if (x[i] != 0.0)
        {
            double a = fabs(x[i]);
            if (isinf(a))
                return INFINITY;
            if (scale < a)
            {
                ssq = 1.0 + ssq * (scale / a) * (scale / a);
                scale = a;
            }
            else
                ssq += (a / scale) * (a / scale);
        }
    return scale * sqrt(ssq);
}

static int l_eval_f(lws *w, const double *x, double *y)
{
    int s = w->f(x, w->params, y), i;
    ++w->nevalf;
    if (w->sqrt_wts)
        for (i = 0; i < w->n; ++i)
            y[i] *= w->sqrt_wts[i];
    return s;
}

static int l_eval_df(lws *w, int trans, const double *x, const double *u, double *v, double *JTJ)
{
    int s = w->df(trans, x, u, w->params, v, JTJ);
    if (v)
        ++w->nevaldfu;
    if (JTJ)
        ++w->nevaldf2;
    return s;
}

static void l_scale(lws *w, int init)
{
    int j;
    for (j = 0; j < w->p; ++j)
    {
        double norm;
        if (w->scale == 1)
        {
            if (init)
                w->diag[j] = 1.0;
            continue;
        }
        norm = sqrt(w->JTJ[j * w->p + j]);
        if (norm == 0.0)
            norm = 1.0;
        if (init || w->scale == 2)
            w->diag[j] = norm;
        else
            w->diag[j] = fmax(w->diag[j], norm);
    }
}

static double cgst_tau(int p, const double *z, const double *d, double delta)
{
    double norm_p = nrm2(p, z), norm_d = nrm2(p, d), u = 0.0, t1, t2;
    int i;
    for (i = 0; i < p; ++i)
        u += z[i] * d[i];
    t1 = u / (norm_d * norm_d);
    t2 = t1 * u + (delta + norm_p) * (delta - norm_p);
    return -t1 + sqrt(t2) / norm_d;
}

/* GSL multilarge_nlinear/cgst.c cgst_step (App. A.6) */
static int cgst_step(lws *w, double *dx)
{
    const int p = w->p, n = w->n;
    double alpha, beta, u, norm_Jd, norm_r, norm_rp1, norm_g;
    long it;
    int i, status;
    for (i = 0; i < p; ++i)
    {
        w->z[i] = 0.0;
        w->r[i] = -w->g[i] / w->diag[i];
        w->d[i] = w->r[i];
        w->workp[i] = w->g[i] / w->diag[i];
    }
    norm_g = nrm2(p, w->workp);
    for (it = 0; it < w->cgmaxit; ++it)
    {
        for (i = 0; i < p; ++i)
            w->workp[i] = w->d[i] / w->diag[i];
        status = l_eval_df(w, 0, w->x, w->workp, w->workn, NULL);
        if (status)
            return status;
        norm_Jd = nrm2(n, w->workn);
        if (norm_Jd == 0.0)
        {
            double tau = cgst_tau(p, w->z, w->d, w->delta);
            for (i = 0; i < p; ++i)
                dx[i] = (w->z[i] + tau * w->d[i]) / w->diag[i];
            return GSLREF_SUCCESS;
        }
        norm_r = nrm2(p, w->r);
        u = norm_r / norm_Jd;
        alpha = u * u;
        for (i = 0; i < p; ++i)
            w->workp[i] = w->z[i] + alpha * w->d[i];
        u = nrm2(p, w->workp);
        if (u >= w->delta)
        {
            double tau = cgst_tau(p, w->z, w->d, w->delta);
            for (i = 0; i < p; ++i)
                dx[i] = (w->z[i] + tau * w->d[i]) / w->diag[i];
            return GSLREF_SUCCESS;
        }
        memcpy(w->z, w->workp, sizeof(double) * p);
        status = l_eval_df(w, 1, w->x, w->workn, w->workp, NULL);
        if (status)
            return status;
        for (i = 0; i < p; ++i)
            w->r[i] -= alpha * (w->workp[i] / w->diag[i]);
        norm_rp1 = nrm2(p, w->r);
        u = norm_rp1 / norm_g;
        if (u < w->cgtol)
        {
            for (i = 0; i < p; ++i)
                dx[i] = w->z[i] / w->diag[i];
            return GSLREF_SUCCESS;
        }
        u = norm_rp1 / norm_r;
        beta = u * u;
        for (i = 0; i < p; ++i)
            w->d[i] = w->r[i] + beta * w->d[i];
    }
    for (i = 0; i < p; ++i)
        dx[i] = w->z[i] / w->diag[i];
    return GSLREF_EMAXITER;
}

/* multilarge lm.c: normal equations + mcholesky */
static int llm_step(lws *w, double *dx)
{
    const int p = w->p;
    int i, j;
    for (i = 0; i < p; ++i)
        for (j = 0; j <= i; ++j)
            w->work_JTJ[i * p + j] = w->JTJ[i * p + j];
    for (i = 0; i < p; ++i)
        w->work_JTJ[i * p + i] += w->mu * w->diag[i] * w->diag[i];
    gslref_mcholesky_decomp(p, w->work_JTJ, w->perm);
    for (i = 0; i < p; ++i)
        w->rhs[i] = -w->g[i];
    gslref_mcholesky_solve(p, w->work_JTJ, w->perm, w->rhs, w->vel);
    memcpy(dx, w->vel, sizeof(double) * p);
    return GSLREF_SUCCESS;
}

static double l_calc_rho(lws *w, const double *dx)
{
    const int n = w->n, p = w->p;
    const double normf = nrm2(n, w->f_), normf_trial = nrm2(n, w->f_trial);
    double u, ared, pred;
    int i, j;
    if (normf_trial >= normf)
        return -1.0;
    u = normf_trial / normf;
    ared = 1.0 - u * u;
    if (w->trs == 5)
    {
        /* quadratic model: pred = -(||J dx||/||f||)^2 - 2 g.dx/||f||^2 */
        double normu, gTdx = 0.0;
        if (l_eval_df(w, 0, w->x, dx, w->workn, NULL))
            return -1.0;
        for (i = 0; i < n; ++i)
            w->workn[i] /= normf;
        normu = nrm2(n, w->workn);
        for (i = 0; i < p; ++i)
            gTdx += (w->g[i] / normf) * dx[i];
        gTdx /= normf;
        pred = -normu * normu - 2.0 * gTdx;
    }
    else
    {
        double norm_Dp = 0.0, vJv = 0.0, v;
        for (i = 0; i < p; ++i)
        {
            double t = w->diag[i] * w->vel[i];
            double s = 0.0;
            norm_Dp += t * t;
            for (j = 0; j < p; ++j)
                s += (j <= i ? w->JTJ[i * p + j] : w->JTJ[j * p + i]) * w->vel[j];
            vJv += s * w->vel[i];
        }
        norm_Dp = sqrt(norm_Dp);
        u = sqrt(vJv) / normf;
        v = norm_Dp / normf;
        pred = u * u + 2.0 * w->mu * v * v;
    }
    return pred > 0.0 ? ared / pred : -1.0;
}

static int l_init(lws *w)
{
    const int p = w->p;
    int status, j;
    double Dx = 0.0, max = -1.0;
    status = l_eval_f(w, w->x, w->f_);
    if (status)
        return status;
    status = l_eval_df(w, 1, w->x, w->f_, w->g, w->JTJ);
    if (status)
        return status;
    l_scale(w, 1);
    for (j = 0; j < p; ++j)
        Dx += (w->diag[j] * w->x[j]) * (w->diag[j] * w->x[j]);
    w->delta = 0.3 * fmax(1.0, sqrt(Dx));
    w->nu = 2;
    for (j = 0; j < p; ++j)
        max = fmax(max, w->JTJ[j * p + j] / (w->diag[j] * w->diag[j]));
    w->mu = 1.0e-3 * max;
    return GSLREF_SUCCESS;
}

static int l_iterate(lws *w)
{
    const int n = w->n, p = w->p;
    int status, foundstep = 0, bad_steps = 0, i;
    double rho;
    while (!foundstep)
    {
        status = (w->trs == 5) ? cgst_step(w, w->dx) : llm_step(w, w->dx);
        if (status == GSLREF_SUCCESS)
        {
            for (i = 0; i < p; ++i)
                w->x_trial[i] = w->x[i] + w->dx[i];
            status = l_eval_f(w, w->x_trial, w->f_trial);
            if (status)
                return status;
            rho = l_calc_rho(w, w->dx);
            if (rho > 0.0)
                foundstep = 1;
        }
        else if (status == GSLREF_EBADFUNC)
            return status;
        else
            rho = -1.0;

        if (rho > 0.75)
            w->delta *= w->factor_up;
        else if (rho < 0.25)
            w->delta /= w->factor_down;

        if (foundstep)
        {
            status = l_eval_df(w, 1, w->x_trial, w->f_trial, w->g, w->JTJ);
            if (status)
                return status;
            memcpy(w->x, w->x_trial, sizeof(double) * p);
            memcpy(w->f_, w->f_trial, sizeof(double) * n);
            l_scale(w, 0);
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
            w->mu *= (double)w->nu;
            w->nu <<= 1;
            if (++bad_steps > 15)
                return GSLREF_ENOPROG;
        }
    }
    return GSLREF_SUCCESS;
}

static int l_test(const lws *w, double xtol, double gtol, int *info)
{
    const int p = w->p;
    int i, ok = 1;
    double gnorm = 0.0, fnorm, phi;
    *info = 0;
    for (i = 0; i < p; ++i)
    {
        if (fabs(w->dx[i]) < xtol * xtol + xtol * fabs(w->x[i]))
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
        double t = fabs(fmax(w->x[i], 1.0) * w->g[i]);
        if (t > gnorm)
            gnorm = t;
    }
    fnorm = nrm2(w->n, w->f_);
    phi = 0.5 * fnorm * fnorm;
    if (gnorm <= gtol * fmax(phi, 1.0))
    {
        *info = 2;
        return GSLREF_SUCCESS;
    }
    return GSLREF_CONTINUE;
}

int gslref_nls_large(const gslref_large_problem *prob, gslref_large_result *res)
{
    const int n = prob->n, p = prob->p;
    const int *ci = prob->control_int;
    const double *cd = prob->control_dbl;
    const int maxiter = ci[0];
    const double xtol = cd[5], gtol = cd[7];
    lws W, *w = &W;
    int status = GSLREF_CONTINUE, iter = 0, info = GSLREF_CONTINUE, i, k, ok;
    double chisq0, chisq1;

    memset(w, 0, sizeof(*w));
    if (!(ci[2] == 0 || ci[2] == 5))
        return GSLREF_EINVAL; /* lmaccel/dogleg/ddogleg/subspace2D not restated */
    w->n = n;
    w->p = p;
    w->trs = ci[2];
    w->scale = ci[3];
    w->factor_up = cd[0];
    w->factor_down = cd[1];
    w->f = prob->f;
    w->df = prob->df;
    w->params = prob->params;
    w->cgtol = 1.0e-6;
    w->cgmaxit = n; /* GSL cgst_alloc: max_iter == 0 -> n */
    w->x = (double *)calloc(p, sizeof(double));
    w->f_ = (double *)calloc(n, sizeof(double));
    w->g = (double *)calloc(p, sizeof(double));
    w->dx = (double *)calloc(p, sizeof(double));
    w->JTJ = (double *)calloc((size_t)p * p, sizeof(double));
    w->diag = (double *)calloc(p, sizeof(double));
    w->x_trial = (double *)calloc(p, sizeof(double));
    w->f_trial = (double *)calloc(n, sizeof(double));
    w->workp = (double *)calloc(p, sizeof(double));
    w->workn = (double *)calloc(n, sizeof(double));
    w->z = (double *)calloc(p, sizeof(double));
    w->r = (double *)calloc(p, sizeof(double));
    w->d = (double *)calloc(p, sizeof(double));
    w->vel = (double *)calloc(p, sizeof(double));
    w->work_JTJ = (double *)calloc((size_t)p * p, sizeof(double));
    w->rhs = (double *)calloc(p, sizeof(double));
    w->perm = (int *)calloc(p, sizeof(int));
    if (prob->weights)
    {
        w->sqrt_wts = (double *)malloc(sizeof(double) * n);
        for (i = 0; i < n; ++i)
            w->sqrt_wts[i] = sqrt(prob->weights[i]);
    }
    memcpy(w->x, prob->start, sizeof(double) * p);

    status = l_init(w);
    chisq1 = 0.0;
    for (i = 0; i < n; ++i)
        chisq1 += w->f_[i] * w->f_[i];
    chisq0 = chisq1;
    res->chisq_init = chisq1;
    if (res->ssrtrace)
        res->ssrtrace[0] = chisq1;
    if (res->partrace)
        for (k = 0; k < p; ++k)
            res->partrace[(size_t)(maxiter + 1) * k] = w->x[k];

    /* src/nls_fit.c:153-224 */
    status = GSLREF_CONTINUE;
    do
    {
        chisq0 = chisq1;
        status = l_iterate(w);
        w->niter++;
        chisq1 = 0.0;
        for (i = 0; i < n; ++i)
            chisq1 += w->f_[i] * w->f_[i];
        if (status == GSLREF_EBADFUNC || (status == GSLREF_ENOPROG && iter == 0))
        {
            info = status;
            goto finish;
        }
        ++iter;
        if (res->ssrtrace)
            res->ssrtrace[iter] = chisq1;
        if (res->partrace)
            for (k = 0; k < p; ++k)
                res->partrace[iter + (size_t)(maxiter + 1) * k] = w->x[k];
        status = l_test(w, xtol, gtol, &info);
    } while (status == GSLREF_CONTINUE && iter < maxiter);
    if (iter >= maxiter && status != GSLREF_SUCCESS)
        status = GSLREF_EMAXITER;

finish:
    ok = (status == GSLREF_SUCCESS || status == GSLREF_EMAXITER);
    res->niter = w->niter;
    res->conv = status;
    res->info = info;
    res->ssr = chisq1;
    res->ssrtol = chisq0 - chisq1;
    res->neval[0] = (int)w->nevalf;
    res->neval[1] = (int)w->nevaldfu;
    res->neval[2] = (int)w->nevaldf2;
    res->neval[3] = 0;
    for (k = 0; k < p; ++k)
        res->par[k] = ok ? w->x[k] : prob->start[k];
    if (res->resid)
        for (i = 0; i < n; ++i)
            res->resid[i] = ok ? w->f_[i] : NAN;
    if (res->covar)
    {
        /* gsl_multilarge_nlinear_covar: (J^T J)^{-1} from the Cholesky factor of the stored J^T J */
        int bad = !ok;
        if (!bad)
        {
            double *A = (double *)malloc(sizeof(double) * (size_t)p * p);
            memcpy(A, w->JTJ, sizeof(double) * (size_t)p * p);
            if (gslref_cholesky_decomp1(p, A) || gslref_cholesky_invert(p, A))
                bad = 1;
            else
                for (i = 0; i < p; ++i)
                    for (k = 0; k < p; ++k)
                        res->covar[i + p * k] = A[i * p + k];
            free(A);
        }
        if (bad)
            for (i = 0; i < p * p; ++i)
                res->covar[i] = NAN;
    }
    free(w->x); free(w->f_); free(w->g); free(w->dx); free(w->JTJ); free(w->diag);
    free(w->x_trial); free(w->f_trial); free(w->workp); free(w->workn); free(w->z); free(w->r);
    free(w->d); free(w->vel); free(w->work_JTJ); free(w->rhs); free(w->perm); free(w->sqrt_wts);
    return status;
}
