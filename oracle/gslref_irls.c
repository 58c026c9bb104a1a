/*
 * gslref_irls.c -- ORACLE (test infrastructure, never shipped).
 * Robust-loss IRLS behind gsl_nls(loss=...):
 *   psi / psi' families        src/nls_irls.c:10-341 (index = loss_config$rho, R/nls.R:663)
 *   test_delta_irls            src/nls_irls.c:343-362
 *   gsl_multifit_nlinear_rho_driver   src/nls_irls.c:412-546
 *   gsl_median / gsl_mad       src/nls_utils.c:162-217
 *   hat_values / cooks_d       src/nls_utils.c:88-150
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include "gslref_internal.h"

#define SQRT_EPS 1.4901161193847656e-08

/* ---- loss 1: Huber (nls_irls.c:10-20) ---- */
static double huber_psi(double r, double c) { return r <= -c ? -c : (r < c ? r : c); }
static double huber_dpsi(double r, double c) { return fabs(r) >= c ? 0.0 : 1.0; }

/* ---- loss 2: Barron family, cc = (alpha, c) (nls_irls.c:22-56) ---- */
static double barron_psi(double r, const double *cc)
{
    const double alpha = cc[0], c2 = cc[1] * cc[1], z = (r * r) / c2;
    if (fabs(alpha - 2.0) < SQRT_EPS)
        return r / c2;
    if (fabs(alpha) < SQRT_EPS)
        return 2.0 * r / (r * r + 2 * c2);
    if (alpha > -1e8)
        return r / c2 * pow((z / fabs(alpha - 2.0) + 1), 0.5 * alpha - 1.0);
    return r / c2 * exp(-0.5 * z);
}
static double barron_dpsi(double r, const double *cc)
{
    const double alpha = cc[0], c2 = cc[1] * cc[1], r2 = r * r;
    if (fabs(alpha - 2.0) < SQRT_EPS)
        return 1.0 / c2;
    if (fabs(alpha) < SQRT_EPS)
        return -2. * (r2 - 2. * c2) / ((2. * c2 + r2) * (2. * c2 + r2));
    if (alpha > -1e8)
    {
        const double den = r2 - (alpha - 2.) * c2;
        return (alpha - 2.) * ((alpha - 2.) * c2 - (alpha - 1.) * r2) *
               pow(1. - r2 / ((alpha - 2.) * c2), 0.5 * alpha) / (den * den);
    }
    return exp(-r2 / (2. * c2)) * (c2 - r2) / (c2 * c2);
}

/* ---- loss 3: Tukey bisquare (nls_irls.c:58-82) ---- */
static double bisq_psi(double r, double c)
{
    double a, u;
    if (fabs(r) > c)
        return 0.;
    a = r / c;
    u = 1. - a * a;
    return r * u * u;
}
static double bisq_dpsi(double r, double c)
{
    double t2;
    if (fabs(r) > c)
        return 0.;
    r /= c;
    t2 = r * r;
    return (1. - t2) * (1 - 5 * t2);
}

/* ---- loss 4: Welsh / Gauss weight (nls_irls.c:84-110) ---- */
static double welsh_psi(double r, double c)
{
    const double a = r / c;
    return fabs(a) > 37.7 ? 0. : r * exp(-(a * a) / 2);
}
static double welsh_dpsi(double r, double c)
{
    double a2;
    r /= c;
    if (fabs(r) > 37.7)
        return 0.;
    a2 = r * r;
    return exp(-a2 / 2) * (1. - a2);
}

/* ---- loss 5: "optimal" (nls_irls.c:112-148) ---- */
static const double OPT_R1 = -1.944, OPT_R2 = 1.728, OPT_R3 = -0.312, OPT_R4 = 0.016;
static double opt_psi(double r, double c)
{
    const double ac = r / c, ax = fabs(ac);
    if (ax > 3.)
        return 0.;
    if (ax > 2.)
    {
        const double a2 = ac * ac;
        const double poly = c * ((((OPT_R4 * a2 + OPT_R3) * a2 + OPT_R2) * a2 + OPT_R1) * ac);
        return ac > 0. ? fmax(0., poly) : -fabs(poly);
    }
    return r;
}
static double opt_dpsi(double r, double c)
{
    double ax = fabs(r / c);
    if (ax > 3.)
        return 0.;
    if (ax > 2.)
    {
        ax *= ax;
        return OPT_R1 + ax * (3 * OPT_R2 + ax * (5 * OPT_R3 + ax * 7 * OPT_R4));
    }
    return 1.;
}

/* ---- loss 6: Hampel with (a,b,r) = (1.5,3.5,8) k (nls_irls.c:151-203) ---- */
static double hampel_psi(double r, double k)
{
    const double a = 1.5 * k, b = 3.5 * k, t = 8.0 * k;
    const double sgn = r < 0 ? -1. : 1., u = fabs(r);
    if (u <= a)
        return r;
    if (u <= b)
        return sgn * a;
    if (u <= t)
        return sgn * a * (t - u) / (t - b);
    return 0.;
}
static double hampel_dpsi(double r, double k)
{
    const double a = 1.5 * k, b = 3.5 * k, t = 8.0 * k, u = fabs(r);
    if (u <= a)
        return 1.;
    if (u <= b)
        return 0.;
    if (u <= t)
        return a / (b - t);
    return 0.;
}

/* ---- loss 7: GGW, cc = (a, b, c) (nls_irls.c:205-239) ---- */
static double ggw_psi(double r, const double *k)
{
    const double ax = fabs(r);
    double e;
    if (ax < k[2])
        return r;
    e = -pow(ax - k[2], k[1]) / 2 / k[0];
    return e < -708.4 ? 0. : r * exp(e);
}
static double ggw_dpsi(double r, const double *k)
{
    const double ax = fabs(r);
    double a, b, c, e;
    if (ax < k[2])
        return 1.;
    a = 2 * k[0];
    b = k[1];
    c = k[2];
    e = -pow(ax - c, b) / a;
    return e < -708.4 ? 0. : exp(e) * (1 - b / a * ax * pow(ax - c, b - 1));
}

/* ---- loss 8: LQQ, cc = (b, c, s) (nls_irls.c:241-291) ---- */
static double lqq_psi(double r, const double *k)
{
    const double ax = fabs(r);
    double k01, s5, s6;
    if (ax <= k[1])
        return r;
    k01 = k[0] + k[1];
    if (ax <= k01)
        return (double)(r > 0 ? 1 : (r < 0 ? -1 : 0)) * (ax - k[2] * pow(ax - k[1], 2.) / k[0] / 2.);
    s5 = k[2] - 1.;
    s6 = -2 * k01 + k[0] * k[2];
    if (ax < k01 - s6 / s5)
        return (double)(r > 0 ? 1 : -1) *
               (-s6 / 2. - pow(s5, 2.) / s6 * (pow(ax - k01, 2.) / 2. + s6 / s5 * (ax - k01)));
    return 0.;
}
static double lqq_dpsi(double r, const double *k)
{
    const double ax = fabs(r);
    double k01, s5, a;
    if (ax <= k[1])
        return 1.;
    k01 = k[0] + k[1];
    if (ax <= k01)
        return 1. - k[2] / k[0] * (ax - k[1]);
    s5 = 1. - k[2];
    a = (k[0] * k[2] - 2 * k01) / s5;
    if (ax < k01 + a)
        return -s5 * ((ax - k01) / a - 1.);
    return 0.;
}

/* dispatch (nls_irls.c:293-341); unknown index falls through to Huber */
double gslref_psi(double x, const double *cc, int i)
{
    switch (i)
    {
    case 2: return barron_psi(x, cc);
    case 3: return bisq_psi(x, cc[0]);
    case 4: return welsh_psi(x, cc[0]);
    case 5: return opt_psi(x, cc[0]);
    case 6: return hampel_psi(x, cc[0]);
    case 7: return ggw_psi(x, cc);
    case 8: return lqq_psi(x, cc);
    default: return huber_psi(x, cc[0]);
    }
}
double gslref_psip(double x, const double *cc, int i)
{
    switch (i)
    {
    case 2: return barron_dpsi(x, cc);
    case 3: return bisq_dpsi(x, cc[0]);
    case 4: return welsh_dpsi(x, cc[0]);
    case 5: return opt_dpsi(x, cc[0]);
    case 6: return hampel_dpsi(x, cc[0]);
    case 7: return ggw_dpsi(x, cc);
    case 8: return lqq_dpsi(x, cc);
    default: return huber_dpsi(x, cc[0]);
    }
}

/* R_orderVector1(indx, n, x, nalast = TRUE, decreasing = FALSE): stable ascending
 * order, NA/NaN last (src/nls_mstart.c:131, src/nls_utils.c:175). */
typedef struct { double v; int i; } ord_t;
static int ord_cmp(const void *a, const void *b)
{
    const ord_t *p = (const ord_t *)a, *q = (const ord_t *)b;
    const int pn = isnan(p->v), qn = isnan(q->v);
    if (pn || qn)
    {
        if (pn && qn)
            return p->i - q->i;
        return pn ? 1 : -1;
    }
    if (p->v < q->v) return -1;
    if (p->v > q->v) return 1;
    return p->i - q->i;
}
void gslref_order(const double *x, int n, int *order)
{
    ord_t *t = (ord_t *)malloc(sizeof(ord_t) * (n > 0 ? n : 1));
    int i;
    for (i = 0; i < n; ++i)
    {
        t[i].v = x[i];
        t[i].i = i;
    }
    qsort(t, n, sizeof(ord_t), ord_cmp);
    for (i = 0; i < n; ++i)
        order[i] = t[i].i;
    free(t);
}

/* src/nls_utils.c:162-189 */
double gslref_median(const double *data, int n)
{
    int *ord, lhs, rhs;
    double med;
    if (n == 0)
        return 0.0;
    ord = (int *)malloc(sizeof(int) * n);
    gslref_order(data, n, ord);
    lhs = (n - 1) / 2;
    rhs = n / 2;
    med = (lhs == rhs) ? data[ord[lhs]] : (data[ord[lhs]] + data[ord[rhs]]) / 2.0;
    free(ord);
    return med;
}

/* src/nls_utils.c:201-217 */
double gslref_mad(const double *data, int n)
{
    const double med = gslref_median(data, n);
    double *dev = (double *)malloc(sizeof(double) * (n > 0 ? n : 1));
    double mad;
    int i;
    for (i = 0; i < n; ++i)
        dev[i] = fabs(data[i] - med);
    mad = 1.482602218505602 * gslref_median(dev, n);
    free(dev);
    return mad;
}

/* src/nls_utils.c:88-110: h = rowsums( (J (J^T J)^-1) o J ) */
int gslref_hat_values(int n, int p, const double *J, double *h)
{
    double *A = (double *)calloc((size_t)p * p, sizeof(double));
    int i, j, k, status;
    for (k = 0; k < n; ++k)
        for (i = 0; i < p; ++i)
            for (j = 0; j <= i; ++j)
                A[i * p + j] += J[(size_t)k * p + i] * J[(size_t)k * p + j];
    status = gslref_cholesky_decomp1(p, A);
    if (!status)
        status = gslref_cholesky_invert(p, A);
    if (status)
    {
        free(A);
        return status;
    }
    for (k = 0; k < n; ++k)
    {
        double hk = 0.0;
        for (j = 0; j < p; ++j)
        {
            double s = 0.0;
            for (i = 0; i < p; ++i)
                s += J[(size_t)k * p + i] * A[i * p + j];
            hk += s * J[(size_t)k * p + j];
        }
        h[k] = hk;
    }
    free(A);
    return GSLREF_SUCCESS;
}

/* src/nls_utils.c:126-150 */
int gslref_cooks_d(int n, int p, const double *f, const double *J, double *d)
{
    double chisq = 0.0, s2;
    int i, status;
    for (i = 0; i < n; ++i)
        chisq += f[i] * f[i];
    s2 = chisq / (n - p);
    status = gslref_hat_values(n, p, J, d);
    if (status)
        return status;
    for (i = 0; i < n; ++i)
    {
        const double e = f[i], h = d[i];
        d[i] = (e * e) / (p * s2) * (h / ((1 - h) * (1 - h)));
    }
    return GSLREF_SUCCESS;
}

/* src/nls_irls.c:343-362 */
static int test_delta_irls(int p, const double *x0, const double *x1, double xtol)
{
    int status = GSLREF_CONTINUE, i;
    for (i = 0; i < p; ++i)
    {
        const double xi = x1[i], dxi = fabs(x0[i] - xi);
        if (fmin(dxi / fabs(xi), dxi) < xtol)
            status = GSLREF_SUCCESS;
        else
        {
            status = GSLREF_CONTINUE;
            break;
        }
    }
    return status;
}

/* src/nls_irls.c:412-546.  wts = user weights (length n), workn_wts = IRLS weights
 * (in/out; ends up as the final irls weights), workp = previous iterate. */
int gslref_rho_driver(gslref_ws *w, const gslref_problem *prob, const double *mpopt, double *wts,
                      double *workn_wts, double *workp, double *psi, double *psip,
                      int wgt_i, int maxiter, double xtol, double gtol, double ftol,
                      gslref_cb_t cb, void *cbp,
                      int *info, double *chisq0, double *chisq1, double *irls_sigma,
                      int *irls_iter, int *irls_status)
{
    const int n = w->n, p = w->p;
    const double *cc = prob->loss_cc;
    const int irls_maxiter = prob->control_int[14];
    const double irls_xtol = prob->control_dbl[10];
    const int has_user_wts = (prob->swts != NULL || prob->swts_mat != NULL);
    int status = GSLREF_CONTINUE, i;
    double *resid = (double *)calloc(n, sizeof(double));
    double *absr = (double *)calloc(n, sizeof(double));

    memcpy(workn_wts, wts, sizeof(double) * n);

    do
    {
        double sum_wts = 0.0;
        *irls_iter += 1;
        if (*irls_iter > 1)
        {
            memcpy(workp, w->x, sizeof(double) * p);
            gslref_winit(w, mpopt, workn_wts); /* cold restart from mpopt, App. D */
            *chisq0 = INFINITY;
        }
        else
            memcpy(workp, mpopt, sizeof(double) * p);

        status = gslref_driver2(w, maxiter, xtol, gtol, ftol, cb, cbp, info, chisq0, chisq1);

        if (status == GSLREF_EBADFUNC || (status == GSLREF_ENOPROG && *irls_iter == 1))
        {
            *info = status;
            goto done;
        }

        for (i = 0; i < n; ++i)
        {
            resid[i] = w->f_[i] / w->sqrt_wts[i];
            absr[i] = fabs(resid[i]);
        }
        *irls_sigma = 1.482602218505602 * gslref_median(absr, n);

        for (i = 0; i < n; ++i)
        {
            const double rs = resid[i] / *irls_sigma;
            const double ps = gslref_psi(rs, cc, wgt_i);
            const double wt = fmax(ps / rs, DBL_EPSILON);
            workn_wts[i] = wt;
            psi[i] = ps;
            psip[i] = gslref_psip(rs, cc, wgt_i);
            sum_wts += wt;
        }
        for (i = 0; i < n; ++i)
            workn_wts[i] *= n / sum_wts;
        if (has_user_wts)
            for (i = 0; i < n; ++i)
                workn_wts[i] = wts[i] * workn_wts[i];

        *irls_status = test_delta_irls(p, workp, w->x, irls_xtol);
        if (*irls_status == GSLREF_SUCCESS)
        {
            *info = status;
            goto done;
        }
    } while (*irls_status == GSLREF_CONTINUE && *irls_iter < irls_maxiter);

    if (*irls_iter >= irls_maxiter && *irls_status != GSLREF_SUCCESS)
    {
        *irls_status = GSLREF_EMAXITER;
        *info = GSLREF_EMAXITER;
        status = GSLREF_EMAXITER;
    }
done:
    free(resid);
    free(absr);
    return status;
}
