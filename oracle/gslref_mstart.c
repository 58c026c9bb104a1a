/*
 * gslref_mstart.c -- ORACLE (test infrastructure, never shipped).
 *   GSL qrng sobol / halton  (un-vendored GSL; selected src/nls.c:277-280, drawn
 *                             src/nls_mstart.c:48, re-initialised src/nls.c:447)
 *   gsl_multistart_driver    src/nls_mstart.c:24-350
 * The major-iteration loop and stopping rule (src/nls.c:372-399) live in
 * gslref_nls.c next to the rest of C_nls_internal.
 *
 * Sobol: Antonov-Saleev Gray-code generator with the Bratley-Fox (ACM TOMS 659)
 * direction numbers, 40 dimensions, 30 bits, first returned point = 0.5 in every
 * dimension (GSL qrng/sobol.c).  PARITY: dimensions 1-2 are pinned against an
 * independent implementation (tests/golden/sobol_d2.json); the direction-number
 * table for dimensions 3..40 is restated from the published algorithm and is
 * "parity unpinned".
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include "gslref_internal.h"
#include "gslref_mstart.h"

#define SOBOL_MAX_DIM 40
#define SOBOL_BITS 30

static const int sobol_poly[SOBOL_MAX_DIM] = {
    1, 3, 7, 11, 13, 19, 25, 37, 59, 47, 61, 55, 41, 67, 97, 91, 109, 103, 115, 131,
    193, 137, 145, 143, 241, 157, 185, 167, 229, 171, 213, 191, 253, 203, 211, 239, 247, 285, 369, 299};
static const int sobol_deg[SOBOL_MAX_DIM] = {
    0, 1, 2, 3, 3, 4, 4, 5, 5, 5, 5, 5, 5, 6, 6, 6, 6, 6, 6, 7,
    7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 8, 8, 8};
static const int sobol_vinit[8][SOBOL_MAX_DIM] = {
    {0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
     1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1},
    {0, 0, 1, 3, 1, 3, 1, 3, 3, 1, 3, 1, 3, 1, 3, 1, 1, 3, 1, 3,
     1, 3, 1, 3, 3, 1, 3, 1, 3, 1, 3, 1, 1, 3, 1, 3, 1, 3, 1, 3},
    {0, 0, 0, 7, 5, 1, 3, 3, 7, 5, 5, 7, 7, 1, 3, 3, 7, 5, 1, 1,
     5, 3, 3, 1, 7, 5, 1, 3, 3, 7, 5, 1, 1, 5, 7, 7, 5, 1, 3, 3},
    {0, 0, 0, 0, 0, 1, 7, 9, 13, 11, 1, 3, 7, 9, 5, 13, 13, 11, 3, 15,
     5, 3, 15, 7, 9, 13, 9, 1, 11, 7, 5, 15, 1, 15, 11, 5, 3, 1, 7, 9},
    {0, 0, 0, 0, 0, 0, 0, 9, 3, 27, 15, 29, 21, 23, 19, 11, 25, 7, 13, 17,
     1, 25, 29, 3, 31, 11, 5, 23, 27, 19, 21, 5, 1, 17, 13, 7, 15, 9, 31, 9},
    {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 37, 33, 7, 5, 11, 39, 63,
     27, 17, 15, 23, 29, 3, 21, 13, 31, 25, 9, 49, 33, 19, 29, 11, 19, 27, 15, 25},
    {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 13,
     33, 115, 41, 79, 17, 29, 119, 75, 73, 105, 7, 59, 65, 21, 3, 113, 61, 89, 45, 107},
    {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
     0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 7, 23, 39}};

struct gslref_qrng
{
    int dim;
    int halton;
    unsigned int count;
    int vdir[SOBOL_BITS][SOBOL_MAX_DIM];
    int num[SOBOL_MAX_DIM];
    double denom_inv;
    int *primes;
};

static void sobol_setup(gslref_qrng *q)
{
    int d, j, k, ell;
    for (k = 0; k < SOBOL_BITS; ++k)
        q->vdir[k][0] = 1;
    for (d = 1; d < q->dim; ++d)
    {
        const int deg = sobol_deg[d];
        int incl[8], pp = sobol_poly[d];
        for (k = deg - 1; k >= 0; --k)
        {
            incl[k] = (pp % 2) == 1;
            pp /= 2;
        }
        for (j = 0; j < deg; ++j)
            q->vdir[j][d] = sobol_vinit[j][d];
        for (j = deg; j < SOBOL_BITS; ++j)
        {
            int nv = q->vdir[j - deg][d];
            ell = 1;
            for (k = 0; k < deg; ++k)
            {
                ell *= 2;
                if (incl[k])
                    nv ^= (ell * q->vdir[j - k - 1][d]);
            }
            q->vdir[j][d] = nv;
        }
    }
    ell = 1;
    for (j = SOBOL_BITS - 2; j >= 0; --j)
    {
        ell *= 2;
        for (d = 0; d < q->dim; ++d)
            q->vdir[j][d] *= ell;
    }
    q->denom_inv = 1.0 / (2.0 * ell);
}

gslref_qrng *gslref_qrng_alloc(int dim)
{
    gslref_qrng *q = (gslref_qrng *)calloc(1, sizeof(gslref_qrng));
    q->dim = dim;
    q->halton = dim > SOBOL_MAX_DIM; /* src/nls.c:277-280: p < 41 -> sobol */
    if (q->halton)
    {
        int c = 0, cand = 2;
        q->primes = (int *)malloc(sizeof(int) * dim);
        while (c < dim)
        {
            int ok = 1, t;
            for (t = 2; t * t <= cand; ++t)
                if (cand % t == 0)
                {
                    ok = 0;
                    break;
                }
            if (ok)
                q->primes[c++] = cand;
            ++cand;
        }
    }
    else
        sobol_setup(q);
    gslref_qrng_init(q);
    return q;
}

void gslref_qrng_init(gslref_qrng *q)
{
    q->count = 0;
    memset(q->num, 0, sizeof(q->num));
}

void gslref_qrng_free(gslref_qrng *q)
{
    if (q)
    {
        free(q->primes);
        free(q);
    }
}

void gslref_qrng_get(gslref_qrng *q, double *v)
{
    int d;
    if (q->halton)
    {
        q->count++;
        for (d = 0; d < q->dim; ++d)
        {
            /* radical inverse of count in base primes[d] */
            unsigned int t = q->count;
            const double binv = 1.0 / q->primes[d];
            double r = 0.0, f = binv;
            while (t > 0)
            {
                r += f * (double)(t % q->primes[d]);
                t /= q->primes[d];
                f *= binv;
            }
            v[d] = r;
        }
    }
    else
    {
        int ell = 0;
        unsigned int c = q->count;
        for (;;)
        {
            ++ell;
            if ((c % 2) == 1)
                c /= 2;
            else
                break;
        }
        for (d = 0; d < q->dim; ++d)
        {
            q->num[d] ^= q->vdir[ell - 1][d];
            v[d] = q->num[d] * q->denom_inv;
        }
        q->count++;
    }
}

int gslref_sobol(int dim, int skip, int npts, double *out)
{
    gslref_qrng *q;
    double *tmp;
    int i;
    if (dim < 1 || dim > SOBOL_MAX_DIM)
        return GSLREF_EINVAL;
    q = gslref_qrng_alloc(dim);
    tmp = (double *)malloc(sizeof(double) * dim);
    for (i = 0; i < skip; ++i)
        gslref_qrng_get(q, tmp);
    for (i = 0; i < npts; ++i)
        gslref_qrng_get(q, out + (size_t)i * dim);
    free(tmp);
    gslref_qrng_free(q);
    return GSLREF_SUCCESS;
}

int gslref_halton(int dim, int skip, int npts, double *out)
{
    gslref_qrng *q = gslref_qrng_alloc(dim);
    double *tmp = (double *)malloc(sizeof(double) * dim);
    int i;
    q->halton = 1;
    if (!q->primes)
    {
        /* force halton for small dims: rebuild primes */
        gslref_qrng_free(q);
        q = (gslref_qrng *)calloc(1, sizeof(gslref_qrng));
        q->dim = dim;
        q->halton = 1;
        q->primes = (int *)malloc(sizeof(int) * dim);
        {
            int c = 0, cand = 2;
            while (c < dim)
            {
                int ok = 1, t;
                for (t = 2; t * t <= cand; ++t)
                    if (cand % t == 0)
                    {
                        ok = 0;
                        break;
                    }
                if (ok)
                    q->primes[c++] = cand;
                ++cand;
            }
        }
    }
    for (i = 0; i < skip; ++i)
        gslref_qrng_get(q, tmp);
    for (i = 0; i < npts; ++i)
        gslref_qrng_get(q, out + (size_t)i * dim);
    free(tmp);
    gslref_qrng_free(q);
    return GSLREF_SUCCESS;
}

/* ------------------------------------------------------------------ */
/* (re-)initialise the workspace at workp the way src/nls_mstart.c:84-89 / :246-251 does */
static void ms_init_ws(gslref_mstate *m, const double *xstart, int use_weights)
{
    if (m->w->Lw)
        gslref_winit(m->w, xstart, m->wts);
    else if (use_weights || m->has_swts)
        gslref_winit(m->w, xstart, m->wts);
    else
        gslref_winit(m->w, xstart, NULL);
}

/* one major iteration: src/nls_mstart.c:24-350 */
void gslref_multistart_driver(gslref_mstate *m, double xtol, double ftol, int use_weights)
{
    gslref_ws *w = m->w;
    const int p = w->p, N = m->n;
    int minfo, nn, k;
    double kd, l0, l1, diagmin, det_jtj;
    double mchisq0 = INFINITY, mchisq1 = INFINITY;
    double *mssr = m->mssr;

    /* sample initial points + concentration (:42-128) */
    for (nn = 0; nn < N; ++nn)
    {
        mssr[nn] = NAN; /* NA_REAL */
        if (m->ntix[nn] == 0)
        {
            gslref_qrng_get(m->q, m->qmp);
            for (k = 0; k < p; ++k)
            {
                l0 = m->start[2 * k];
                l1 = m->start[2 * k + 1];
                if (l1 > l0)
                {
                    double qk;
                    kd = m->diag[k];
                    qk = l0 + (l1 - l0) * m->qmp[k];
                    m->qmp[k] = qk;
                    if (l0 > 0.0)
                        m->mx[(size_t)nn * p + k] = (pow(qk - l0 + 1.0, kd) - 1.0) / kd + l0;
                    else if (l1 < 0.0)
                        m->mx[(size_t)nn * p + k] = -(pow(-qk + l1 + 1.0, kd) - 1.0) / kd + l1;
                    else if (qk > 0.0)
                        m->mx[(size_t)nn * p + k] = (pow(qk + 1.0, kd) - 1.0) / kd;
                    else
                        m->mx[(size_t)nn * p + k] = -(pow(-qk + 1.0, kd) - 1.0) / kd;
                }
                else
                    m->mx[(size_t)nn * p + k] = l0;
            }
        }
        memcpy(w->x, m->mx + (size_t)nn * p, sizeof(double) * p);
        det_jtj = gslref_det_eval_jtj(w);

        if (det_jtj > m->dtol)
        {
            ms_init_ws(m, m->mx + (size_t)nn * p, use_weights);
            gslref_driver2(w, m->p, xtol, 1e-3, ftol, NULL, NULL, &minfo, &mchisq0, &mchisq1);
            det_jtj = gslref_det_cholesky_jtj(w->n, p, w->J);
            if (mchisq1 < INFINITY)
            {
                if (det_jtj > m->dtol)
                {
                    memcpy(m->mx + (size_t)nn * p, w->x, sizeof(double) * p);
                    mssr[nn] = mchisq1;
                    if (mchisq1 < 0.99 * fmin(m->mssropt[0], m->mssropt[1]))
                    {
                        m->mssropt[0] = mchisq1;
                        m->ssrconv[0] = mchisq0 - mchisq1;
                        memcpy(m->mpopt, w->x, sizeof(double) * p);
                    }
                }
                else if (mchisq1 < 0.99 * fmin(m->mssropt[0], m->mssropt[1]))
                {
                    m->mssropt[1] = mchisq1;
                    m->ssrconv[1] = mchisq0 - mchisq1;
                    memcpy(m->mpopt1, w->x, sizeof(double) * p);
                }
            }
        }
        else if (!(m->mssropt[0] < INFINITY) && det_jtj > DBL_EPSILON)
        {
            int i;
            mchisq1 = 0.0;
            for (i = 0; i < w->n; ++i)
                mchisq1 += w->f_[i] * w->f_[i];
            if (mchisq1 < 0.99 * m->mssropt[1])
            {
                m->mssropt[1] = mchisq1;
                m->ssrconv[1] = mchisq0 - mchisq1;
                memcpy(m->mpopt1, w->x, sizeof(double) * p);
            }
        }
    }

    /* reduce sample points (:131-138) */
    gslref_order(mssr, N, m->mssr_order);
    for (nn = 0; nn < N; ++nn)
    {
        if (nn < m->qtop && !isnan(mssr[m->mssr_order[nn]]))
            m->ntix[m->mssr_order[nn]] += 1;
        else
            m->ntix[m->mssr_order[nn]] = 0;
    }

    /* dynamic lower/upper limits (:141-233) */
    if (!m->all_start)
    {
        double pk, pmin = 0.0, pmax = 1.0;
        double mssr_diff = mssr[m->mssr_order[0]];
        if (!isnan(mssr_diff))
        {
            for (nn = N - 1; nn > 0; --nn)
                if (!isnan(mssr[m->mssr_order[nn]]))
                {
                    mssr_diff -= mssr[m->mssr_order[nn]];
                    break;
                }
        }
        if (isnan(mssr_diff) || fabs(mssr_diff) < 1e-5)
            for (k = 0; k < p; ++k)
                m->luchange[k] += 1;

        for (k = 0; k < p; ++k)
        {
            int luchange_add = 0;
            if (m->mssropt[0] < INFINITY)
            {
                const double *best = (m->mssropt[1] < m->mssropt[0]) ? m->mpopt1 : m->mpopt;
                pmin = best[k];
                pmax = best[k];
            }
            for (nn = 0; nn < m->qtop; ++nn)
            {
                const int o = m->mssr_order[nn];
                if (m->ntix[o] > 0 && mssr[o] < 1.25 * m->mssropt[0])
                {
                    pk = m->mx[(size_t)o * p + k];
                    pmin = (pk < pmin) ? pk : pmin;
                    pmax = (pk > pmax) ? pk : pmax;
                }
            }
            l0 = m->start[2 * k];
            l1 = m->start[2 * k + 1];
            if (!m->has_start[2 * k])
            {
                if (pmin < 0.9 * l0 || m->luchange[k] > 4)
                {
                    m->start[2 * k] = l0 < 0 ? fmax(l0 / pow(-1e-5 * (l0 - 1.0), 0.1) - 1.0, -1.0E5) : -0.1;
                    if (w->lu)
                        m->start[2 * k] = fmax(m->start[2 * k], w->lu[k]);
                    m->maxlims[2 * k] = fmin(m->start[2 * k], m->maxlims[2 * k]);
                    luchange_add = -1;
                }
                else if (pmin > 0.2 * l0)
                {
                    m->start[2 * k] = fmin(l0 / pow(-0.05 * (l0 - 1.0), 0.05), -0.01);
                    if (w->lu)
                        m->start[2 * k] = fmax(m->start[2 * k], w->lu[k]);
                    luchange_add = (m->mssropt[0] < INFINITY) ? -1 : 1;
                }
                else
                    luchange_add = 1;
            }
            if (!m->has_start[2 * k + 1])
            {
                if (pmax > 0.9 * l1 || m->luchange[k] > 4)
                {
                    m->start[2 * k + 1] = fmin(l1 / pow(1e-5 * (l1 + 1.0), 0.1) + 1.0, 1.0E5);
                    if (w->lu)
                        m->start[2 * k + 1] = fmin(m->start[2 * k + 1], w->lu[p + k]);
                    m->maxlims[2 * k + 1] = fmax(m->start[2 * k + 1], m->maxlims[2 * k + 1]);
                    luchange_add = -1;
                }
                else if (pmax < 0.2 * l1)
                {
                    m->start[2 * k + 1] = fmax(l1 / pow(0.05 * (l1 + 1.0), 0.05), 0.1);
                    if (w->lu)
                        m->start[2 * k + 1] = fmin(m->start[2 * k + 1], w->lu[p + k]);
                    luchange_add = (m->mssropt[0] < INFINITY) ? -1 : 1;
                }
                else
                    luchange_add = 1;
            }
            if (luchange_add)
                m->luchange[k] = (luchange_add > 0) ? m->luchange[k] + 1 : 0;
        }
    }

    /* local optimisation stage (:236-349) */
    for (nn = 0; nn < N; ++nn)
    {
        if (m->ntix[nn] >= m->s)
        {
            m->ntix[nn] = 0;
            m->nwsp += 1;
            if (m->nsp == 0 || mssr[nn] < (1 + m->tol) * m->mssropt[0])
            {
                ms_init_ws(m, m->mx + (size_t)nn * p, use_weights);
                mchisq1 = mssr[nn];
                gslref_driver2(w, m->niter, xtol, 1e-3, ftol, NULL, NULL, &minfo, &mchisq0, &mchisq1);
                det_jtj = gslref_det_cholesky_jtj(w->n, p, w->J);

                if (mchisq1 < INFINITY && (m->nsp == 0 || mchisq1 < 0.99 * m->mssropt[0]) &&
                    (det_jtj > m->dtol || mchisq1 < (2 * ftol)))
                {
                    int reject = 0;
                    if (m->rejectscl > 0)
                    {
                        for (k = 0; k < p; ++k)
                        {
                            const double xk = w->x[k];
                            if (m->all_start)
                                reject += (xk > fmax(m->maxlims[2 * k + 1], 1.0) || xk < fmin(m->maxlims[2 * k], -1.0));
                            else
                                reject += (xk > fmax(pow(m->maxlims[2 * k + 1], m->rejectscl), 1.0) ||
                                           xk < fmin(-pow(-m->maxlims[2 * k], m->rejectscl), -1.0));
                            if (reject > 0)
                                break;
                        }
                        if (!m->all_start)
                            m->rejectscl += 0.05;
                    }
                    if (!reject)
                    {
                        m->mssropt[0] = mchisq1;
                        m->ssrconv[0] = mchisq0 - mchisq1;
                        memcpy(m->mpopt, w->x, sizeof(double) * p);
                        m->nsp += 1;
                        m->nwsp = 0;
                        if (m->rejectscl > 0)
                            m->rejectscl = 1.25;
                        if (m->all_start)
                        {
                            diagmin = w->diag[0];
                            for (k = 1; k < p; ++k)
                                diagmin = fmin(diagmin, w->diag[k]);
                            for (k = 0; k < p; ++k)
                                m->diag[k] = pow(diagmin / w->diag[k], 0.25);
                        }
                    }
                }
                else if (mchisq1 < 0.99 * fmin(m->mssropt[0], m->mssropt[1]))
                {
                    m->mssropt[1] = mchisq1;
                    m->ssrconv[1] = mchisq0 - mchisq1;
                    memcpy(m->mpopt1, w->x, sizeof(double) * p);
                }
            }
        }
    }
}
