/*
 * gslref_nls.c -- ORACLE (test infrastructure, never shipped).
 * Restates the numeric orchestration of C_nls_internal (src/nls.c:66-813):
 * control decoding (:77-152), weights (:219-242), bounds (:248-263), the
 * multi-start major loop and its stopping rule (:274-532) including the robust
 * second pass (:401-509), the final single-start solve (:539-576) or IRLS
 * (:577-596), covariance (:600-608) and result packing (:632-812).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include "gslref_internal.h"
#include "gslref_mstart.h"

typedef struct
{
    int maxiter, p;
    double *partrace, *ssrtrace;
} trace_t;

/* callback (src/nls.c:980-995) */
static void trace_cb(int iter, void *cbp, const gslref_ws *w, double chisq)
{
    trace_t *t = (trace_t *)cbp;
    int k;
    if (t->ssrtrace)
        t->ssrtrace[iter] = chisq;
    if (t->partrace)
        for (k = 0; k < t->p; ++k)
            t->partrace[iter + (size_t)(t->maxiter + 1) * k] = w->x[k];
}

static void ms_major_loop(gslref_mstate *m, const double *startptr, double xtol, double ftol, int second)
{
    const int p = m->w->p;
    int k;
    do
    {
        gslref_multistart_driver(m, xtol, ftol, second);
        m->mstarts += 1;
        if (m->mstarts > m->max)
            m->mstop = GSLREF_EMAXITER;
        if (m->nsp >= m->minsp && m->nwsp > (m->r + sqrt(m->r) * m->nsp))
            m->mstop = GSLREF_SUCCESS;
        if (!(m->mstarts % 10) && !(m->mssropt[0] < INFINITY))
        {
            m->dtol = fmax(0.5 * m->dtol, DBL_EPSILON);
            if (!(m->mstarts % 100))
                for (k = 0; k < p; ++k)
                {
                    m->start[2 * k] = startptr[2 * k];
                    m->start[2 * k + 1] = startptr[2 * k + 1];
                }
        }
    } while (m->mstop == GSLREF_CONTINUE);
}

int gslref_nls(const gslref_problem *prob, gslref_result *res)
{
    const int n = prob->n, p = prob->p;
    const int *ci = prob->control_int;
    const double *cd = prob->control_dbl;
    const int niter = ci[0];
    const int verbose = (res->partrace != NULL || res->ssrtrace != NULL);
    const int wgt_i = prob->loss_rho;
    const double xtol = cd[5], ftol = cd[6], gtol = cd[7];
    gslref_ws *w = gslref_ws_alloc(n, p);
    double *wts = (double *)malloc(sizeof(double) * n);
    double *Lw = NULL, *lu = NULL;
    double *mpopt = (double *)calloc(p, sizeof(double));
    double *workp = (double *)calloc(p, sizeof(double));
    double *workn = (double *)calloc(n, sizeof(double));
    double *psi = NULL, *psip = NULL;
    double chisq_init, chisq0, chisq1;
    int info = GSLREF_CONTINUE, status = GSLREF_FAILURE, irls_status = GSLREF_FAILURE;
    int irls_iter = 0, i, k, ok;
    double irls_delta = 0.0, irls_sigma = 1.0;
    trace_t tr;
    const int has_swts = (prob->swts != NULL || prob->swts_mat != NULL);

    /* control decoding, nls.c:94-152 */
    w->trs = (ci[2] == 1) ? 1 : 0;
    if (ci[2] > 1)
    {
        /* dogleg / ddogleg / subspace2D are not restated */
        gslref_ws_free(w);
        free(wts); free(mpopt); free(workp); free(workn);
        return GSLREF_EINVAL;
    }
    w->scale = ci[3];
    w->solver = ci[4];
    w->fdtype = ci[5] ? 1 : 0;
    w->factor_up = cd[0];
    w->factor_down = cd[1];
    w->avmax = cd[2];
    w->h_df = cd[3];
    w->h_fvv = cd[4];
    w->f = prob->f;
    w->df = prob->df;
    w->fvv = prob->fvv;
    w->params = prob->params;

    /* weights, nls.c:219-242 */
    for (i = 0; i < n; ++i)
        wts[i] = 1.0;
    if (prob->swts_mat)
    {
        const double *sw = prob->swts_mat;
        int n1, n2;
        Lw = (double *)malloc(sizeof(double) * (size_t)n * n);
        for (n1 = 0; n1 < n; ++n1)
        {
            wts[n1] = sw[n1 + (size_t)n * n1] * sw[n1 + (size_t)n * n1];
            for (n2 = 0; n2 < n; ++n2)
                Lw[(size_t)n1 * n + n2] = sw[n1 + (size_t)n * n2] / sw[n1 + (size_t)n * n1];
        }
        w->Lw = Lw;
    }
    else if (prob->swts)
        for (i = 0; i < n; ++i)
            wts[i] = prob->swts[i] * prob->swts[i];

    /* bounds, nls.c:248-263 */
    if (prob->lupars)
    {
        lu = (double *)malloc(sizeof(double) * 2 * p);
        for (k = 0; k < p; ++k)
        {
            lu[k] = isfinite(prob->lupars[2 * k]) ? prob->lupars[2 * k] : -INFINITY;
            lu[p + k] = isfinite(prob->lupars[2 * k + 1]) ? prob->lupars[2 * k + 1] : INFINITY;
        }
        w->lu = lu;
    }

    res->mstart_nsp = res->mstart_nwsp = res->mstart_iters = 0;
    res->mstart_stop = GSLREF_CONTINUE;
    res->mstart_ssropt = INFINITY;

    if (prob->mstart)
    {
        gslref_mstate m;
        const double *startptr = prob->start;
        memset(&m, 0, sizeof(m));
        m.w = w;
        m.q = gslref_qrng_alloc(p);
        /* manual workspace init, nls.c:283-294 */
        w->sqrt_wts = w->sqrt_wts_work;
        for (i = 0; i < n; ++i)
            w->sqrt_wts[i] = has_swts ? sqrt(wts[i]) : 1.0;
        m.n = ci[6];
        m.p = ci[7];
        m.qtop = ci[8];
        m.s = ci[9];
        m.niter = ci[10];
        m.max = ci[11];
        m.minsp = ci[12];
        m.wgt_i = wgt_i;
        m.all_start = 1;
        m.has_start = prob->has_start;
        m.r = cd[8];
        m.tol = cd[9];
        m.dtol = 1.0e-6;
        m.ntix = (int *)calloc(m.n, sizeof(int));
        m.qmp = (double *)calloc(p, sizeof(double));
        m.mssr_order = (int *)calloc(m.n, sizeof(int));
        m.mstop = GSLREF_CONTINUE;
        m.luchange = (int *)calloc(p, sizeof(int));
        m.rejectscl = 1.25;
        m.mssropt[0] = m.mssropt[1] = INFINITY;
        m.ssrconv[0] = m.ssrconv[1] = 1.0;
        m.start = (double *)malloc(sizeof(double) * 2 * p);
        m.maxlims = (double *)malloc(sizeof(double) * 2 * p);
        m.mssr = (double *)calloc(m.n, sizeof(double));
        m.mx = (double *)calloc((size_t)m.n * p, sizeof(double));
        m.diag = (double *)calloc(p, sizeof(double));
        m.mpopt = mpopt;
        m.mpopt1 = (double *)calloc(p, sizeof(double));
        m.wts = wts;
        m.has_swts = has_swts;
        for (k = 0; k < p; ++k)
        {
            m.start[2 * k] = m.maxlims[2 * k] = startptr[2 * k];
            m.start[2 * k + 1] = m.maxlims[2 * k + 1] = startptr[2 * k + 1];
        }
        /* sampling exponents, nls.c:356-369 */
        for (k = 0; k < p; ++k)
        {
            if (!m.has_start[2 * k] || !m.has_start[2 * k + 1])
            {
                m.diag[k] = 1.0;
                m.all_start = 0;
            }
            else
            {
                m.diag[k] = 0.75;
                if (m.start[2 * k] + xtol > m.start[2 * k + 1])
                    m.rejectscl = -1.0;
            }
        }

        ms_major_loop(&m, startptr, xtol, ftol, 0);

        /* robust second pass, nls.c:401-509 */
        if (wgt_i)
        {
            if (m.mssropt[1] < m.mssropt[0])
                memcpy(mpopt, m.mpopt1, sizeof(double) * p);
            gslref_winit(w, mpopt, wts);
            gslref_det_eval_jtj(w);
            m.mstop = gslref_cooks_d(n, p, w->f_, w->J, workn);
            if (!m.mstop)
            {
                int noutlier = 0;
                const double mad = gslref_mad(workn, n);
                const double thresh = fmin(4.0 / n, 5 * mad);
                w->sqrt_wts = w->sqrt_wts_work;
                for (i = 0; i < n; ++i)
                {
                    if (workn[i] > thresh)
                    {
                        wts[i] = 0.0;
                        w->sqrt_wts[i] = 0.0;
                        noutlier += 1;
                    }
                    else
                        w->sqrt_wts[i] = sqrt(wts[i]);
                }
                if (noutlier > 0 && noutlier < (n - p))
                {
                    gslref_qrng_init(m.q);
                    m.mstop = GSLREF_CONTINUE;
                    m.mstarts = 0;
                    m.nsp = 0;
                    m.nwsp = 0;
                    m.dtol = 1.0e-6;
                    m.rejectscl = 1.25;
                    m.mssropt[0] = m.mssropt[1] = INFINITY;
                    m.ssrconv[0] = m.ssrconv[1] = 1.0;
                    memset(m.ntix, 0, sizeof(int) * m.n);
                    memset(m.luchange, 0, sizeof(int) * p);
                    ms_major_loop(&m, startptr, xtol, ftol, 1);
                }
                for (i = 0; i < n; ++i)
                    wts[i] = 1.0;
                if (prob->swts_mat)
                    for (i = 0; i < n; ++i)
                        wts[i] = prob->swts_mat[i + (size_t)n * i] * prob->swts_mat[i + (size_t)n * i];
                else if (prob->swts)
                    for (i = 0; i < n; ++i)
                        wts[i] = prob->swts[i] * prob->swts[i];
            }
        }
        if (m.mssropt[1] < m.mssropt[0])
        {
            m.mssropt[0] = m.mssropt[1];
            m.ssrconv[0] = m.ssrconv[1];
            memcpy(mpopt, m.mpopt1, sizeof(double) * p);
        }
        /* jitter, nls.c:524-531 */
        if (m.mssropt[0] < ftol || m.ssrconv[0] < ftol)
        {
            if (lu)
                mpopt[0] = fmin(mpopt[0] + 1.0e-4, lu[p + 0]);
            else
                mpopt[0] = mpopt[0] + 1.0e-4;
        }
        res->mstart_nsp = m.nsp;
        res->mstart_nwsp = m.nwsp;
        res->mstart_iters = m.mstarts;
        res->mstart_stop = m.mstop;
        res->mstart_ssropt = m.mssropt[0];
        gslref_qrng_free(m.q);
        free(m.ntix); free(m.qmp); free(m.mssr_order); free(m.luchange);
        free(m.start); free(m.maxlims); free(m.mssr); free(m.mx); free(m.diag); free(m.mpopt1);
    }
    else
        memcpy(mpopt, prob->start, sizeof(double) * p);

    /* (re-)initialise, nls.c:539-545 */
    if (w->Lw)
        gslref_winit(w, mpopt, wts);
    else if (has_swts || wgt_i)
        gslref_winit(w, mpopt, wts);
    else
        gslref_winit(w, mpopt, NULL);

    chisq_init = 0.0;
    for (i = 0; i < n; ++i)
        chisq_init += w->f_[i] * w->f_[i];
    chisq0 = chisq1 = chisq_init;
    res->chisq_init = chisq_init;

    tr.maxiter = niter;
    tr.p = p;
    tr.partrace = res->partrace;
    tr.ssrtrace = res->ssrtrace;
    if (verbose)
    {
        if (res->ssrtrace)
            res->ssrtrace[0] = chisq_init;
        if (res->partrace)
            for (k = 0; k < p; ++k)
                res->partrace[(size_t)(niter + 1) * k] = mpopt[k];
    }

    if (!wgt_i)
        status = gslref_driver2(w, niter, xtol, gtol, ftol, verbose ? trace_cb : NULL, &tr,
                                &info, &chisq0, &chisq1);
    else
    {
        psi = (double *)calloc(n, sizeof(double));
        psip = (double *)calloc(n, sizeof(double));
        status = gslref_rho_driver(w, prob, mpopt, wts, workn, workp, psi, psip, wgt_i, niter, xtol,
                                   gtol, ftol, verbose ? trace_cb : NULL, &tr, &info, &chisq0, &chisq1,
                                   &irls_sigma, &irls_iter, &irls_status);
        memcpy(wts, workn, sizeof(double) * n);
        for (k = 0; k < p; ++k)
            irls_delta = fmax(irls_delta, fabs(workp[k] - w->x[k]));
    }

    ok = (status == GSLREF_SUCCESS || status == GSLREF_EMAXITER);
    res->niter = w->niter;
    res->conv = status;
    res->info = info;
    res->ssr = chisq1;
    res->ssrtol = chisq0 - chisq1;
    res->neval[0] = (int)w->nevalf;
    res->neval[1] = (int)w->nevaldf;
    res->neval[2] = (int)w->nevalfvv;
    for (k = 0; k < p; ++k)
        res->par[k] = ok ? w->x[k] : mpopt[k];
    if (res->covar)
    {
        if (ok)
        {
            double *cov = (double *)malloc(sizeof(double) * p * p);
            int k2;
            gslref_covar(n, p, w->J, cov);
            for (k = 0; k < p; ++k)
                for (k2 = 0; k2 < p; ++k2)
                    res->covar[k + p * k2] = cov[k * p + k2];
            free(cov);
        }
        else
            for (k = 0; k < p * p; ++k)
                res->covar[k] = NAN;
    }
    if (res->resid)
        for (i = 0; i < n; ++i)
            res->resid[i] = ok ? w->f_[i] : NAN;
    if (res->grad)
        for (i = 0; i < n; ++i)
            for (k = 0; k < p; ++k)
                res->grad[i + (size_t)n * k] = ok ? w->J[(size_t)i * p + k] : NAN;
    res->irls_sigma = irls_sigma;
    res->irls_status = irls_status;
    res->irls_niter = irls_iter;
    res->irls_tol = irls_delta;
    if (wgt_i)
    {
        for (i = 0; i < n; ++i)
        {
            if (res->irls_weights)
                res->irls_weights[i] = ok ? wts[i] : NAN;
            if (res->irls_psi)
                res->irls_psi[i] = ok ? psi[i] : NAN;
            if (res->irls_dpsi)
                res->irls_dpsi[i] = ok ? psip[i] : NAN;
        }
    }

    gslref_ws_free(w);
    free(wts); free(Lw); free(lu); free(mpopt); free(workp); free(workn); free(psi); free(psip);
    return status;
}
