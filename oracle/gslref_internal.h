/* gslref_internal.h -- ORACLE internals (test infrastructure, never shipped). */
#ifndef GSLREF_INTERNAL_H
#define GSLREF_INTERNAL_H
#include "gslref.h"

int gslref_cholesky_decomp1(int N, double *A);
int gslref_cholesky_invert(int N, double *A);
int gslref_lstsq(int m, int p, double *A, double *b, double *x);
int gslref_covar(int n, int p, const double *J, double *cov);

/* workspace of the dense solver == gsl_multifit_nlinear_workspace + trust_state_t
 * + lm_state_t + solver state (src/gsl_nls.h:110-142 mirrors the GSL-private ones) */
typedef struct
{
    int n, p;
    /* gsl_multifit_nlinear_parameters */
    int trs;    /* 0 lm, 1 lmaccel */
    int scale;  /* 0 more, 1 levenberg, 2 marquardt */
    int solver; /* 0 qr, 1 cholesky, 2 svd (treated as qr) */
    int fdtype; /* 0 forward, 1 center */
    double factor_up, factor_down, avmax, h_df, h_fvv;
    /* fdf */
    gslref_f_t f;
    gslref_df_t df;
    gslref_fvv_t fvv;
    void *params;
    long nevalf, nevaldf, nevalfvv;
    /* workspace */
    double *x, *f_, *J, *g, *dx;
    double *sqrt_wts;      /* NULL or == sqrt_wts_work */
    double *sqrt_wts_work; /* n */
    int niter;
    /* trust state */
    double *diag, *x_trial, *f_trial, *workp, *workn;
    double delta, mu, avratio;
    long nu;
    /* lm state */
    double *vel, *acc, *fvvv;
    /* solver state */
    double *JTJ, *work_JTJ, *rhs;
    int *perm;
    double *aug, *augrhs; /* (n+p) x p and n+p for the qr solver */
    /* extensions of the in-tree iterator (src/trust.c:408-549) */
    const double *Lw; /* n x n row-major unit-lower, or NULL */
    const double *lu; /* 2 x p row-major [lower row; upper row], or NULL */
} gslref_ws;

gslref_ws *gslref_ws_alloc(int n, int p);
void gslref_ws_free(gslref_ws *w);
/* gsl_multifit_nlinear_winit / _init / _winit_LD : wts are WEIGHTS (not sqrt), NULL = unweighted */
int gslref_winit(gslref_ws *w, const double *x, const double *wts);
/* one trust-region iteration */
int gslref_iterate(gslref_ws *w);
/* gsl_multifit_nlinear_test */
int gslref_test(const gslref_ws *w, double xtol, double gtol, double ftol, int *info);
typedef void (*gslref_cb_t)(int iter, void *cbp, const gslref_ws *w, double chisq);
/* src/nls_fit.c:40-121 */
int gslref_driver2(gslref_ws *w, int maxiter, double xtol, double gtol, double ftol,
                   gslref_cb_t cb, void *cbp, int *info, double *chisq0, double *chisq1);
int gslref_eval_f(gslref_ws *w, const double *x, double *y);
int gslref_eval_df(gslref_ws *w, const double *x, const double *f, double *J);
double gslref_det_eval_jtj(gslref_ws *w);

/* IRLS driver, src/nls_irls.c:412-546 */
int gslref_rho_driver(gslref_ws *w, const gslref_problem *prob, const double *mpopt, double *wts,
                      double *workn_wts, double *workp, double *psi, double *psip,
                      int wgt_i, int maxiter, double xtol, double gtol, double ftol,
                      gslref_cb_t cb, void *cbp,
                      int *info, double *chisq0, double *chisq1, double *irls_sigma,
                      int *irls_iter, int *irls_status);

/* multi-start, src/nls_mstart.c + src/nls.c:274-532 */
typedef struct gslref_qrng gslref_qrng;
gslref_qrng *gslref_qrng_alloc(int dim);
void gslref_qrng_init(gslref_qrng *q);
void gslref_qrng_get(gslref_qrng *q, double *v);
void gslref_qrng_free(gslref_qrng *q);

void gslref_order(const double *x, int n, int *order); /* R_orderVector1(.., nalast=TRUE, decreasing=FALSE) */

#endif
