/*
 * gslref.h -- CPU ORACLE for the gslnls nonlinear least-squares hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped product (gslnls_amd/,
 * include/) may include, link or call this code; only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() use it, and only as a checker.
 *
 * What it restates: the single-threaded fp64 algorithm that the reference
 * (JorisChau/gslnls v1.4.2) runs behind .Call(C_nls) / .Call(C_nls_large):
 *   - the in-tree pieces: src/nls.c, src/nls_fit.c, src/trust.c, src/fdf.c,
 *     src/fdjac.c, src/fdfvv.c, src/nls_utils.c, src/nls_irls.c,
 *     src/nls_mstart.c, src/nls_large.c (cited function by function below);
 *   - the pieces those reach through GSL vtables.  GNU GSL (>= 2.3, CI pin
 *     2.8; DESCRIPTION:16, Dockerfile:4) is an un-vendored dependency that is
 *     NOT under /root/reference, so its published algorithm is restated from
 *     the GSL manual/source semantics: multifit_nlinear
 *     {trust, lm, scaling, cholesky(mcholesky), qr, convergence, covar},
 *     multilarge_nlinear {trust, cgst, lm}, qrng {sobol, halton}.
 *
 * Parity pinning: tests/test_oracle_golden.py checks this oracle against the
 * reference's own golden vectors (README iteration traces and counts, NIST
 * certified values in R/nls_test.R, scalars pinned in
 * inst/unit_tests/unit_tests_gslnls.R).  Unpinned parts are listed in
 * oracle/README.md ("parity unpinned").
 *
 * Layout conventions: matrices handed across this API are documented per
 * argument; internally J is row-major n x p with tda = p exactly as GSL keeps
 * it (src/nls.c:910-912 copies R's column-major matrix into that layout).
 */
#ifndef GSLREF_H
#define GSLREF_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* GSL errno values used by the reference */
#define GSLREF_SUCCESS 0
#define GSLREF_FAILURE (-1)
#define GSLREF_CONTINUE (-2)
#define GSLREF_EINVAL 4
#define GSLREF_EBADFUNC 9
#define GSLREF_EMAXITER 11
#define GSLREF_ENOPROG 27
#define GSLREF_ETOLF 29
#define GSLREF_ETOLX 30
#define GSLREF_ETOLG 31

/* user callbacks: same contract as gsl_f / gsl_df / gsl_fvv in src/nls.c:815-978.
 * f   : residual model(x) - y, length n  (non-finite model value -> +Inf, nls.c:854-855)
 * df  : Jacobian, ROW-major n x p
 * fvv : second directional derivative, length n */
typedef int (*gslref_f_t)(const double *x, void *params, double *f);
typedef int (*gslref_df_t)(const double *x, void *params, double *J);
typedef int (*gslref_fvv_t)(const double *x, const double *v, void *params, double *fvv);

/* multilarge callback, contract of gsl_df_large (src/nls_large.c:474-653):
 * trans=0: v = J u (u length p, v length n); trans=1: v = J^T u; JTJ (p x p
 * row-major, lower triangle significant) filled when non-NULL. v may be NULL. */
typedef int (*gslref_dfl_t)(int trans, const double *x, const double *u, void *params,
                            double *v, double *JTJ);

typedef struct
{
    int n, p;
    gslref_f_t f;
    gslref_df_t df;   /* NULL -> finite differences (src/fdjac.c) */
    gslref_fvv_t fvv; /* NULL -> finite differences (src/fdfvv.c) */
    void *params;
    /* C_nls arguments (src/nls.c:54) */
    const double *start;      /* single start: p values; multi-start: 2 x p col-major [lo,hi] pairs */
    int mstart;               /* start is a 2 x p matrix (Rf_isMatrix(start), nls.c:79) */
    const double *swts;       /* sqrt(weights), length n, or NULL (nls.c:236-241) */
    const double *swts_mat;   /* n x n col-major t(chol(W)) or NULL (nls.c:224-235) */
    const double *lupars;     /* 2 x p col-major [lower,upper] pairs with +-Inf, or NULL (nls.c:248-263) */
    const int *control_int;   /* 15 ints, App. C.1 */
    const double *control_dbl;/* 11 doubles, App. C.2 */
    const int *has_start;     /* 2 x p logical (multi-start only; nls.c:307) */
    int loss_rho;             /* loss_config$rho: 0 default .. 8 lqq (nls.c:219) */
    const double *loss_cc;    /* loss_config$cc tuning constants */
} gslref_problem;

typedef struct
{
    /* outputs mirror the list returned by C_nls (src/nls.c:632-812) */
    double *par;      /* p */
    double *covar;    /* p x p col-major (symmetric) ; NaN on failure */
    double *resid;    /* n  (weighted residual) */
    double *grad;     /* n x p col-major (weighted J), nls.c:718 */
    int niter;
    int conv;         /* status code */
    double ssr;       /* chisq1 */
    double ssrtol;    /* chisq0 - chisq1 */
    int neval[3];     /* f, J, fvv */
    int info;
    double chisq_init;
    /* irls slot (nls.c:756-791) */
    double *irls_weights, *irls_psi, *irls_dpsi; /* n each, may be NULL */
    double irls_sigma, irls_tol;
    int irls_status, irls_niter;
    /* traces (nls.c:980-995): (maxiter+1) x p col-major and maxiter+1; may be NULL */
    double *partrace, *ssrtrace;
    /* multi-start bookkeeping exposed for tests */
    int mstart_nsp, mstart_nwsp, mstart_iters, mstart_stop;
    double mstart_ssropt;
} gslref_result;

/* behind .Call(C_nls): single start, multi-start, IRLS */
int gslref_nls(const gslref_problem *prob, gslref_result *res);

typedef struct
{
    int n, p;
    gslref_f_t f;
    gslref_dfl_t df;
    void *params;
    const double *start;
    const double *weights;     /* length n or NULL (nls_large.c:216-222) */
    const int *control_int;    /* 7 ints, App. C.3 */
    const double *control_dbl; /* 8 doubles */
} gslref_large_problem;

typedef struct
{
    double *par, *covar, *resid;
    int niter, conv;
    double ssr, ssrtol;
    int neval[4]; /* f, dfu, df2, fvv */
    int info;
    double chisq_init;
    double *partrace, *ssrtrace;
} gslref_large_result;

/* behind .Call(C_nls_large) */
int gslref_nls_large(const gslref_large_problem *prob, gslref_large_result *res);

/* ---- pieces exposed individually so tests can pin them ---- */

/* GSL qrng: sobol (p <= 40) / halton; writes npts x dim row-major; skips `skip` draws first */
int gslref_sobol(int dim, int skip, int npts, double *out);
int gslref_halton(int dim, int skip, int npts, double *out);

/* GSL gsl_linalg_mcholesky_decomp + _solve on a p x p row-major SPD-ish matrix (lower used).
 * A is overwritten by L (unit lower) and D (diagonal); perm gets the pivoting. */
int gslref_mcholesky_decomp(int p, double *A, int *perm);
int gslref_mcholesky_solve(int p, const double *LDLT, const int *perm, const double *b, double *x);

/* src/nls_utils.c:55-73 det_cholesky_jtj on row-major n x p J */
double gslref_det_cholesky_jtj(int n, int p, const double *J);
/* src/nls_utils.c:162-217 */
double gslref_median(const double *data, int n);
double gslref_mad(const double *data, int n);
/* src/nls_irls.c:10-341 psi / psi' by loss index 1..8 */
double gslref_psi(double x, const double *cc, int i);
double gslref_psip(double x, const double *cc, int i);
/* src/nls_utils.c:88-150: hat values and Cook's distance; J row-major n x p, f length n */
int gslref_hat_values(int n, int p, const double *J, double *h);
int gslref_cooks_d(int n, int p, const double *f, const double *J, double *d);

/* ---- built-in row models in C (used for CPU baselines where a Python callback
 * would dominate the timing).  params for each is a gslref_rowdata. ---- */
typedef struct
{
    int n;
    int nx;            /* number of regressor columns */
    const double *x;   /* nx columns, column-major n x nx */
    const double *y;   /* n */
    int model;         /* GSLREF_MODEL_* */
    int p;
} gslref_rowdata;

#define GSLREF_MODEL_EXPDECAY 1 /* A*exp(-lam*x)+b            (R/nls.R:143-151)  */
#define GSLREF_MODEL_MISRA1A 2  /* b1*(1-exp(-b2*x))          (R/nls_test.R:174,793) */
#define GSLREF_MODEL_GAUSSPK 3  /* a*exp(-(x-b)^2/(2 c^2))    (README.md:545) */
#define GSLREF_MODEL_GAUSS1 4   /* Gauss1 family p=8          (R/nls_test.R:301) */
#define GSLREF_MODEL_GLMEXP 5   /* exp(a_i^T theta), p = nx    */

/* opt-in: the row models evaluate exp with the device's arithmetic (gslref_models.c), bit for bit */
void gslref_set_device_exp(int on);
double gslref_device_exp(double x);
void gslref_device_exp_array(const double *x, double *out, int n);
int gslref_model_f(const double *x, void *params, double *f);
int gslref_model_df(const double *x, void *params, double *J);
int gslref_model_fvv(const double *x, const double *v, void *params, double *fvv);
int gslref_model_dfl(int trans, const double *x, const double *u, void *params, double *v, double *JTJ);

const char *gslref_strerror(int code);

#ifdef __cplusplus
}
#endif
#endif
