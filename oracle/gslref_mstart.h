/* gslref_mstart.h -- ORACLE internals: multi-start state == mdata + the pdata fields the
 * multi-start driver touches (src/gsl_nls.h:20-50, :78-107). Test infrastructure only. */
#ifndef GSLREF_MSTART_H
#define GSLREF_MSTART_H
#include "gslref_internal.h"

typedef struct
{
    gslref_ws *w;
    gslref_qrng *q;
    int n;      /* mstart_n: number of sample slots */
    int p;      /* mstart_p: concentration iterations */
    int qtop;   /* mstart_q */
    int s;      /* mstart_s */
    int niter;  /* mstart_maxiter */
    int max;    /* mstart_maxstart */
    int minsp;  /* mstart_minsp */
    int wgt_i;
    int all_start;
    const int *has_start;
    double r, tol, dtol;
    int *ntix;
    double *qmp;
    int *mssr_order;
    int mstop, mstarts, nsp, nwsp;
    int *luchange;
    double rejectscl;
    double mssropt[2], ssrconv[2];
    double *start;   /* 2p working ranges */
    double *maxlims; /* 2p */
    double *mssr;    /* n */
    double *mx;      /* n x p row-major */
    double *diag;    /* p sampling exponents (pars->diag) */
    double *mpopt, *mpopt1;
    double *wts;     /* n weights (pars->wts) */
    int has_swts;    /* !Rf_isNull(pars->swts) */
} gslref_mstate;

void gslref_multistart_driver(gslref_mstate *m, double xtol, double ftol, int use_weights);
#endif
