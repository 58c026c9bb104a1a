/*
 * gslnls_hip_large_shim.c -- reference-side binding for .Call(C_nls_large, ...) (src/init.c:16,
 * src/nls_large.c:66-75): gsl_nls_large() with `fn` / `jac` R closures, the Jacobian dense or a Matrix-package
 * sparse matrix.  Not linked in this repository's image (no R); syntax-checked by build() against tests/r_stub/.
 * A maintainer registers
 *     {"C_nls_large", (DL_FUNC) &C_nls_large_hip, 9}
 * in src/init.c in place of C_nls_large; algorithms other than lm / cgst fall through to the original.
 *
 * The closures are evaluated on the R main thread exactly as gsl_f_large / gsl_df_large do
 * (src/nls_large.c:426-472, :474-653) -- but f once per trial point and J once per accepted point instead
 * of once per matrix-vector product; every product with J then runs on the device
 * (gslnls_amd/csrc/sparse_large.hpp).
 *
 * gsl_nls_large(formula, ...): the closures come from the formula method (R/nls_large.R:273, :295-309), whose frame binds
 * `formula` and `mf` exactly as gsl_nls.formula's does.  Then no closure is evaluated during the fit at all: the
 * right-hand side itself goes to the core as a GSLNLS_MODEL_EXPR descriptor (symbolic gradient, up to 64 parameters and
 * 8 data columns) and gslnls_nls_large() runs the multilarge driver on the device's row passes; `grad` of the result is
 * one evaluation of the `jac` closure at the returned coefficients.
 */
#define R_NO_REMAP
#include <R.h>
#include <Rinternals.h>
#include <string.h>
#include "gslnls_core.h"
#include "gslnls_shim_common.h"

SEXP C_nls_large(SEXP fn, SEXP y, SEXP jac, SEXP fvv, SEXP env, SEXP start, SEXP weights, SEXP control_int,
                 SEXP control_dbl); /* original entry, kept for what is not lowered */

typedef struct
{
    SEXP fn, jac, env, parnames;
    SEXP keep;   /* last Jacobian object: its slots must stay valid until the next callback */
    int n, p, failed;
} shim_state;

static SEXP named_par(const double *theta, shim_state *s)
{
    SEXP par = PROTECT(Rf_allocVector(REALSXP, s->p));
    memcpy(REAL(par), theta, sizeof(double) * s->p);
    Rf_setAttrib(par, R_NamesSymbol, s->parnames); /* src/nls_large.c:436-441 */
    UNPROTECT(1);
    return par;
}

static int shim_f(const double *theta, int p, double *fval, int n, void *user)
{
    shim_state *s = (shim_state *)user;
    (void)p;
    SEXP par = PROTECT(named_par(theta, s));
    SEXP call = PROTECT(Rf_lang2(s->fn, par));
    SEXP val = PROTECT(Rf_eval(call, s->env));
    int ok = Rf_isReal(val) && Rf_length(val) == n; /* src/nls_large.c:449-455 */
    if (ok)
        memcpy(fval, REAL(val), sizeof(double) * n); /* non-finite values are mapped to +Inf by the core */
    else
        s->failed = 1;
    UNPROTECT(3);
    return ok ? 0 : 1;
}

/* same classification as match_dg_class, src/nls_large.c:16-49 */
static int dg_class(SEXP obj)
{
    if (Rf_inherits(obj, "dgTMatrix"))
        return GSLNLS_SPARSE_COO;
    if (Rf_inherits(obj, "dgCMatrix"))
        return GSLNLS_SPARSE_CSC;
    if (Rf_inherits(obj, "dgRMatrix"))
        return GSLNLS_SPARSE_CSR;
    return -1; /* dgeMatrix or base matrix: dense */
}

static int shim_jac(const double *theta, int p, gslnls_sparse *J, void *user)
{
    shim_state *s = (shim_state *)user;
    (void)p;
    SEXP par = PROTECT(named_par(theta, s));
    SEXP call = PROTECT(Rf_lang2(s->jac, par));
    SEXP val = PROTECT(Rf_eval(call, s->env));
    R_ReleaseObject(s->keep);
    R_PreserveObject(val); /* slots stay valid until the next call, as gslnls_large_jac_cb asks */
    s->keep = val;
    UNPROTECT(3);
    J->nrow = s->n;
    J->ncol = s->p;
    const int cls = dg_class(val);
    if (cls >= 0)
    {
        SEXP x = R_do_slot(val, Rf_install("x"));
        J->format = cls;
        J->nnz = Rf_length(x);
        J->x = REAL(x);
        if (cls != GSLNLS_SPARSE_COO)
            J->p = INTEGER(R_do_slot(val, Rf_install("p")));
        if (cls != GSLNLS_SPARSE_CSR)
            J->i = INTEGER(R_do_slot(val, Rf_install("i")));
        if (cls != GSLNLS_SPARSE_CSC)
            J->j = INTEGER(R_do_slot(val, Rf_install("j")));
        return 0;
    }
    /* dense n x p (base matrix or dgeMatrix, jacclass -2 / -1 of src/nls_large.c:504): the column-major block as R holds
     * it -- the core multiplies with it as a dense matrix (GSLNLS_SPARSE_DENSE), no index arrays */
    SEXP dx = Rf_inherits(val, "dgeMatrix") ? R_do_slot(val, Rf_install("x")) : val;
    if (!Rf_isReal(dx) || Rf_length(dx) != s->n * s->p)
    {
        s->failed = 1;
        return 1;
    }
    J->format = GSLNLS_SPARSE_DENSE;
    J->nnz = (long)s->n * s->p;
    J->x = REAL(dx);
    return 0;
}

SEXP C_nls_large_hip(SEXP fn, SEXP y, SEXP jac, SEXP fvv, SEXP env, SEXP start, SEXP weights, SEXP control_int,
                     SEXP control_dbl)
{
    const int trs = INTEGER(control_int)[2];
    if ((trs != 0 && trs != 5) || !Rf_isFunction(jac) || gslnls_device_count() < 1)
        return C_nls_large(fn, y, jac, fvv, env, start, weights, control_int, control_dbl);

    SEXP startvec = PROTECT(Rf_coerceVector(start, REALSXP));
    const int p = Rf_length(startvec), n = Rf_length(y);
    shim_state s = {fn, jac, env, Rf_getAttrib(start, R_NamesSymbol), R_NilValue, n, p, 0};

    /* ---- the formula method: lower the right-hand side itself ---- */
    gslnls_model model;
    memset(&model, 0, sizeof model);
    const char *pn[64], *xn[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    char cols[256];
    int lowered = 0;
    SEXP formula = closure_formula(fn);
    SEXP rhs = PROTECT(formula == R_NilValue ? R_NilValue : deparse_rhs(formula));
    if (rhs != R_NilValue && p <= 64 && !Rf_isNull(s.parnames))
    {
        int nxe = 0;
        for (int k = 0; k < p; k++)
            pn[k] = CHAR(STRING_ELT(s.parnames, k));
        if (formula_columns(formula, pn, p, cols, sizeof(cols), xn, &nxe))
        {
            SEXP mf = Rf_findVarInFrame(CLOENV(fn), Rf_install("mf"));
            const int nx = nxe > 0 ? nxe : 1;
            double *X = (double *)R_alloc((size_t)n * nx, sizeof(double));
            memset(X, 0, sizeof(double) * (size_t)n * nx);
            lowered = 1;
            for (int c = 0; c < nxe && lowered; c++)
            {
                /* (a symbol that is not a numeric column of length n of the model frame is something the closure would
                 * find through its enclosure and the device cannot: the callbacks serve that fit) */
                SEXP raw = (mf == R_UnboundValue) ? R_NilValue : frame_column(mf, xn[c], n);
                if (raw == R_NilValue)
                    lowered = 0;
                else
                {
                    SEXP col = PROTECT(Rf_coerceVector(raw, REALSXP));
                    memcpy(X + (size_t)c * n, REAL(col), sizeof(double) * n);
                    UNPROTECT(1);
                }
            }
            model.id = GSLNLS_MODEL_EXPR;
            model.p = p;
            model.nx = nxe;
            model.x = X;
            model.expr = CHAR(STRING_ELT(rhs, 0));
            model.parnames = pn;
            model.xnames = xn;
            model.lowering = GSLNLS_LOWER_AUTO;
        }
    }

    int err = 0;
    gslnls_large *h = NULL;
    if (!lowered)
    {
        h = gslnls_large_create_sparse(n, p, REAL(y), Rf_isNull(weights) ? NULL : REAL(weights), shim_f, shim_jac, &s, &err);
        if (!h)
        {
            UNPROTECT(2);
            return C_nls_large(fn, y, jac, fvv, env, start, weights, control_int, control_dbl);
        }
    }
    const int maxiter = INTEGER(control_int)[0], verbose = INTEGER(control_int)[1];
    /* result list of src/nls_large.c:275-416: par, covar, resid, grad, niter, status, conv, ssr, ssrtol, algorithm,
     * neval [, partrace, ssrtrace] */
    const char *nms13[] = {"par", "covar", "resid", "grad", "niter", "status", "conv", "ssr", "ssrtol", "algorithm",
                           "neval", "partrace", "ssrtrace", ""};
    const char *nms11[] = {"par", "covar", "resid", "grad", "niter", "status", "conv", "ssr", "ssrtol", "algorithm",
                           "neval", ""};
    SEXP ans = PROTECT(Rf_mkNamed(VECSXP, verbose ? nms13 : nms11));
    SEXP par = PROTECT(Rf_allocVector(REALSXP, p)), covar = PROTECT(Rf_allocMatrix(REALSXP, p, p)),
         resid = PROTECT(Rf_allocVector(REALSXP, n));
    gslnls_large_result out;
    memset(&out, 0, sizeof out);
    out.par = REAL(par);
    out.covar = REAL(covar);
    out.resid = REAL(resid);
    int nprot = 6;
    SEXP pt = R_NilValue;
    if (verbose)
    {
        pt = PROTECT(Rf_allocMatrix(REALSXP, maxiter + 1, p));
        SEXP st = PROTECT(Rf_allocVector(REALSXP, maxiter + 1));
        nprot += 2;
        out.partrace = REAL(pt);
        out.ssrtrace = REAL(st);
        SET_VECTOR_ELT(ans, 11, pt);
        SET_VECTOR_ELT(ans, 12, st);
    }
    int rc;
    if (lowered)
        rc = gslnls_nls_large(&model, REAL(y), n, REAL(startvec), Rf_isNull(weights) ? NULL : REAL(weights),
                              INTEGER(control_int), REAL(control_dbl), &out);
    else
    {
        rc = gslnls_large_solve(h, REAL(startvec), INTEGER(control_int), REAL(control_dbl), &out);
        gslnls_large_destroy(h);
    }
    if (rc == GSLNLS_E_UNSUPPORTED || rc == GSLNLS_E_NODEVICE)
    {
        if (s.keep != R_NilValue)
            R_ReleaseObject(s.keep);
        UNPROTECT(nprot);
        return C_nls_large(fn, y, jac, fvv, env, start, weights, control_int, control_dbl);
    }
    const int ok = (out.conv == GSLNLS_SUCCESS || out.conv == GSLNLS_EMAXITER);
    if (verbose)
        print_trace_text(); /* callback_large's lines and the summary block (src/nls_large.c:259-273, :715-739) */
    if (lowered && ok)
    {
        /* `grad`: the Jacobian at the returned coefficients, from the closure the reference would have called last */
        gslnls_sparse unused;
        memset(&unused, 0, sizeof unused);
        (void)shim_jac(out.par, p, &unused, &s); /* (leaves the evaluated object in s.keep) */
    }
    Rf_setAttrib(par, R_NamesSymbol, s.parnames);
    SET_VECTOR_ELT(ans, 0, par);
    if (!Rf_isNull(s.parnames))
    {
        SEXP dn = PROTECT(Rf_allocVector(VECSXP, 2)); /* src/nls_large.c:327-334 */
        SET_VECTOR_ELT(dn, 0, s.parnames);
        SET_VECTOR_ELT(dn, 1, s.parnames);
        Rf_setAttrib(covar, R_DimNamesSymbol, dn);
        UNPROTECT(1);
    }
    SET_VECTOR_ELT(ans, 1, covar);
    SET_VECTOR_ELT(ans, 2, resid);
    {
        /* grad: the reference returns the dense n x p copy of the Jacobian it evaluated last (src/nls_large.c:353-384).
         * The device never densifies J; the last Jacobian object the closure returned (evaluated at the last accepted
         * point = the returned par) is still held, so R's own as.matrix() produces the same matrix. */
        /* every intermediate stays protected across the next allocating call: the call object during Rf_eval, the
         * as.matrix() result during Rf_coerceVector (an ngCMatrix / lgCMatrix Jacobian coerces from logical) */
        SEXP g = R_NilValue;
        PROTECT_INDEX gi;
        PROTECT_WITH_INDEX(g, &gi);
        if (ok && s.keep != R_NilValue)
        {
            SEXP call = PROTECT(Rf_lang2(Rf_install("as.matrix"), s.keep));
            REPROTECT(g = Rf_eval(call, R_BaseEnv), gi);
            UNPROTECT(1); /* call */
            REPROTECT(g = Rf_coerceVector(g, REALSXP), gi);
        }
        else
            REPROTECT(g = Rf_allocMatrix(REALSXP, n, p), gi);
        if (!ok || s.keep == R_NilValue)
            for (size_t i = 0; i < (size_t)n * p; i++)
                REAL(g)[i] = NA_REAL;
        if (!Rf_isNull(s.parnames))
        {
            SEXP dn = PROTECT(Rf_allocVector(VECSXP, 2));
            SET_VECTOR_ELT(dn, 1, s.parnames);
            Rf_setAttrib(g, R_DimNamesSymbol, dn);
            UNPROTECT(1);
        }
        SET_VECTOR_ELT(ans, 3, g);
        UNPROTECT(1);
    }
    if (s.keep != R_NilValue)
        R_ReleaseObject(s.keep);
    if (!ok)
    {
        for (int k = 0; k < p * p; k++)
            REAL(covar)[k] = NA_REAL;
        for (int i = 0; i < n; i++)
            REAL(resid)[i] = NA_REAL;
    }
    SET_VECTOR_ELT(ans, 4, Rf_ScalarInteger(out.niter));
    SET_VECTOR_ELT(ans, 5, Rf_mkString(gslnls_strerror(out.conv)));
    SET_VECTOR_ELT(ans, 6, Rf_ScalarInteger(out.conv));
    SET_VECTOR_ELT(ans, 7, Rf_ScalarReal(out.ssr));
    SET_VECTOR_ELT(ans, 8, Rf_ScalarReal(out.ssrtol));
    SET_VECTOR_ELT(ans, 9, Rf_mkString(gslnls_algorithm_name(trs)));
    {
        const char *en[] = {"f", "dfu", "df2", "fvv", ""}; /* src/nls_large.c:395-401 */
        SEXP neval = PROTECT(Rf_mkNamed(INTSXP, en));
        memcpy(INTEGER(neval), out.neval, sizeof(int) * 4);
        SET_VECTOR_ELT(ans, 10, neval);
        UNPROTECT(1);
    }
    if (verbose && !Rf_isNull(s.parnames))
    {
        SEXP dn = PROTECT(Rf_allocVector(VECSXP, 2));
        SET_VECTOR_ELT(dn, 0, R_NilValue);
        SET_VECTOR_ELT(dn, 1, s.parnames);
        Rf_setAttrib(pt, R_DimNamesSymbol, dn);
        UNPROTECT(1);
    }
    UNPROTECT(nprot);
    return ans;
}
