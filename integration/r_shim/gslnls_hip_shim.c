/*
 * gslnls_hip_shim.c -- the reference-side binding: what a gslnls maintainer adds to src/ so
 * that .Call(C_nls, ...) runs on the MI355X core instead of GSL.
 *
 * Not linked in this repository's image (no R here); __graft_entry__.build() and
 * tests/test_abi.py syntax-check it with gcc -fsyntax-only -Wall against the declaration-only R API
 * stand-in under tests/r_stub/.  It is the complete translation unit a maintainer would drop next to
 * src/nls.c and register in src/init.c in place of C_nls:    {"C_nls", (DL_FUNC) &C_nls_hip, 12}
 * Everything it does is SEXP <-> plain-pointer translation, in the order src/nls.c:66-263 unpacks
 * and src/nls.c:632-812 packs; no numerics.
 *
 * Build (inside the R package):  PKG_LIBS += -L$(GSLNLS_HIP_LIB) -lgslnls_hip
 */
#define R_NO_REMAP
#include <R.h>
#include <Rinternals.h>
#include <string.h>
#include "gslnls_core.h"
#include "gslnls_shim_common.h"

/* original entry, kept as the path for models that do not lower (src/nls.c:54) */
SEXP C_nls(SEXP fn, SEXP y, SEXP jac, SEXP fvv, SEXP env, SEXP start, SEXP swts, SEXP lupars,
           SEXP control_int, SEXP control_dbl, SEXP has_start, SEXP loss_config);

/* user interrupts (Ctrl-C) while the device loop runs: R_CheckUserInterrupt long-jumps, so it is probed under
 * R_ToplevelExec and reported to the core as a status; the core abandons the fit and C_nls_hip raises the error
 * after its own clean-up */
static void probe_interrupt(void *dummy)
{
    (void)dummy;
    R_CheckUserInterrupt();
}
static int interrupt_hook(void) { return R_ToplevelExec(probe_interrupt, NULL) == FALSE; }

/* Unload hook of the package's shared object (library.dynam.unload / R session end via the namespace's .onUnload):
 * formulas seen under the default lowering start a compiler thread inside libgslnls_hip.so; it must be stopped before
 * the C runtime tears the compiler's own static objects down (include/gslnls_core.h, gslnls_shutdown).  The maintainer
 * adds this call to the package's existing R_unload_gslnls, or this function if there is none; R/zzz.R gets
 * .onUnload <- function(libpath) library.dynam.unload("gslnls", libpath) and a
 * reg.finalizer(asNamespace("gslnls"), function(e) .Call(C_gslnls_shutdown), onexit = TRUE). */
void R_unload_gslnls(DllInfo *dll)
{
    (void)dll;
    gslnls_shutdown();
}
SEXP C_gslnls_shutdown(void)
{
    gslnls_shutdown();
    return R_NilValue;
}

/* ---- function models: gsl_nls(fn = <function>, y = ...) (R/nls.R:778) --------------------------------------------
 * The closures are evaluated HERE, on the R thread, the way gsl_f / gsl_df / gsl_fvv evaluate them (src/nls.c:815-978:
 * par as a named numeric vector, or a named list of scalars when start was a list -- control_int[13] == 0 --, then
 * Rf_eval of the prepared call in `env`, then the type / length checks); the core (gslnls_nls_fn) calls back for every
 * evaluation and does everything after it on the device.  An R error inside a closure must not long-jump through the
 * core's C++ frames: R_tryEval catches it, the callback reports failure, C_nls_hip raises the error after the core
 * has returned. */
typedef struct
{
    SEXP fcall, dfcall, fvvcall, rho, names;
    int n, p, startisnum, warn, r_error;
    int bad_result; /* 1 fn, 2 jac, 3 fvv returned the wrong type / length: the warning of src/nls.c:843-844 (:891-892,
                       :953-954) is raised by C_nls_hip once the core has returned -- Rf_warning long-jumps under
                       options(warn = 2) and must not run below the core's C++ frames */
} fn_route;

static SEXP make_par(const fn_route *d, const double *theta)
{
    SEXP par;
    if (d->startisnum)
    {
        par = PROTECT(Rf_allocVector(REALSXP, d->p));
        for (int k = 0; k < d->p; k++)
            REAL(par)[k] = theta[k];
    }
    else
    {
        par = PROTECT(Rf_allocVector(VECSXP, d->p));
        for (int k = 0; k < d->p; k++)
            SET_VECTOR_ELT(par, k, Rf_ScalarReal(theta[k]));
    }
    Rf_setAttrib(par, R_NamesSymbol, d->names);
    UNPROTECT(1);
    return par;
}

static int cb_f(const double *theta, int p, double *fval, int n, void *user)
{
    fn_route *d = (fn_route *)user;
    (void)p;
    SEXP par = PROTECT(make_par(d, theta));
    SETCADR(d->fcall, par);
    int err = 0;
    SEXP v = PROTECT(R_tryEval(d->fcall, d->rho, &err));
    if (err)
    {
        d->r_error = 1;
        UNPROTECT(2);
        return 1;
    }
    if (TYPEOF(v) != REALSXP || Rf_length(v) != n)
    {
        if (d->warn && !d->bad_result)
            d->bad_result = 1;
        UNPROTECT(2);
        return 1; /* GSL_EBADFUNC, src/nls.c:843-844 */
    }
    memcpy(fval, REAL(v), sizeof(double) * (size_t)n); /* model values: the core subtracts y and applies :846-849 */
    UNPROTECT(2);
    return 0;
}

static int cb_jac(const double *theta, int p, double *J, int n, void *user)
{
    fn_route *d = (fn_route *)user;
    SEXP par = PROTECT(make_par(d, theta));
    SETCADR(d->dfcall, par);
    int err = 0;
    SEXP v = PROTECT(R_tryEval(d->dfcall, d->rho, &err));
    if (err)
    {
        d->r_error = 1;
        UNPROTECT(2);
        return 1;
    }
    if (TYPEOF(v) != REALSXP || !Rf_isMatrix(v) || Rf_ncols(v) != p || Rf_nrows(v) != n)
    {
        if (d->warn && !d->bad_result)
            d->bad_result = 2;
        UNPROTECT(2);
        return 1;
    }
    /* column-major n x p as R holds it (src/nls.c:905-912 transposes it into GSL's rows; the core keeps R's layout);
     * non-finite entries are found by the device's pass over the matrix (:894-903) */
    memcpy(J, REAL(v), sizeof(double) * (size_t)n * p);
    UNPROTECT(2);
    return 0;
}

static int cb_fvv(const double *theta, const double *vdir, int p, double *out, int n, void *user)
{
    fn_route *d = (fn_route *)user;
    SEXP par = PROTECT(make_par(d, theta));
    SEXP vpar = PROTECT(Rf_allocVector(REALSXP, p));
    for (int k = 0; k < p; k++)
        REAL(vpar)[k] = vdir[k];
    Rf_setAttrib(vpar, R_NamesSymbol, d->names);
    SETCADR(d->fvvcall, par);
    SETCADDR(d->fvvcall, vpar);
    int err = 0;
    SEXP v = PROTECT(R_tryEval(d->fvvcall, d->rho, &err));
    if (err)
    {
        d->r_error = 1;
        UNPROTECT(3);
        return 1;
    }
    if (TYPEOF(v) != REALSXP || Rf_length(v) != n)
    {
        if (d->warn && !d->bad_result)
            d->bad_result = 3;
        UNPROTECT(3);
        return 1;
    }
    memcpy(out, REAL(v), sizeof(double) * (size_t)n);
    UNPROTECT(3);
    return 0;
}

/* NA-filled REALSXP of length n (src/nls.c:774-780) */
static SEXP na_vector(int n)
{
    SEXP v = PROTECT(Rf_allocVector(REALSXP, n));
    for (int i = 0; i < n; i++)
        REAL(v)[i] = NA_REAL;
    UNPROTECT(1);
    return v;
}

SEXP C_nls_hip(SEXP fn, SEXP y, SEXP jac, SEXP fvv, SEXP env, SEXP start, SEXP swts, SEXP lupars,
               SEXP control_int, SEXP control_dbl, SEXP has_start, SEXP loss_config)
{
    const int n = Rf_length(y);
    const int mstart = Rf_isMatrix(start);                       /* src/nls.c:79 */
    gslnls_set_interrupt_hook(interrupt_hook);
    const int p = mstart ? Rf_ncols(start) : Rf_length(start);
    SEXP parnames = mstart ? VECTOR_ELT(Rf_getAttrib(start, R_DimNamesSymbol), 1)
                           : Rf_getAttrib(start, R_NamesSymbol); /* src/nls.c:158-161 */

    /* ---- model lowering: formula RHS -> registry id, parameter order, data columns ---- */
    SEXP formula = closure_formula(fn);
    SEXP rhs = PROTECT(formula == R_NilValue ? R_NilValue : deparse_rhs(formula));
    const char **pn = (const char **)R_alloc((size_t)(p > 0 ? p : 1), sizeof(char *));
    int *order = (int *)R_alloc((size_t)(p > 0 ? p : 1), sizeof(int));
    char cols[256];
    int model_id = 0;
    for (int k = 0; k < p; k++)
    {
        order[k] = k;
        pn[k] = Rf_isNull(parnames) ? "" : CHAR(STRING_ELT(parnames, k)); /* (R_alloc does not zero: every entry, here) */
    }
    cols[0] = 0;
    if (rhs != R_NilValue && p <= 64 && !Rf_isNull(parnames))
        model_id = gslnls_lower_formula(CHAR(STRING_ELT(rhs, 0)), p, pn, order, cols, sizeof(cols));
    const char *xn[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    /* (up to 9 parameters and 3 data columns: interpreted or natively compiled row models; up to 64 parameters and 8
     * columns: the wide path, J^T J on the matrix cores; up to 512 parameters, single start: the Jacobian
     * as a matrix in HBM -- the core decides, include/gslnls_core.h) */
    if (model_id <= 0 && rhs != R_NilValue && p <= 512 && !Rf_isNull(parnames))
    {
        /* not a hand-written device model: hand the expression itself to the core (GSLNLS_MODEL_EXPR), which
         * compiles it with its symbolic gradient -- the analogue of R/nls.R:565,588-599.  Data columns are the
         * variables of the RHS that are not parameters: all.vars(formula[[3]]) minus names(start). */
        int nxe = 0;
        if (formula_columns(formula, pn, p, cols, sizeof(cols), xn, &nxe))
            model_id = GSLNLS_MODEL_EXPR;
    }
    /* What still leaves for GSL: GLS weights (an n x n factor), the dogleg family, more than 4096 parameters.  Everything
     * else is served: a formula the core can lower has its rows evaluated on the device; ANY other closure -- a `function`
     * model, or the .fn / .jac / .fvv closures gsl_nls.formula builds around a formula the compiler cannot lower (ifelse,
     * pmax, user functions, symbols that are not columns of the model frame; R/nls.R:565-640) -- goes to the core as host
     * callbacks (gslnls_nls_fn_loss / gslnls_nls_fn_mstart: single start or start ranges, default or robust loss). */
    if (Rf_isMatrix(swts) || INTEGER(control_int)[2] > 1 || p > 4096 || TYPEOF(fn) != CLOSXP)
    {
        UNPROTECT(1);
        return C_nls(fn, y, jac, fvv, env, start, swts, lupars, control_int, control_dbl, has_start, loss_config);
    }
    int fn_model = model_id <= 0; /* the closures as callbacks */

    /* data columns from the model frame, column-major n x nx in device regressor order */
    SEXP mf = fn_model ? R_UnboundValue : Rf_findVarInFrame(CLOENV(fn), Rf_install("mf"));
    int nx = 1;
    for (const char *c = cols; *c && !fn_model; c++)
        nx += (*c == ',');
    double *X = fn_model ? NULL : (double *)R_alloc((size_t)n * nx, sizeof(double));
    if (!fn_model)
    {
        char buf[256];
        strncpy(buf, cols, sizeof(buf) - 1);
        buf[sizeof(buf) - 1] = 0;
        int c = 0;
        for (char *tok = strtok(buf, ","); tok; tok = strtok(NULL, ","), c++)
        {
            SEXP raw = (mf == R_UnboundValue) ? R_NilValue : frame_column(mf, tok, n);
            if (raw == R_NilValue)
            {
                /* the symbol is not a column of the model frame (a global, a length-1 constant, ...): the closure
                 * evaluates it through its enclosure, a device row model cannot -> the closures as callbacks */
                fn_model = 1;
                break;
            }
            SEXP col = PROTECT(Rf_coerceVector(raw, REALSXP));
            memcpy(X + (size_t)c * n, REAL(col), sizeof(double) * n);
            UNPROTECT(1);
        }
    }
    gslnls_model model = {model_id, p, nx, X, 0, NULL, NULL, NULL, GSLNLS_LOWER_AUTO};
    if (model_id == GSLNLS_MODEL_EXPR)
    {
        model.expr = CHAR(STRING_ELT(rhs, 0));
        model.parnames = pn;
        model.xnames = xn;
        /* options(gslnls.lowering = "jit") waits for native code of the formula (built in process by hiprtc, cached on disk);
         * the default lets the first fit run interpreted while it is built in the background */
        SEXP opt = Rf_GetOption1(Rf_install("gslnls.lowering"));
        if (Rf_isString(opt) && !strcmp(CHAR(STRING_ELT(opt, 0)), "jit"))
            model.lowering = GSLNLS_LOWER_JIT;
        if (Rf_isString(opt) && !strcmp(CHAR(STRING_ELT(opt, 0)), "vm"))
            model.lowering = GSLNLS_LOWER_VM;
    }

    /* start / bounds / has_start permuted into device parameter order (the identity for the callback route: the closures
     * take `par` in the caller's order) */
    double *st = (double *)R_alloc(2 * p, sizeof(double)), *lu = NULL;
    int *hs = (int *)R_alloc(2 * p, sizeof(int));
    SEXP startvec = PROTECT(Rf_coerceVector(start, REALSXP));
    if (fn_model)
        for (int k = 0; k < p; k++)
            order[k] = k;
    for (int k = 0; k < p; k++)
    {
        if (mstart)
        {
            st[2 * k] = REAL(startvec)[2 * order[k]];
            st[2 * k + 1] = REAL(startvec)[2 * order[k] + 1];
            hs[2 * k] = LOGICAL(has_start)[2 * order[k]];
            hs[2 * k + 1] = LOGICAL(has_start)[2 * order[k] + 1];
        }
        else
            st[k] = REAL(startvec)[order[k]];
    }
    if (Rf_isMatrix(lupars))
    {
        lu = (double *)R_alloc(2 * p, sizeof(double));
        for (int k = 0; k < p; k++)
        {
            lu[2 * k] = REAL(lupars)[2 * order[k]];
            lu[2 * k + 1] = REAL(lupars)[2 * order[k] + 1];
        }
    }

    /* outputs (src/nls.c:632-812) */
    const int niter = INTEGER(control_int)[0], verbose = INTEGER(control_int)[1];
    const int wgt_i = INTEGER(VECTOR_ELT(loss_config, 0))[0];
    const char *nms14[] = {"par", "covar", "resid", "grad", "niter", "status", "conv", "ssr", "ssrtol",
                           "algorithm", "neval", "irls", "partrace", "ssrtrace", ""};
    const char *nms12[] = {"par", "covar", "resid", "grad", "niter", "status", "conv", "ssr", "ssrtol",
                           "algorithm", "neval", "irls", ""};
    SEXP ans = PROTECT(Rf_mkNamed(VECSXP, verbose ? nms14 : nms12));
    SEXP par = PROTECT(Rf_allocVector(REALSXP, p)), cov = PROTECT(Rf_allocMatrix(REALSXP, p, p));
    SEXP resid = PROTECT(Rf_allocVector(REALSXP, n)), grad = PROTECT(Rf_allocMatrix(REALSXP, n, p));
    double *par_d = (double *)R_alloc(p, sizeof(double)), *cov_d = (double *)R_alloc(p * p, sizeof(double));
    double *grad_d = (double *)R_alloc((size_t)n * p, sizeof(double));
    gslnls_result res;
    memset(&res, 0, sizeof(res));
    res.par = par_d;
    res.covar = cov_d;
    res.resid = REAL(resid);
    res.grad = grad_d;
    /* robust loss: the irls slot's three n-vectors (src/nls.c:756-762) are filled by the core */
    SEXP irlswts = R_NilValue, irlspsi = R_NilValue, irlspsip = R_NilValue;
    int nprot = 7; /* rhs, startvec, ans, par, cov, resid, grad */
    if (wgt_i)
    {
        irlswts = PROTECT(Rf_allocVector(REALSXP, n));
        irlspsi = PROTECT(Rf_allocVector(REALSXP, n));
        irlspsip = PROTECT(Rf_allocVector(REALSXP, n));
        nprot += 3;
        res.irls_weights = REAL(irlswts);
        res.irls_psi = REAL(irlspsi);
        res.irls_dpsi = REAL(irlspsip);
    }
    SEXP ptrace = R_NilValue, strace = R_NilValue;
    double *ptrace_d = NULL;
    if (verbose)
    {
        ptrace = PROTECT(Rf_allocMatrix(REALSXP, niter + 1, p));
        strace = PROTECT(Rf_allocVector(REALSXP, niter + 1));
        nprot += 2;
        ptrace_d = (double *)R_alloc((size_t)(niter + 1) * p, sizeof(double));
        res.partrace = ptrace_d;
        res.ssrtrace = REAL(strace);
    }

    int rc = GSLNLS_E_UNSUPPORTED;
    if (verbose)
        gslnls_trace_set_order(order, p); /* the printed parameter vectors in the caller's order */
    if (!fn_model)
    {
        rc = gslnls_nls(&model, REAL(y), n, !Rf_isNull(jac), !Rf_isNull(fvv), st, mstart,
                        Rf_isNull(swts) ? NULL : REAL(swts), 0, lu, INTEGER(control_int), REAL(control_dbl),
                        hs, wgt_i, REAL(VECTOR_ELT(loss_config, 1)), &res);
        if (rc == GSLNLS_E_UNSUPPORTED)
        {
            /* the core refused the expression after all (a function outside the lowering vocabulary, too many
             * instructions, ...): the formula's own closures are still there.  Only GSLNLS_MODEL_EXPR can end up here,
             * and it never permutes the parameters, so st / lu / hs are already in the caller's order. */
            fn_model = 1;
        }
    }
    if (fn_model)
    {
        /* the calls the reference prepares once and re-aims at every evaluation (src/nls.c:155-216) */
        fn_route d;
        memset(&d, 0, sizeof(d));
        d.fcall = PROTECT(Rf_lang2(fn, R_NilValue));
        d.dfcall = PROTECT(Rf_isNull(jac) ? R_NilValue : Rf_lang2(jac, R_NilValue));
        d.fvvcall = PROTECT(Rf_isNull(fvv) ? R_NilValue : Rf_lang3(fvv, R_NilValue, R_NilValue));
        nprot += 3;
        d.rho = env;
        d.names = parnames;
        d.n = n;
        d.p = p;
        d.startisnum = INTEGER(control_int)[13];
        d.warn = !mstart; /* params.warn: TRUE for a single start only, src/nls.c:175-178 */
        if (mstart)
            rc = gslnls_nls_fn_mstart(n, p, REAL(y), cb_f, Rf_isNull(jac) ? NULL : cb_jac, Rf_isNull(fvv) ? NULL : cb_fvv, &d, st,
                                      hs, Rf_isNull(swts) ? NULL : REAL(swts), lu, INTEGER(control_int), REAL(control_dbl),
                                      wgt_i, REAL(VECTOR_ELT(loss_config, 1)), &res);
        else
            rc = gslnls_nls_fn_loss(n, p, REAL(y), cb_f, Rf_isNull(jac) ? NULL : cb_jac, Rf_isNull(fvv) ? NULL : cb_fvv, &d, st,
                                    Rf_isNull(swts) ? NULL : REAL(swts), lu, INTEGER(control_int), REAL(control_dbl), wgt_i,
                                    REAL(VECTOR_ELT(loss_config, 1)), &res);
        if (d.r_error)
        {
            UNPROTECT(nprot);
            Rf_error("error in the model function, its Jacobian or fvv (see the message above)");
        }
        /* the core's frames are gone: the warning the reference raises inside gsl_f / gsl_df / gsl_fvv */
        if (d.bad_result == 1)
            Rf_warning("Evaluating fn does not return numeric vector of expected length n");
        else if (d.bad_result == 2)
            Rf_warning("Evaluating jac does not return numeric matrix of dimensions n x p");
        else if (d.bad_result == 3)
            Rf_warning("Evaluating fvv does not return numeric vector of expected length n");
    }
    if (rc == GSLNLS_E_INTERRUPTED)
        Rf_onintr(); /* does not return: the pending interrupt is re-raised now that the device loop is drained */
    /* no device / not lowered after all, or a qr / svd request on a problem too ill-conditioned for the normal
     * equations (gslnls_solver_served, include/gslnls_core.h): the GSL path still exists */
    if (rc <= GSLNLS_E_NODEVICE || !gslnls_solver_served(INTEGER(control_int), &res))
    {
        UNPROTECT(nprot);
        return C_nls(fn, y, jac, fvv, env, start, swts, lupars, control_int, control_dbl, has_start, loss_config);
    }
    const int ok = (res.conv == GSLNLS_SUCCESS || res.conv == GSLNLS_EMAXITER);
    if (verbose)
        print_trace_text(); /* iteration lines, multi-start / IRLS progress, summary block (src/nls.c:610-630) */

    /* back to the caller's parameter order */
    for (int k = 0; k < p; k++)
    {
        REAL(par)[order[k]] = par_d[k];
        for (int j = 0; j < p; j++)
            REAL(cov)[order[k] + p * order[j]] = cov_d[k + p * j];
        memcpy(REAL(grad) + (size_t)n * order[k], grad_d + (size_t)n * k, sizeof(double) * n);
        if (verbose)
            memcpy(REAL(ptrace) + (size_t)(niter + 1) * order[k], ptrace_d + (size_t)(niter + 1) * k,
                   sizeof(double) * (niter + 1));
    }
    Rf_setAttrib(par, R_NamesSymbol, parnames);
    SET_VECTOR_ELT(ans, 0, par);
    {
        SEXP dn = PROTECT(Rf_allocVector(VECSXP, 2)); /* src/nls.c:686-693 */
        SET_VECTOR_ELT(dn, 0, parnames);
        SET_VECTOR_ELT(dn, 1, parnames);
        Rf_setAttrib(cov, R_DimNamesSymbol, dn);
        UNPROTECT(1);
    }
    SET_VECTOR_ELT(ans, 1, cov);
    SET_VECTOR_ELT(ans, 2, resid);
    {
        SEXP dn = PROTECT(Rf_allocVector(VECSXP, 2)); /* column names only, src/nls.c:728-734 */
        SET_VECTOR_ELT(dn, 1, parnames);
        Rf_setAttrib(grad, R_DimNamesSymbol, dn);
        UNPROTECT(1);
    }
    SET_VECTOR_ELT(ans, 3, grad);
    SET_VECTOR_ELT(ans, 4, Rf_ScalarInteger(res.niter));
    SET_VECTOR_ELT(ans, 5, Rf_mkString(gslnls_strerror(res.conv)));
    SET_VECTOR_ELT(ans, 6, Rf_ScalarInteger(res.conv));
    SET_VECTOR_ELT(ans, 7, Rf_ScalarReal(res.ssr));
    SET_VECTOR_ELT(ans, 8, Rf_ScalarReal(res.ssrtol));
    SET_VECTOR_ELT(ans, 9, Rf_mkString(gslnls_algorithm_name(INTEGER(control_int)[2])));
    {
        const char *en[] = {"f", "J", "fvv", ""};
        SEXP ne = PROTECT(Rf_mkNamed(INTSXP, en));
        for (int k = 0; k < 3; k++)
            INTEGER(ne)[k] = res.neval[k];
        SET_VECTOR_ELT(ans, 10, ne);
        UNPROTECT(1);
    }
    if (wgt_i)
    {
        /* the irls slot, element for element as src/nls.c:756-791 builds it (gslModel reads all eight,
         * R/nls.R:1413-1484); NA vectors when the fit failed (the core fills them with NaN in that case too, but R
         * distinguishes NA_real_ from NaN) */
        const char *irlsnms[] = {"irls_weights", "irls_psi", "irls_dpsi", "irls_sigma", "irls_status", "irls_niter",
                                 "irls_tol", "irls_conv", ""};
        SEXP ansirls = PROTECT(Rf_mkNamed(VECSXP, irlsnms));
        SET_VECTOR_ELT(ansirls, 0, ok ? irlswts : na_vector(n));
        SET_VECTOR_ELT(ansirls, 1, ok ? irlspsi : na_vector(n));
        SET_VECTOR_ELT(ansirls, 2, ok ? irlspsip : na_vector(n));
        SET_VECTOR_ELT(ansirls, 3, Rf_ScalarReal(res.irls_sigma));
        SET_VECTOR_ELT(ansirls, 4, Rf_mkString(gslnls_strerror(res.irls_status)));
        SET_VECTOR_ELT(ansirls, 5, Rf_ScalarInteger(res.irls_niter));
        SET_VECTOR_ELT(ansirls, 6, Rf_ScalarReal(res.irls_tol));
        SET_VECTOR_ELT(ansirls, 7, Rf_ScalarInteger(res.irls_status));
        SET_VECTOR_ELT(ans, 11, ansirls);
        UNPROTECT(1);
    }
    if (verbose)
    {
        SEXP dn = PROTECT(Rf_allocVector(VECSXP, 2)); /* src/nls.c:793-803 */
        SET_VECTOR_ELT(dn, 0, R_NilValue);
        SET_VECTOR_ELT(dn, 1, parnames);
        Rf_setAttrib(ptrace, R_DimNamesSymbol, dn);
        UNPROTECT(1);
        SET_VECTOR_ELT(ans, 12, ptrace);
        SET_VECTOR_ELT(ans, 13, strace);
    }
    if (!ok)
    {
        /* src/nls.c:667-737: NA (not NaN) fills on failure */
        for (int k = 0; k < p * p; k++)
            REAL(cov)[k] = NA_REAL;
        for (int i = 0; i < n; i++)
            REAL(resid)[i] = NA_REAL;
        for (size_t i = 0; i < (size_t)n * p; i++)
            REAL(grad)[i] = NA_REAL;
    }
    UNPROTECT(nprot);
    return ans;
}
