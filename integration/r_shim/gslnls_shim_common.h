/*
 * gslnls_shim_common.h -- what both reference-side shims need to find the formula behind a closure: gsl_nls.formula
 * (R/nls.R:565) and gsl_nls_large.formula (R/nls_large.R:273) build the same
 *     .fn <- function(par, .data = mf) eval(formula[[3L]], envir = c(as.list(par), .data))
 * inside their own call frame, which binds `formula` and `mf`.
 */
#ifndef GSLNLS_SHIM_COMMON_H
#define GSLNLS_SHIM_COMMON_H
#define R_NO_REMAP
#include <R.h>
#include <Rinternals.h>
#include <string.h>
#include "gslnls_core.h"

/* `formula` as the closure .fn sees it: .fn <- function(par, .data = mf) eval(formula[[3]], ...) is created inside
 * gsl_nls.formula (R/nls.R:565), so its enclosure is that call's frame, which binds `formula` and `mf`.  Only that
 * frame is searched (Rf_findVarInFrame): a plain `function` passed as fn must not pick up some unrelated `formula`
 * further up its enclosing environments. */
static SEXP closure_formula(SEXP fn)
{
    if (TYPEOF(fn) != CLOSXP)
        return R_NilValue;
    SEXP formula = Rf_findVarInFrame(CLOENV(fn), Rf_install("formula"));
    if (formula == R_UnboundValue || TYPEOF(formula) != LANGSXP || Rf_length(formula) < 3)
        return R_NilValue;
    return formula;
}

static SEXP deparse_rhs(SEXP formula)
{
    SEXP quoted = PROTECT(Rf_lang2(Rf_install("quote"), CADDR(formula))); /* held while the outer call is allocated */
    SEXP call = PROTECT(Rf_lang2(Rf_install("deparse1"), quoted));
    SEXP txt = PROTECT(Rf_eval(call, R_BaseEnv));
    UNPROTECT(3);
    return txt;
}

/* one data column by name: `mf` is a list (R/nls.R:481 as.list(mf), or the user's `data` list, :448) -- or, for
 * robustness, an environment.  R_NilValue when the name is not a numeric column of length n. */
static SEXP frame_column(SEXP mf, const char *name, int n)
{
    SEXP col = R_NilValue;
    if (TYPEOF(mf) == ENVSXP)
    {
        col = Rf_findVarInFrame(mf, Rf_install(name));
        if (col == R_UnboundValue)
            col = R_NilValue;
    }
    else if (TYPEOF(mf) == VECSXP)
    {
        SEXP nms = Rf_getAttrib(mf, R_NamesSymbol);
        for (int k = 0; !Rf_isNull(nms) && k < Rf_length(mf); k++)
            if (!strcmp(CHAR(STRING_ELT(nms, k)), name))
            {
                col = VECTOR_ELT(mf, k);
                break;
            }
    }
    if (col == R_NilValue || !(TYPEOF(col) == REALSXP || TYPEOF(col) == INTSXP || TYPEOF(col) == LGLSXP) ||
        Rf_length(col) != n)
        return R_NilValue;
    return col;
}

/* the data columns of a formula's right-hand side: all.vars(formula[[3]]) minus the parameter names and `pi`, in that
 * order, as a comma-separated list in `cols` and as pointers in xn (at most 8, the core's limit).  0 when there are
 * more, or the list does not fit. */
static int formula_columns(SEXP formula, const char **pn, int p, char *cols, size_t colsz, const char **xn, int *nxe_out)
{
    SEXP quoted = PROTECT(Rf_lang2(Rf_install("quote"), CADDR(formula)));
    SEXP avcall = PROTECT(Rf_lang2(Rf_install("all.vars"), quoted));
    SEXP vars = Rf_eval(avcall, R_BaseEnv);
    UNPROTECT(2);
    PROTECT(vars);
    int nxe = 0, ok = 1;
    cols[0] = 0;
    for (int v = 0; v < Rf_length(vars) && ok; v++)
    {
        const char *nm = CHAR(STRING_ELT(vars, v)); /* (CHARSXPs of symbols: they outlive `vars`) */
        int is_par = !strcmp(nm, "pi");
        for (int k = 0; k < p; k++)
            is_par |= !strcmp(nm, pn[k]);
        if (is_par)
            continue;
        if (nxe == 8 || strlen(cols) + strlen(nm) + 2 > colsz)
            ok = 0;
        else
        {
            if (nxe)
                strcat(cols, ",");
            strcat(cols, nm);
            xn[nxe++] = nm;
        }
    }
    UNPROTECT(1);
    *nxe_out = nxe;
    return ok;
}

/* trace = TRUE: the text the reference prints while it runs, collected by the core during the call that has just
 * returned (include/gslnls_core.h, gslnls_trace_text), handed to Rprintf piece by piece (Rprintf's own buffer is
 * bounded) */
static void print_trace_text(void)
{
    const size_t len = gslnls_trace_text(NULL, 0);
    if (!len)
        return;
    char *txt = (char *)R_alloc(len + 1, 1);
    gslnls_trace_text(txt, len + 1);
    for (size_t off = 0; off < len;)
    {
        char piece[1025];
        size_t k = len - off < 1024 ? len - off : 1024;
        memcpy(piece, txt + off, k);
        piece[k] = 0;
        Rprintf("%s", piece);
        off += k;
    }
}

#endif
