/*
 * rmini.c -- a small FUNCTIONAL implementation of the slice of R's C API that integration/r_shim/gslnls_hip_shim.c and
 * gslnls_hip_large_shim.c use on their `function`-model routes (incl. Matrix-package objects by class and slot, as.matrix()), so that the shim can be EXECUTED in an image that has no R ("the
 * shim has only ever been type-checked").  tests/test_gpu_r_shim.py compiles this file together with the shim against the
 * declarations in tests/r_stub/, links libgslnls_hip.so, builds the twelve .Call arguments out of the rm_* helpers below
 * (closures are C callbacks supplied through ctypes) and reads the returned list back.
 *
 * Test infrastructure only.  What it is NOT: an R interpreter -- Rf_eval evaluates a call whose head is a closure object
 * created by rm_closure and nothing else (no `quote`, `deparse1`, `all.vars`: the formula route of the shim needs R itself);
 * nothing is ever freed (PROTECT is the identity); Rf_error aborts the process with the message.
 */
#define R_NO_REMAP
#include <R.h>
#include <Rinternals.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef SEXP (*rm_cb)(SEXP *args, int nargs, void *user);
struct SEXPREC
{
    int type, len;
    void *data;                    /* REAL / INTEGER / LOGICAL payload, SEXP[] of a list / string vector / call, char[] */
    SEXP names, dim, dimnames;     /* the three attributes the shim touches */
    rm_cb cb;                      /* CLOSXP */
    void *user;
    SEXP env;                      /* CLOENV of a closure; ENVSXP: data = SEXP[2 * len] (symbol, value) */
    const char *klass;             /* class of an S4-like object made by rm_s4 (Matrix package: dgCMatrix, dgRMatrix, dgTMatrix, dgeMatrix) */
    SEXP slots;                    /* its slots: a named list */
};

static struct SEXPREC nil_rec = {NILSXP, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, unbound_rec = {NILSXP, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
static struct SEXPREC baseenv_rec = {ENVSXP, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
static struct SEXPREC sym_names = {SYMSXP, 0, (void *)"names", 0, 0, 0, 0, 0, 0, 0, 0}, sym_dimnames = {SYMSXP, 0, (void *)"dimnames", 0, 0, 0, 0, 0, 0, 0, 0},
                      sym_dim = {SYMSXP, 0, (void *)"dim", 0, 0, 0, 0, 0, 0, 0, 0};
SEXP R_NilValue = &nil_rec, R_UnboundValue = &unbound_rec, R_BaseEnv = &baseenv_rec, R_GlobalEnv = &baseenv_rec;
SEXP R_NamesSymbol = &sym_names, R_DimNamesSymbol = &sym_dimnames, R_DimSymbol = &sym_dim;
double R_NaReal;
int R_NaInt = -2147483647 - 1;

static char warnings_buf[4096], print_buf[1 << 20];
static size_t print_len;
static int fell_through;

static SEXP new_rec(int type, int len, size_t bytes)
{
    SEXP s = (SEXP)calloc(1, sizeof(struct SEXPREC));
    s->type = type;
    s->len = len;
    s->data = bytes ? calloc(1, bytes) : NULL;
    s->names = s->dim = s->dimnames = R_NilValue;
    s->env = s->slots = R_NilValue;
    return s;
}
__attribute__((constructor)) static void rm_init(void)
{
    unsigned long long na = 0x7FF00000000007A2ull; /* R's NA_real_: a NaN with payload 1954 */
    memcpy(&R_NaReal, &na, sizeof(na));
    SEXP statics[] = {&nil_rec, &unbound_rec, &baseenv_rec, &sym_names, &sym_dimnames, &sym_dim};
    for (int k = 0; k < 6; k++) /* (static initialisers cannot name R_NilValue: the attribute slots are set here) */
        statics[k]->names = statics[k]->dim = statics[k]->dimnames = statics[k]->env = statics[k]->slots = &nil_rec;
}

SEXP Rf_protect(SEXP s) { return s; }
void Rf_unprotect(int n) { (void)n; }
void R_ProtectWithIndex(SEXP s, PROTECT_INDEX *i) { (void)s; *i = 0; }
void R_Reprotect(SEXP s, PROTECT_INDEX i) { (void)s; (void)i; }
void R_PreserveObject(SEXP s) { (void)s; }
void R_ReleaseObject(SEXP s) { (void)s; }

SEXP Rf_allocVector(SEXPTYPE t, R_xlen_t n)
{
    size_t w = t == REALSXP ? sizeof(double) : (t == INTSXP || t == LGLSXP) ? sizeof(int) : sizeof(SEXP);
    SEXP s = new_rec((int)t, (int)n, w * (size_t)(n > 0 ? n : 1));
    if (t == VECSXP || t == STRSXP)
        for (R_xlen_t k = 0; k < n; k++)
            ((SEXP *)s->data)[k] = R_NilValue;
    return s;
}
SEXP Rf_allocMatrix(SEXPTYPE t, int nr, int nc)
{
    SEXP s = Rf_allocVector(t, (R_xlen_t)nr * nc), d = Rf_allocVector(INTSXP, 2);
    ((int *)d->data)[0] = nr;
    ((int *)d->data)[1] = nc;
    s->dim = d;
    return s;
}
SEXP Rf_mkChar(const char *c)
{
    SEXP s = new_rec(9 /* CHARSXP */, (int)strlen(c), strlen(c) + 1);
    strcpy((char *)s->data, c);
    return s;
}
SEXP Rf_mkString(const char *c)
{
    SEXP s = Rf_allocVector(STRSXP, 1);
    ((SEXP *)s->data)[0] = Rf_mkChar(c);
    return s;
}
SEXP Rf_ScalarString(SEXP c)
{
    SEXP s = Rf_allocVector(STRSXP, 1);
    ((SEXP *)s->data)[0] = c;
    return s;
}
SEXP Rf_mkNamed(SEXPTYPE t, const char **names)
{
    int n = 0;
    while (names[n][0])
        n++;
    SEXP s = Rf_allocVector(t, n), nm = Rf_allocVector(STRSXP, n);
    for (int k = 0; k < n; k++)
        ((SEXP *)nm->data)[k] = Rf_mkChar(names[k]);
    s->names = nm;
    return s;
}
SEXP Rf_ScalarInteger(int v)
{
    SEXP s = Rf_allocVector(INTSXP, 1);
    ((int *)s->data)[0] = v;
    return s;
}
SEXP Rf_ScalarReal(double v)
{
    SEXP s = Rf_allocVector(REALSXP, 1);
    ((double *)s->data)[0] = v;
    return s;
}
SEXP Rf_coerceVector(SEXP x, SEXPTYPE t)
{
    if ((SEXPTYPE)x->type == t)
        return x;
    if (t == REALSXP && (x->type == INTSXP || x->type == LGLSXP))
    {
        SEXP s = Rf_allocVector(REALSXP, x->len);
        for (int k = 0; k < x->len; k++)
            ((double *)s->data)[k] = ((int *)x->data)[k] == R_NaInt ? R_NaReal : (double)((int *)x->data)[k];
        s->names = x->names;
        s->dim = x->dim;
        s->dimnames = x->dimnames;
        return s;
    }
    fprintf(stderr, "rmini: Rf_coerceVector %d -> %u not implemented\n", x->type, t);
    abort();
}
SEXP Rf_install(const char *name)
{
    static SEXP table[256];
    static int ntable;
    if (!strcmp(name, "names"))
        return R_NamesSymbol;
    if (!strcmp(name, "dimnames"))
        return R_DimNamesSymbol;
    if (!strcmp(name, "dim"))
        return R_DimSymbol;
    for (int k = 0; k < ntable; k++)
        if (!strcmp((const char *)table[k]->data, name))
            return table[k];
    SEXP s = new_rec(SYMSXP, 0, strlen(name) + 1);
    strcpy((char *)s->data, name);
    if (ntable < 256)
        table[ntable++] = s;
    return s;
}
static SEXP lang(int n, SEXP a, SEXP b, SEXP c)
{
    SEXP s = new_rec(LANGSXP, n, sizeof(SEXP) * 3);
    ((SEXP *)s->data)[0] = a;
    ((SEXP *)s->data)[1] = b;
    ((SEXP *)s->data)[2] = c;
    return s;
}
SEXP Rf_lang2(SEXP f, SEXP a) { return lang(2, f, a, R_NilValue); }
SEXP Rf_lang3(SEXP f, SEXP a, SEXP b) { return lang(3, f, a, b); }
SEXP SETCADR(SEXP call, SEXP v) { return ((SEXP *)call->data)[1] = v; }
SEXP SETCADDR(SEXP call, SEXP v) { return ((SEXP *)call->data)[2] = v; }
SEXP CAR(SEXP call) { return call->type == LANGSXP ? ((SEXP *)call->data)[0] : R_NilValue; }
SEXP CDR(SEXP call) { (void)call; return R_NilValue; }
SEXP CADR(SEXP call) { return call->type == LANGSXP && call->len > 1 ? ((SEXP *)call->data)[1] : R_NilValue; }
SEXP CADDR(SEXP call) { return call->type == LANGSXP && call->len > 2 ? ((SEXP *)call->data)[2] : R_NilValue; }
SEXP CLOENV(SEXP f) { return f->env; }

static SEXP rm_as_matrix(SEXP x);
SEXP R_tryEval(SEXP call, SEXP rho, int *err)
{
    (void)rho;
    if (err)
        *err = 0;
    if (call->type == LANGSXP && ((SEXP *)call->data)[0]->type == SYMSXP)
    {
        /* the four base functions the shims ask R for: as.matrix(J); quote(e), deparse1(quote(e)), all.vars(quote(e)) with e
         * the right-hand side of a formula -- here an object made by rm_expr that carries its own text and symbols */
        const char *f = (const char *)((SEXP *)call->data)[0]->data;
        SEXP a = ((SEXP *)call->data)[1];
        if (!strcmp(f, "as.matrix"))
            return rm_as_matrix(a);
        if (a->type == LANGSXP && ((SEXP *)a->data)[0]->type == SYMSXP && !strcmp((const char *)((SEXP *)a->data)[0]->data, "quote"))
            a = ((SEXP *)a->data)[1];
        if (!strcmp(f, "quote"))
            return a;
        if (a->type == 99 && !strcmp(f, "deparse1"))
            return Rf_mkString((const char *)a->data);
        if (a->type == 99 && !strcmp(f, "all.vars"))
            return a->slots;
    }
    if (call->type != LANGSXP || ((SEXP *)call->data)[0]->type != CLOSXP)
    {
        fprintf(stderr, "rmini: only calls of rm_closure objects can be evaluated (the formula route of the shim needs R itself)\n");
        abort();
    }
    SEXP f = ((SEXP *)call->data)[0];
    SEXP v = f->cb((SEXP *)call->data + 1, call->len - 1, f->user);
    if (!v)
    {
        if (err)
            *err = 1; /* the closure "raised an R error" */
        return R_NilValue;
    }
    return v;
}
SEXP Rf_eval(SEXP call, SEXP rho) { return R_tryEval(call, rho, NULL); }

SEXP Rf_findVarInFrame(SEXP env, SEXP sym)
{
    if (env->type != ENVSXP)
        return R_UnboundValue;
    for (int k = 0; k < env->len; k++)
        if (((SEXP *)env->data)[2 * k] == sym)
            return ((SEXP *)env->data)[2 * k + 1];
    return R_UnboundValue;
}
SEXP Rf_findVar(SEXP sym, SEXP env) { return Rf_findVarInFrame(env, sym); }
SEXP Rf_getAttrib(SEXP x, SEXP what)
{
    return what == R_NamesSymbol ? x->names : what == R_DimSymbol ? x->dim : what == R_DimNamesSymbol ? x->dimnames : R_NilValue;
}
SEXP Rf_setAttrib(SEXP x, SEXP what, SEXP v)
{
    if (what == R_NamesSymbol)
        x->names = v;
    else if (what == R_DimSymbol)
        x->dim = v;
    else if (what == R_DimNamesSymbol)
        x->dimnames = v;
    return v;
}
SEXP Rf_GetOption1(SEXP sym) { (void)sym; return R_NilValue; }
SEXP R_do_slot(SEXP x, SEXP sym)
{
    if (x->slots != R_NilValue)
        for (int k = 0; k < x->slots->len; k++)
            if (!strcmp((const char *)((SEXP *)x->slots->names->data)[k]->data, (const char *)sym->data))
                return ((SEXP *)x->slots->data)[k];
    fprintf(stderr, "rmini: no slot %s\n", (const char *)sym->data);
    abort();
}
/* as.matrix() of a Matrix-package object made by rm_s4 (slots as the package names them), or of a base matrix (itself) */
static SEXP rm_as_matrix(SEXP x)
{
    if (!x->klass)
        return x;
    const int *Dim = (const int *)R_do_slot(x, Rf_install("Dim"))->data;
    const int nr = Dim[0], nc = Dim[1];
    SEXP m = Rf_allocMatrix(REALSXP, nr, nc), xs = R_do_slot(x, Rf_install("x"));
    double *d = (double *)m->data;
    const double *v = (const double *)xs->data;
    if (!strcmp(x->klass, "dgeMatrix"))
        memcpy(d, v, sizeof(double) * (size_t)nr * nc);
    else if (!strcmp(x->klass, "dgCMatrix"))
    {
        const int *pp = (const int *)R_do_slot(x, Rf_install("p"))->data, *ii = (const int *)R_do_slot(x, Rf_install("i"))->data;
        for (int c = 0; c < nc; c++)
            for (int e = pp[c]; e < pp[c + 1]; e++)
                d[ii[e] + (size_t)nr * c] += v[e];
    }
    else if (!strcmp(x->klass, "dgRMatrix"))
    {
        const int *pp = (const int *)R_do_slot(x, Rf_install("p"))->data, *jj = (const int *)R_do_slot(x, Rf_install("j"))->data;
        for (int r = 0; r < nr; r++)
            for (int e = pp[r]; e < pp[r + 1]; e++)
                d[r + (size_t)nr * jj[e]] += v[e];
    }
    else
    {
        const int *ii = (const int *)R_do_slot(x, Rf_install("i"))->data, *jj = (const int *)R_do_slot(x, Rf_install("j"))->data;
        for (int e = 0; e < xs->len; e++)
            d[ii[e] + (size_t)nr * jj[e]] += v[e];
    }
    return m;
}
R_len_t Rf_length(SEXP x) { return x->len; }
int Rf_nrows(SEXP x) { return x->dim != R_NilValue ? ((int *)x->dim->data)[0] : x->len; }
int Rf_ncols(SEXP x) { return x->dim != R_NilValue ? ((int *)x->dim->data)[1] : 1; }
Rboolean Rf_isNull(SEXP x) { return x == R_NilValue ? TRUE : FALSE; }
Rboolean Rf_isMatrix(SEXP x) { return (x->dim != R_NilValue && x->dim->len == 2) ? TRUE : FALSE; }
Rboolean Rf_isReal(SEXP x) { return x->type == REALSXP ? TRUE : FALSE; }
Rboolean Rf_isString(SEXP x) { return x->type == STRSXP ? TRUE : FALSE; }
Rboolean Rf_isFunction(SEXP x) { return x->type == CLOSXP ? TRUE : FALSE; }
Rboolean Rf_isEnvironment(SEXP x) { return x->type == ENVSXP ? TRUE : FALSE; }
Rboolean Rf_isNewList(SEXP x) { return x->type == VECSXP ? TRUE : FALSE; }
Rboolean Rf_inherits(SEXP x, const char *c) { return (x->klass && !strcmp(x->klass, c)) ? TRUE : FALSE; }
int TYPEOF(SEXP x) { return x->type; }
double *REAL(SEXP x) { return (double *)x->data; }
int *INTEGER(SEXP x) { return (int *)x->data; }
int *LOGICAL(SEXP x) { return (int *)x->data; }
const char *CHAR(SEXP x) { return (const char *)x->data; }
SEXP STRING_ELT(SEXP x, R_xlen_t k) { return ((SEXP *)x->data)[k]; }
SEXP VECTOR_ELT(SEXP x, R_xlen_t k) { return ((SEXP *)x->data)[k]; }
SEXP SET_VECTOR_ELT(SEXP x, R_xlen_t k, SEXP v) { return ((SEXP *)x->data)[k] = v; }
void SET_STRING_ELT(SEXP x, R_xlen_t k, SEXP v) { ((SEXP *)x->data)[k] = v; }

void Rf_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    fprintf(stderr, "rmini: Rf_error: ");
    vfprintf(stderr, fmt, ap);
    fprintf(stderr, "\n");
    va_end(ap);
    abort();
}
void Rf_warning(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    size_t used = strlen(warnings_buf);
    vsnprintf(warnings_buf + used, sizeof(warnings_buf) - used - 2, fmt, ap);
    strcat(warnings_buf, "\n");
    va_end(ap);
}
void Rprintf(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    int k = vsnprintf(print_buf + print_len, sizeof(print_buf) - print_len, fmt, ap);
    if (k > 0)
        print_len = print_len + (size_t)k < sizeof(print_buf) ? print_len + (size_t)k : sizeof(print_buf) - 1;
    va_end(ap);
}
void Rf_onintr(void) { Rf_error("interrupt"); }
void R_CheckUserInterrupt(void) {}
Rboolean R_ToplevelExec(void (*fun)(void *), void *data)
{
    fun(data);
    return TRUE;
}
char *R_alloc(size_t n, int size) { return (char *)malloc(n * (size_t)size + 16); }

/* the original entry the shim keeps as its fall-back: here it only records that the call left for GSL */
SEXP C_nls(SEXP fn, SEXP y, SEXP jac, SEXP fvv, SEXP env, SEXP start, SEXP swts, SEXP lupars, SEXP control_int, SEXP control_dbl,
           SEXP has_start, SEXP loss_config)
{
    (void)fn; (void)y; (void)jac; (void)fvv; (void)env; (void)start; (void)swts; (void)lupars; (void)control_int; (void)control_dbl;
    (void)has_start; (void)loss_config;
    fell_through += 1;
    return R_NilValue;
}

SEXP C_nls_large(SEXP fn, SEXP y, SEXP jac, SEXP fvv, SEXP env, SEXP start, SEXP weights, SEXP control_int, SEXP control_dbl)
{
    (void)fn; (void)y; (void)jac; (void)fvv; (void)env; (void)start; (void)weights; (void)control_int; (void)control_dbl;
    fell_through += 1;
    return R_NilValue;
}

/* ---- what the test builds its arguments with and reads the answer by ---- */
/* the right-hand side of a formula as R would hold it, reduced to what the shims ask about it: its text (deparse1) and its
 * symbols in order of appearance (all.vars) */
SEXP rm_expr(const char *text, SEXP vars)
{
    SEXP s = new_rec(99, 0, strlen(text) + 1);
    strcpy((char *)s->data, text);
    s->slots = vars;
    return s;
}
SEXP rm_formula(SEXP lhs, SEXP rhs) { return lang(3, Rf_install("~"), lhs, rhs); }
void rm_env_set(SEXP env, const char *name, SEXP value)
{
    if (env->len >= 16)
        abort();
    ((SEXP *)env->data)[2 * env->len] = Rf_install(name);
    ((SEXP *)env->data)[2 * env->len + 1] = value;
    env->len += 1;
}
SEXP rm_s4(const char *klass, SEXP named_slots)
{
    SEXP s = new_rec(25 /* S4SXP */, 0, 0);
    char *k = (char *)malloc(strlen(klass) + 1);
    strcpy(k, klass);
    s->klass = k;
    s->slots = named_slots;
    return s;
}
SEXP rm_nil(void) { return R_NilValue; }
SEXP rm_real(int n, const double *v)
{
    SEXP s = Rf_allocVector(REALSXP, n);
    if (v)
        memcpy(s->data, v, sizeof(double) * (size_t)n);
    return s;
}
SEXP rm_int(int n, const int *v, int logical)
{
    SEXP s = Rf_allocVector(logical ? LGLSXP : INTSXP, n);
    memcpy(s->data, v, sizeof(int) * (size_t)n);
    return s;
}
SEXP rm_strings(int n, const char **v)
{
    SEXP s = Rf_allocVector(STRSXP, n);
    for (int k = 0; k < n; k++)
        ((SEXP *)s->data)[k] = Rf_mkChar(v[k]);
    return s;
}
void rm_set_names(SEXP x, SEXP names) { x->names = names; }
void rm_set_dim(SEXP x, int nr, int nc, SEXP rownames, SEXP colnames)
{
    SEXP d = Rf_allocVector(INTSXP, 2);
    ((int *)d->data)[0] = nr;
    ((int *)d->data)[1] = nc;
    x->dim = d;
    if (rownames != R_NilValue || colnames != R_NilValue)
    {
        SEXP dn = Rf_allocVector(VECSXP, 2);
        ((SEXP *)dn->data)[0] = rownames;
        ((SEXP *)dn->data)[1] = colnames;
        x->dimnames = dn;
    }
}
SEXP rm_list(int n) { return Rf_allocVector(VECSXP, n); }
void rm_list_set(SEXP l, int k, SEXP v) { ((SEXP *)l->data)[k] = v; }
SEXP rm_list_get(SEXP l, int k) { return ((SEXP *)l->data)[k]; }
SEXP rm_env(void) { return new_rec(ENVSXP, 0, sizeof(SEXP) * 2 * 16); }
SEXP rm_closure(rm_cb cb, void *user, SEXP env)
{
    SEXP s = new_rec(CLOSXP, 0, 0);
    s->cb = cb;
    s->user = user;
    s->env = env;
    return s;
}
int rm_type(SEXP x) { return x->type; }
int rm_length(SEXP x) { return x->len; }
double *rm_real_ptr(SEXP x) { return (double *)x->data; }
int *rm_int_ptr(SEXP x) { return (int *)x->data; }
const char *rm_string(SEXP x, int k) { return x->type == STRSXP ? (const char *)((SEXP *)x->data)[k]->data : ""; }
SEXP rm_names(SEXP x) { return x->names; }
SEXP rm_dimnames(SEXP x) { return x->dimnames; }
int rm_nrow(SEXP x) { return Rf_nrows(x); }
int rm_ncol(SEXP x) { return Rf_ncols(x); }
const char *rm_warnings(void) { return warnings_buf; }
const char *rm_printed(void) { return print_buf; }
int rm_fell_through(void) { return fell_through; }
void rm_reset(void)
{
    warnings_buf[0] = 0;
    print_buf[0] = 0;
    print_len = 0;
    fell_through = 0;
}
