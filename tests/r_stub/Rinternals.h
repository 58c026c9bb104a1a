/*
 * Rinternals.h -- DECLARATION-ONLY stand-in for R's C API, for ONE purpose: `gcc -fsyntax-only -Wall` of the
 * reference-side shims in integration/r_shim/ inside an image that has no R.  It declares the
 * subset of the public API (R >= 4.1, "Writing R Extensions" section 5/6) those two files use, with R's
 * documented signatures.  The shims are compiled for real only inside the R package, against R's own headers; since round 5
 * tests/r_mini/rmini.c IMPLEMENTS the part of these declarations the dense shim's `function`-model route uses, so that the
 * shim can be executed by tests/test_gpu_r_shim.py / tests/test_r_shim_cpu.py.  Test infrastructure (tests/test_abi.py,
 * __graft_entry__.build()).
 */
#ifndef GSLNLS_TEST_RINTERNALS_STUB_H
#define GSLNLS_TEST_RINTERNALS_STUB_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct SEXPREC *SEXP;
typedef ptrdiff_t R_xlen_t;
typedef int R_len_t;
typedef enum { FALSE = 0, TRUE } Rboolean;
typedef unsigned int SEXPTYPE;
typedef void *(*DL_FUNC)(void);
typedef struct _DllInfo DllInfo; /* R_ext/Rdynload.h */

#define NILSXP 0
#define SYMSXP 1
#define LISTSXP 2
#define CLOSXP 3
#define ENVSXP 4
#define LANGSXP 6
#define LGLSXP 10
#define INTSXP 13
#define REALSXP 14
#define STRSXP 16
#define VECSXP 19

extern SEXP R_NilValue, R_UnboundValue, R_BaseEnv, R_GlobalEnv, R_NamesSymbol, R_DimNamesSymbol, R_DimSymbol;
extern double R_NaReal;
extern int R_NaInt;
#define NA_REAL R_NaReal
#define NA_INTEGER R_NaInt

SEXP Rf_protect(SEXP);
void Rf_unprotect(int);
#define PROTECT(s) Rf_protect(s)
#define UNPROTECT(n) Rf_unprotect(n)
typedef int PROTECT_INDEX;
void R_ProtectWithIndex(SEXP, PROTECT_INDEX *);
void R_Reprotect(SEXP, PROTECT_INDEX);
#define PROTECT_WITH_INDEX(x, i) R_ProtectWithIndex(x, i)
#define REPROTECT(x, i) R_Reprotect(x, i)
void R_PreserveObject(SEXP);
void R_ReleaseObject(SEXP);

SEXP Rf_allocVector(SEXPTYPE, R_xlen_t);
SEXP Rf_allocMatrix(SEXPTYPE, int, int);
SEXP Rf_mkNamed(SEXPTYPE, const char **);
SEXP Rf_mkString(const char *);
SEXP Rf_mkChar(const char *);
SEXP Rf_ScalarInteger(int);
SEXP Rf_ScalarReal(double);
SEXP Rf_ScalarString(SEXP);
SEXP Rf_coerceVector(SEXP, SEXPTYPE);
SEXP Rf_install(const char *);
SEXP Rf_lang2(SEXP, SEXP);
SEXP Rf_lang3(SEXP, SEXP, SEXP);
SEXP Rf_eval(SEXP, SEXP);
SEXP R_tryEval(SEXP, SEXP, int *);
SEXP SETCADR(SEXP, SEXP);
SEXP SETCADDR(SEXP, SEXP);
SEXP Rf_findVar(SEXP, SEXP);
SEXP Rf_findVarInFrame(SEXP, SEXP);
SEXP Rf_getAttrib(SEXP, SEXP);
SEXP Rf_setAttrib(SEXP, SEXP, SEXP);
SEXP Rf_GetOption1(SEXP);
SEXP R_do_slot(SEXP, SEXP);
R_len_t Rf_length(SEXP);
int Rf_ncols(SEXP);
int Rf_nrows(SEXP);
Rboolean Rf_isNull(SEXP);
Rboolean Rf_isMatrix(SEXP);
Rboolean Rf_isReal(SEXP);
Rboolean Rf_isString(SEXP);
Rboolean Rf_isFunction(SEXP);
Rboolean Rf_isEnvironment(SEXP);
Rboolean Rf_isNewList(SEXP);
Rboolean Rf_inherits(SEXP, const char *);
int TYPEOF(SEXP);

double *REAL(SEXP);
int *INTEGER(SEXP);
int *LOGICAL(SEXP);
const char *CHAR(SEXP);
SEXP STRING_ELT(SEXP, R_xlen_t);
SEXP VECTOR_ELT(SEXP, R_xlen_t);
SEXP SET_VECTOR_ELT(SEXP, R_xlen_t, SEXP);
void SET_STRING_ELT(SEXP, R_xlen_t, SEXP);
SEXP CLOENV(SEXP);
SEXP CAR(SEXP);
SEXP CDR(SEXP);
SEXP CADR(SEXP);
SEXP CADDR(SEXP);

void Rf_error(const char *, ...);
void Rf_warning(const char *, ...);
void Rf_onintr(void);
void R_CheckUserInterrupt(void);
Rboolean R_ToplevelExec(void (*fun)(void *), void *data);
char *R_alloc(size_t, int);

#ifdef __cplusplus
}
#endif
#endif
