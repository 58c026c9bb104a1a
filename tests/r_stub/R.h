/* R.h -- see Rinternals.h in this directory: declaration-only stand-in used to syntax-check integration/r_shim/ */
#ifndef GSLNLS_TEST_R_STUB_H
#define GSLNLS_TEST_R_STUB_H
#include <stdlib.h>
#include <stdio.h>
#include <math.h>
void Rprintf(const char *, ...);
#endif
