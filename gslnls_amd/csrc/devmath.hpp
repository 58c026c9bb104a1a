// devmath.hpp -- exp() for the row models.
//
// The fp64 exp of the device library is the classic scheme (k = rint(x log2 e), two-step Cody-Waite
// reduction, degree-11 polynomial, ldexp) -- but the compiler materialises every polynomial coefficient
// with two v_mov_b32 into the accumulator of a v_fmac (3 VALU instructions per Horner step; measured: 31 of
// the 85 VALU instructions a C2 row costs were moves).  Here the same arithmetic -- same constants, same
// order, so the same bits for every finite argument -- keeps the coefficients in SGPR pairs and issues one
// v_fma_f64 per step.  fp64 VALU issue is what bounds the row loops, so this is a direct
// cut of the pass time.  Host compilation (tests/hostsim) uses the C library.
#pragma once
#if !defined(__HIPCC_RTC__)
#include <cmath>
#endif
#include "lm_core.hpp"

namespace gslnls
{

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double gslnls_horner(double p, double r, double c)
{
    double o;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(o) : "v"(p), "v"(r), "s"(c));
    return o;
}

#endif

GSLNLS_HD double gexp(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    // clamping keeps k inside int32 and lets v_ldexp_f64 produce 0 / Inf; fmax/fmin drop a NaN, restored below
    const double xc = fmin(fmax(x, -1100.0), 1100.0);
    const double k = rint(xc * 0x1.71547652b82fep+0);              // log2(e)
    double r = fma(k, -0x1.62e42fefa39efp-1, xc);                  // - ln2 (high part)
    r = fma(k, -0x1.abc9e3b39803fp-56, r);                         // - ln2 (low part)
    double p = fma(r, 0x1.ade156a5dcb37p-26, 0x1.28af3fca7ab0cp-22);
    p = gslnls_horner(p, r, 0x1.71dee623fde64p-19);
    p = gslnls_horner(p, r, 0x1.a01997c89e6b0p-16);
    p = gslnls_horner(p, r, 0x1.a01a014761f6ep-13);
    p = gslnls_horner(p, r, 0x1.6c16c1852b7b0p-10);
    p = gslnls_horner(p, r, 0x1.1111111122322p-7);
    p = gslnls_horner(p, r, 0x1.55555555502a1p-5);
    p = gslnls_horner(p, r, 0x1.5555555555511p-3);
    p = gslnls_horner(p, r, 0x1.000000000000bp-1);
    p = fma(r, p, 1.0);
    p = fma(r, p, 1.0);
    double e = ldexp(p, (int)k);
    return (x != x) ? x : e;
#else
    return exp(x);
#endif
}

} // namespace gslnls
