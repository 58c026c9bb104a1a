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

// ---- the gamma family of stats::deriv's table (R/nls.R:588-599 differentiates the formula with it: gamma, lgamma,
// digamma, trigamma, psigamma) ----------------------------------------------------------------------------------------
// psi^(n)(x), n = 0 (digamma) .. 4, R's psigamma(x, n): for x >= 1/2 the recurrence
//     psi^(n)(x) = psi^(n)(x + 1) - (-1)^n n! / x^(n+1)
// up to x >= 16, then the asymptotic series (Abramowitz & Stegun 6.3.18, 6.4.11)
//     psi(x)     ~ log x - 1/(2x) - sum_k B_2k / (2k x^2k)
//     psi^(n)(x) ~ (-1)^(n+1) [ (n-1)!/x^n + n!/(2 x^(n+1)) + sum_k B_2k (2k+n-1)!/((2k)! x^(2k+n)) ]
// with eight Bernoulli terms (relative truncation error below 1e-17 at x = 16); for x < 1/2 the reflection
//     psi^(n)(1 - x) = (-1)^n [ psi^(n)(x) + d^n/dx^n (pi cot(pi x)) ].
// Poles (x = 0, -1, ...) give NaN / Inf as R's do.  n outside 0..4: NaN.
GSLNLS_HD double gpsigamma(double x, int n)
{
    if (n < 0 || n > 4 || x != x)
        return NAN;
    const double fact[5] = {1.0, 1.0, 2.0, 6.0, 24.0};
    const bool reflect = x < 0.5;
    double refl = 0.0; // d^n/dx^n (pi cot(pi x)) in terms of c = cot(pi x)
    if (reflect)
    {
        const double pi = 3.14159265358979323846, c = 1.0 / tan(pi * x), c2 = c * c;
        switch (n)
        {
        case 0: refl = pi * c; break;
        case 1: refl = -pi * pi * (1.0 + c2); break;
        case 2: refl = 2.0 * pi * pi * pi * c * (1.0 + c2); break;
        case 3: refl = -2.0 * pi * pi * pi * pi * (1.0 + 3.0 * c2) * (1.0 + c2); break;
        default: refl = 8.0 * pi * pi * pi * pi * pi * c * (2.0 + 3.0 * c2) * (1.0 + c2); break;
        }
        x = 1.0 - x;
    }
    const double sgn = (n & 1) ? -1.0 : 1.0; // (-1)^n
    double acc = 0.0;
    while (x < 16.0)
    {
        double pw = x;
        for (int k = 0; k < n; ++k)
            pw *= x;
        acc -= sgn * fact[n] / pw;
        x += 1.0;
    }
    // B_2k, k = 1..8
    const double B[8] = {1.0 / 6.0, -1.0 / 30.0, 1.0 / 42.0, -1.0 / 30.0, 5.0 / 66.0, -691.0 / 2730.0, 7.0 / 6.0, -3617.0 / 510.0};
    const double xi = 1.0 / x, xi2 = xi * xi;
    double r;
    if (n == 0)
    {
        double s = 0.0, pw = 1.0;
        for (int k = 1; k <= 8; ++k)
        {
            pw *= xi2;
            s += B[k - 1] / (2.0 * k) * pw;
        }
        r = acc + (log(x) - 0.5 * xi - s);
    }
    else
    {
        // (2k + n - 1)! / (2k)! = (2k + 1)(2k + 2) ... (2k + n - 1)
        double xn = 1.0; // x^-n
        for (int k = 0; k < n; ++k)
            xn *= xi;
        double s = fact[n - 1] * xn + 0.5 * fact[n] * xn * xi, pw = xn;
        for (int k = 1; k <= 8; ++k)
        {
            pw *= xi2;
            double c = 1.0;
            for (int j = 1; j < n; ++j)
                c *= (double)(2 * k + j);
            s += B[k - 1] * c * pw;
        }
        r = acc - sgn * s;
    }
    // psi^(n)(x) = (-1)^n psi^(n)(1 - x) - d^n/dx^n (pi cot(pi x))
    return reflect ? sgn * r - refl : r;
}

} // namespace gslnls
