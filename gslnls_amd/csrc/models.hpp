// models.hpp -- device row models: the lowered form of the R closures .fn/.jac/.fvv.
//
// In the reference the model is an R closure evaluated with Rf_eval on every call
// (gsl_f / gsl_df / gsl_fvv, src/nls.c:815-978).  A GPU cannot call back into R, so
// the boundary lowers the model to a row function: value, gradient and second
// directional derivative of the model at ONE observation, given theta and that row's
// regressors.  Residual, weighting, the +Inf rule for non-finite values
// (src/nls.c:854-855) and finite differencing (src/fdjac.c, src/fdfvv.c) are applied
// around these by rowops.hpp, identically for every model.
//
// Registry ids are part of the C ABI (include/gslnls_core.h, GSLNLS_MODEL_*).
#pragma once
#include "lm_core.hpp"
#include "devmath.hpp"

namespace gslnls
{

// y ~ A*exp(-lam*x) + b            (R/nls.R:143-151, README.md:157-182; BASELINE C2)
struct ModelExpDecay
{
    static constexpr int ID = 1, P = 3, NX = 1;
    static constexpr bool HAS_FVV = true;
    GSLNLS_HD static double value(const double *th, const double *xr)
    {
        return th[0] * gexp(-th[1] * xr[0]) + th[2];
    }
    GSLNLS_HD static double value_grad(const double *th, const double *xr, double *g)
    {
        const double e = gexp(-th[1] * xr[0]);
        g[0] = e;
        g[1] = -th[0] * xr[0] * e;
        g[2] = 1.0;
        return th[0] * e + th[2];
    }
    GSLNLS_HD static double fvv(const double *th, const double *v, const double *xr)
    {
        const double x = xr[0], e = gexp(-th[1] * x);
        return 2.0 * v[0] * v[1] * (-x * e) + v[1] * v[1] * th[0] * x * x * e;
    }
};

// y ~ b1*(1-exp(-b2*x))            (Misra1a R/nls_test.R:174, BoxBOD :793; BASELINE C1, C4)
struct ModelMisra1a
{
    static constexpr int ID = 2, P = 2, NX = 1;
    static constexpr bool HAS_FVV = true;
    GSLNLS_HD static double value(const double *th, const double *xr)
    {
        return th[0] * (1.0 - gexp(-th[1] * xr[0]));
    }
    GSLNLS_HD static double value_grad(const double *th, const double *xr, double *g)
    {
        const double e = gexp(-th[1] * xr[0]);
        g[0] = 1.0 - e;
        g[1] = th[0] * xr[0] * e;
        return th[0] * (1.0 - e);
    }
    GSLNLS_HD static double fvv(const double *th, const double *v, const double *xr)
    {
        const double x = xr[0], e = gexp(-th[1] * x);
        return 2.0 * v[0] * v[1] * (x * e) + v[1] * v[1] * (-th[0] * x * x * e);
    }
};

// y ~ a*exp(-(x-b)^2/(2*c^2))      (README.md:545, example 2)
struct ModelGaussPeak
{
    static constexpr int ID = 3, P = 3, NX = 1;
    static constexpr bool HAS_FVV = true;
    GSLNLS_HD static double value(const double *th, const double *xr)
    {
        const double u = xr[0] - th[1];
        return th[0] * gexp(-(u * u) / (2.0 * th[2] * th[2]));
    }
    GSLNLS_HD static double value_grad(const double *th, const double *xr, double *g)
    {
        const double u = xr[0] - th[1], c2 = th[2] * th[2];
        const double e = gexp(-(u * u) / (2.0 * c2));
        g[0] = e;
        g[1] = th[0] * e * u / c2;
        g[2] = th[0] * e * u * u / (c2 * th[2]);
        return th[0] * e;
    }
    GSLNLS_HD static double fvv(const double *th, const double *v, const double *xr)
    {
        const double a = th[0], c = th[2], u = xr[0] - th[1], c2 = c * c;
        const double e = gexp(-(u * u) / (2.0 * c2));
        const double fab = e * u / c2;
        const double fac = e * u * u / (c2 * c);
        const double fbb = a * e * (u * u / (c2 * c2) - 1.0 / c2);
        const double fbc = a * e * (u * u * u / (c2 * c2 * c) - 2.0 * u / (c2 * c));
        const double fcc = a * e * (u * u * u * u / (c2 * c2 * c2) - 3.0 * u * u / (c2 * c2));
        return 2.0 * v[0] * v[1] * fab + 2.0 * v[0] * v[2] * fac + v[1] * v[1] * fbb + 2.0 * v[1] * v[2] * fbc +
               v[2] * v[2] * fcc;
    }
};

// y ~ b1*exp(-b2*x) + b3*exp(-(x-b4)^2/b5^2) + b6*exp(-(x-b7)^2/b8^2)
//                                   (NIST Gauss1/2/3, R/nls_test.R:301; BASELINE C5)
struct ModelGauss1
{
    static constexpr int ID = 4, P = 8, NX = 1;
    static constexpr bool HAS_FVV = false;
    // The quotients by b5^2, b8^2, b5^3, b8^3 are written as products with reciprocals that depend on the
    // parameters only: inside a row loop the compiler hoists them, which removes six fp64 divisions (~190 of the
    // ~400 VALU instructions) from every row.  Against the literal formula this moves results by <= 1 ulp per term.
    GSLNLS_HD static double value(const double *th, const double *xr)
    {
        const double x = xr[0], u1 = x - th[3], u2 = x - th[6];
        const double i4 = 1.0 / (th[4] * th[4]), i7 = 1.0 / (th[7] * th[7]);
        return th[0] * gexp(-th[1] * x) + th[2] * gexp(-(u1 * u1) * i4) + th[5] * gexp(-(u2 * u2) * i7);
    }
    GSLNLS_HD static double value_grad(const double *th, const double *xr, double *g)
    {
        const double x = xr[0], u1 = x - th[3], u2 = x - th[6];
        const double i4 = 1.0 / (th[4] * th[4]), i7 = 1.0 / (th[7] * th[7]);
        const double c4 = 2.0 * th[2] * i4, c7 = 2.0 * th[5] * i7, d4 = c4 / th[4], d7 = c7 / th[7];
        const double e0 = gexp(-th[1] * x);
        const double e1 = gexp(-(u1 * u1) * i4);
        const double e2 = gexp(-(u2 * u2) * i7);
        const double w1 = e1 * u1, w2 = e2 * u2;
        g[0] = e0;
        g[1] = -th[0] * x * e0;
        g[2] = e1;
        g[3] = c4 * w1;
        g[4] = d4 * (w1 * u1);
        g[5] = e2;
        g[6] = c7 * w2;
        g[7] = d7 * (w2 * u2);
        return th[0] * e0 + th[2] * e1 + th[5] * e2;
    }
    GSLNLS_HD static double fvv(const double *, const double *, const double *) { return NAN; }
};

} // namespace gslnls
