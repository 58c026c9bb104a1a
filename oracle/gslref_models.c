/*
 * gslref_models.c -- ORACLE (test infrastructure, never shipped).
 * Plain-C row models standing in for the R closures .fn/.jac/.fvv that the
 * reference evaluates through gsl_f / gsl_df / gsl_fvv (src/nls.c:815-978):
 * f_i = model(theta, x_i) - y_i with non-finite model values mapped to +Inf
 * (src/nls.c:854-855); Jacobian row-major n x p.  Used where a Python callback
 * would dominate a CPU-baseline timing, and as the CPU twin of the device
 * model registry.
 */
#include <math.h>
#include <stdlib.h>
#include "gslref.h"

/*
 * Opt-in exp for the row models (gslref_set_device_exp): the arithmetic of the device's exp, operation for operation --
 * k = rint(x log2 e), two-step Cody-Waite reduction, degree-11 polynomial by fused multiply-adds, ldexp; the constants
 * are the device library's (the product restates them in gslnls_amd/csrc/devmath.hpp::gexp).  glibc's exp is correctly
 * rounded in more cases than this scheme and differs from it in the last bit for a few per cent of the arguments; a
 * finite-difference Jacobian amplifies that bit by 1 / h ~ 7e7, which is why tests that compare FD trajectories with
 * the default exp can only ask for iteration counts "within the round-off tail".  With this switch on, oracle and device
 * evaluate the same function and FD runs can be held to EXACT iteration and evaluation counts.  Never on by default:
 * the oracle's job is to restate the reference, which calls libm.
 */
static int g_device_exp = 0;
void gslref_set_device_exp(int on) { g_device_exp = on; }
double gslref_device_exp(double x)
{
    const double xc = fmin(fmax(x, -1100.0), 1100.0);
    const double k = rint(xc * 0x1.71547652b82fep+0);
    double r = fma(k, -0x1.62e42fefa39efp-1, xc);
    r = fma(k, -0x1.abc9e3b39803fp-56, r);
    double p = fma(r, 0x1.ade156a5dcb37p-26, 0x1.28af3fca7ab0cp-22);
    p = fma(p, r, 0x1.71dee623fde64p-19);
    p = fma(p, r, 0x1.a01997c89e6b0p-16);
    p = fma(p, r, 0x1.a01a014761f6ep-13);
    p = fma(p, r, 0x1.6c16c1852b7b0p-10);
    p = fma(p, r, 0x1.1111111122322p-7);
    p = fma(p, r, 0x1.55555555502a1p-5);
    p = fma(p, r, 0x1.5555555555511p-3);
    p = fma(p, r, 0x1.000000000000bp-1);
    p = fma(r, p, 1.0);
    p = fma(r, p, 1.0);
    const double e = ldexp(p, (int)k);
    return (x != x) ? x : e;
}
void gslref_device_exp_array(const double *x, double *out, int n)
{
    for (int i = 0; i < n; ++i)
        out[i] = gslref_device_exp(x[i]);
}
static double model_exp(double x) { return g_device_exp ? gslref_device_exp(x) : exp(x); }

static double model_val(const gslref_rowdata *d, const double *th, int i)
{
    const double *X = d->x;
    const int n = d->n;
    switch (d->model)
    {
    case GSLREF_MODEL_EXPDECAY:
        /* (the opt-in device arithmetic covers the contraction as well: the device compiles A * e + b into one fused
         * multiply-add, gcc on x86-64 rounds the product first -- a last-bit difference of the residual that a difference
         * Jacobian amplifies by 1 / h) */
        if (g_device_exp)
            return fma(th[0], model_exp(-th[1] * X[i]), th[2]);
        return th[0] * model_exp(-th[1] * X[i]) + th[2];
    case GSLREF_MODEL_MISRA1A:
        return th[0] * (1.0 - model_exp(-th[1] * X[i]));
    case GSLREF_MODEL_GAUSSPK:
    {
        const double u = X[i] - th[1];
        return th[0] * model_exp(-(u * u) / (2.0 * th[2] * th[2]));
    }
    case GSLREF_MODEL_GAUSS1:
    {
        const double x = X[i];
        const double u1 = x - th[3], u2 = x - th[6];
        return th[0] * model_exp(-th[1] * x) + th[2] * model_exp(-(u1 * u1) / (th[4] * th[4])) +
               th[5] * model_exp(-(u2 * u2) / (th[7] * th[7]));
    }
    case GSLREF_MODEL_GLMEXP:
    {
        double s = 0.0;
        int j;
        for (j = 0; j < d->p; ++j)
            s += X[i + (size_t)n * j] * th[j];
        return model_exp(s);
    }
    default:
        return NAN;
    }
}

int gslref_model_f(const double *th, void *params, double *f)
{
    const gslref_rowdata *d = (const gslref_rowdata *)params;
    int i;
    for (i = 0; i < d->n; ++i)
    {
        const double m = model_val(d, th, i);
        f[i] = isfinite(m) ? m - d->y[i] : INFINITY;
    }
    return GSLREF_SUCCESS;
}

static void model_grad(const gslref_rowdata *d, const double *th, int i, double *g)
{
    const double *X = d->x;
    const int n = d->n;
    switch (d->model)
    {
    case GSLREF_MODEL_EXPDECAY:
    {
        const double e = model_exp(-th[1] * X[i]);
        g[0] = e;
        g[1] = -th[0] * X[i] * e;
        g[2] = 1.0;
        break;
    }
    case GSLREF_MODEL_MISRA1A:
    {
        const double e = model_exp(-th[1] * X[i]);
        g[0] = 1.0 - e;
        g[1] = th[0] * X[i] * e;
        break;
    }
    case GSLREF_MODEL_GAUSSPK:
    {
        const double u = X[i] - th[1], c2 = th[2] * th[2];
        const double e = model_exp(-(u * u) / (2.0 * c2));
        g[0] = e;
        g[1] = th[0] * e * u / c2;
        g[2] = th[0] * e * u * u / (c2 * th[2]);
        break;
    }
    case GSLREF_MODEL_GAUSS1:
    {
        const double x = X[i];
        const double u1 = x - th[3], u2 = x - th[6];
        const double e0 = model_exp(-th[1] * x);
        const double e1 = model_exp(-(u1 * u1) / (th[4] * th[4]));
        const double e2 = model_exp(-(u2 * u2) / (th[7] * th[7]));
        g[0] = e0;
        g[1] = -th[0] * x * e0;
        g[2] = e1;
        g[3] = th[2] * e1 * 2.0 * u1 / (th[4] * th[4]);
        g[4] = th[2] * e1 * 2.0 * u1 * u1 / (th[4] * th[4] * th[4]);
        g[5] = e2;
        g[6] = th[5] * e2 * 2.0 * u2 / (th[7] * th[7]);
        g[7] = th[5] * e2 * 2.0 * u2 * u2 / (th[7] * th[7] * th[7]);
        break;
    }
    case GSLREF_MODEL_GLMEXP:
    {
        double s = 0.0;
        int j;
        for (j = 0; j < d->p; ++j)
            s += X[i + (size_t)n * j] * th[j];
        s = model_exp(s);
        for (j = 0; j < d->p; ++j)
            g[j] = s * X[i + (size_t)n * j];
        break;
    }
    default:
        break;
    }
}

int gslref_model_df(const double *th, void *params, double *J)
{
    const gslref_rowdata *d = (const gslref_rowdata *)params;
    int i, j;
    for (i = 0; i < d->n; ++i)
    {
        model_grad(d, th, i, J + (size_t)i * d->p);
        for (j = 0; j < d->p; ++j)
            if (!isfinite(J[(size_t)i * d->p + j]))
                return GSLREF_EBADFUNC; /* src/nls.c:899-907 */
    }
    return GSLREF_SUCCESS;
}

/* second directional derivative v^T H_i v (analytic, as deriv(..., hessian=TRUE) would give) */
int gslref_model_fvv(const double *th, const double *v, void *params, double *fvv)
{
    const gslref_rowdata *d = (const gslref_rowdata *)params;
    const double *X = d->x;
    int i;
    for (i = 0; i < d->n; ++i)
    {
        double r;
        switch (d->model)
        {
        case GSLREF_MODEL_EXPDECAY:
        {
            const double x = X[i], e = model_exp(-th[1] * x);
            /* H: d2/dA dlam = -x e ; d2/dlam2 = A x^2 e */
            r = 2.0 * v[0] * v[1] * (-x * e) + v[1] * v[1] * th[0] * x * x * e;
            break;
        }
        case GSLREF_MODEL_MISRA1A:
        {
            const double x = X[i], e = model_exp(-th[1] * x);
            r = 2.0 * v[0] * v[1] * (x * e) + v[1] * v[1] * (-th[0] * x * x * e);
            break;
        }
        case GSLREF_MODEL_GAUSSPK:
        {
            const double a = th[0], c = th[2], u = X[i] - th[1], c2 = c * c;
            const double e = model_exp(-(u * u) / (2.0 * c2));
            const double fab = e * u / c2;
            const double fac = e * u * u / (c2 * c);
            const double fbb = a * e * (u * u / (c2 * c2) - 1.0 / c2);
            const double fbc = a * e * (u * u * u / (c2 * c2 * c) - 2.0 * u / (c2 * c));
            const double fcc = a * e * (u * u * u * u / (c2 * c2 * c2) - 3.0 * u * u / (c2 * c2));
            r = 2.0 * v[0] * v[1] * fab + 2.0 * v[0] * v[2] * fac + v[1] * v[1] * fbb +
                2.0 * v[1] * v[2] * fbc + v[2] * v[2] * fcc;
            break;
        }
        default:
            return GSLREF_EBADFUNC;
        }
        if (!isfinite(r))
            return GSLREF_EBADFUNC;
        fvv[i] = r;
    }
    return GSLREF_SUCCESS;
}

/* contract of gsl_df_large (src/nls_large.c:474-653) with the Jacobian re-evaluated per call */
int gslref_model_dfl(int trans, const double *th, const double *u, void *params, double *v, double *JTJ)
{
    const gslref_rowdata *d = (const gslref_rowdata *)params;
    const int n = d->n, p = d->p;
    double *g = (double *)malloc(sizeof(double) * p);
    int i, j, k;
    if (v)
    {
        if (trans)
            for (j = 0; j < p; ++j)
                v[j] = 0.0;
    }
    if (JTJ)
        for (j = 0; j < p * p; ++j)
            JTJ[j] = 0.0;
    for (i = 0; i < n; ++i)
    {
        model_grad(d, th, i, g);
        if (v)
        {
            if (!trans)
            {
                double s = 0.0;
                for (j = 0; j < p; ++j)
                    s += g[j] * u[j];
                v[i] = s;
            }
            else
                for (j = 0; j < p; ++j)
                    v[j] += g[j] * u[i];
        }
        if (JTJ)
            for (j = 0; j < p; ++j)
                for (k = 0; k <= j; ++k)
                    JTJ[j * p + k] += g[j] * g[k];
    }
    free(g);
    return GSLREF_SUCCESS;
}
