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

static double model_val(const gslref_rowdata *d, const double *th, int i)
{
    const double *X = d->x;
    const int n = d->n;
    switch (d->model)
    {
    case GSLREF_MODEL_EXPDECAY:
        return th[0] * exp(-th[1] * X[i]) + th[2];
    case GSLREF_MODEL_MISRA1A:
        return th[0] * (1.0 - exp(-th[1] * X[i]));
    case GSLREF_MODEL_GAUSSPK:
    {
        const double u = X[i] - th[1];
        return th[0] * exp(-(u * u) / (2.0 * th[2] * th[2]));
    }
    case GSLREF_MODEL_GAUSS1:
    {
        const double x = X[i];
        const double u1 = x - th[3], u2 = x - th[6];
        return th[0] * exp(-th[1] * x) + th[2] * exp(-(u1 * u1) / (th[4] * th[4])) +
               th[5] * exp(-(u2 * u2) / (th[7] * th[7]));
    }
    case GSLREF_MODEL_GLMEXP:
    {
        double s = 0.0;
        int j;
        for (j = 0; j < d->p; ++j)
            s += X[i + (size_t)n * j] * th[j];
        return exp(s);
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
        const double e = exp(-th[1] * X[i]);
        g[0] = e;
        g[1] = -th[0] * X[i] * e;
        g[2] = 1.0;
        break;
    }
    case GSLREF_MODEL_MISRA1A:
    {
        const double e = exp(-th[1] * X[i]);
        g[0] = 1.0 - e;
        g[1] = th[0] * X[i] * e;
        break;
    }
    case GSLREF_MODEL_GAUSSPK:
    {
        const double u = X[i] - th[1], c2 = th[2] * th[2];
        const double e = exp(-(u * u) / (2.0 * c2));
        g[0] = e;
        g[1] = th[0] * e * u / c2;
        g[2] = th[0] * e * u * u / (c2 * th[2]);
        break;
    }
    case GSLREF_MODEL_GAUSS1:
    {
        const double x = X[i];
        const double u1 = x - th[3], u2 = x - th[6];
        const double e0 = exp(-th[1] * x);
        const double e1 = exp(-(u1 * u1) / (th[4] * th[4]));
        const double e2 = exp(-(u2 * u2) / (th[7] * th[7]));
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
        s = exp(s);
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
            const double x = X[i], e = exp(-th[1] * x);
            /* H: d2/dA dlam = -x e ; d2/dlam2 = A x^2 e */
            r = 2.0 * v[0] * v[1] * (-x * e) + v[1] * v[1] * th[0] * x * x * e;
            break;
        }
        case GSLREF_MODEL_MISRA1A:
        {
            const double x = X[i], e = exp(-th[1] * x);
            r = 2.0 * v[0] * v[1] * (x * e) + v[1] * v[1] * (-th[0] * x * x * e);
            break;
        }
        case GSLREF_MODEL_GAUSSPK:
        {
            const double a = th[0], c = th[2], u = X[i] - th[1], c2 = c * c;
            const double e = exp(-(u * u) / (2.0 * c2));
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
