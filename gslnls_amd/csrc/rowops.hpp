// rowops.hpp -- what one observation contributes to one pass.
//
// Device twin of the reference's evaluation layer, fused so that neither f nor J is
// ever stored:
//   residual + weighting      gsl_f src/nls.c:849-858, eval_f src/fdf.c:94-113
//   analytic Jacobian row     gsl_df src/nls.c:898-912, eval_df src/fdf.c:135-166
//   forward / central FD row  src/fdjac.c:24-64, :81-128 (delta_j = h|x_j|, 0 -> h;
//                             differences are taken on the WEIGHTED residual)
//   second directional deriv  src/fdf.c:200-233, FD form src/fdfvv.c:35-77
// and of the reductions that follow: ssr = f.f (src/nls_fit.c:75), g = J^T f
// (src/trust.c:331,:521), J^T J (dsyrk in GSL's cholesky solver / src/nls_utils.c:61).
#pragma once
#include "lm_core.hpp"

namespace gslnls
{

enum
{
    JAC_ANALYTIC = 0,
    JAC_FORWARD = 1,
    JAC_CENTER = 2
};

// weighted residual with the reference's non-finite rule
template <class M>
GSLNLS_HD double row_resid(const double *th, const double *xr, double y, double sw)
{
    const double m = M::value(th, xr);
    const double f = isfinite(m) ? m - y : INFINITY;
    return f * sw;
}

// finite-difference step sizes of one Jacobian evaluation (src/fdjac.c:36-38)
template <int P>
GSLNLS_HD void fd_deltas(const double *th, double h, double *delta)
{
#pragma unroll
    for (int j = 0; j < P; ++j)
    {
        double d = h * fabs(th[j]);
        if (d == 0.0)
            d = h;
        delta[j] = d;
    }
}

// f_i and row i of the (weighted) Jacobian at th.  Returns f_i; *nbad becomes non-zero (NaN) when an
// analytic Jacobian entry is not finite (only the analytic path is checked by the reference, which
// then returns GSL_EBADFUNC, src/nls.c:899-907).  The flag is accumulated as sum_j 0 * J_ij -- one FMA per
// entry instead of a class compare + select + add; consumers test `!(badj == 0)`.
template <class M, int JAC>
GSLNLS_HD double row_fj(const double *th, const double *delta, const double *xr, double y, double sw,
                        double *Jrow, double *nbad)
{
    constexpr int P = M::P;
    if (JAC == JAC_ANALYTIC)
    {
        double gr[P];
        const double m = M::value_grad(th, xr, gr);
        const double f = (isfinite(m) ? m - y : INFINITY) * sw;
        double bad = *nbad;
#pragma unroll
        for (int j = 0; j < P; ++j)
        {
            bad = fma(gr[j], 0.0, bad);
            Jrow[j] = gr[j] * sw;
        }
        *nbad = bad;
        return f;
    }
    else if (JAC == JAC_FORWARD)
    {
        const double f = row_resid<M>(th, xr, y, sw);
        double tp[P];
#pragma unroll
        for (int j = 0; j < P; ++j)
            tp[j] = th[j];
#pragma unroll
        for (int j = 0; j < P; ++j)
        {
            tp[j] = th[j] + delta[j];
            const double fn = row_resid<M>(tp, xr, y, sw);
            tp[j] = th[j];
            Jrow[j] = (fn - f) * (1.0 / delta[j]);
        }
        return f;
    }
    else
    {
        const double f = row_resid<M>(th, xr, y, sw);
        double tp[P];
#pragma unroll
        for (int j = 0; j < P; ++j)
            tp[j] = th[j];
#pragma unroll
        for (int j = 0; j < P; ++j)
        {
            tp[j] = th[j] + 0.5 * delta[j];
            const double fp = row_resid<M>(tp, xr, y, sw);
            tp[j] = th[j] - 0.5 * delta[j];
            const double fm = row_resid<M>(tp, xr, y, sw);
            tp[j] = th[j];
            Jrow[j] = (fp - fm) * (1.0 / delta[j]);
        }
        return f;
    }
}

// accumulate ssr, J^T J (packed lower) and J^T f of one row
template <int P>
GSLNLS_HD void acc_fj(PassSums<P> &a, double f, const double *Jrow)
{
    a.ssr += f * f;
#pragma unroll
    for (int i = 0; i < P; ++i)
    {
        a.g[i] += Jrow[i] * f;
#pragma unroll
        for (int j = 0; j <= i; ++j)
            a.A[tri(i, j)] += Jrow[i] * Jrow[j];
    }
}

// second directional derivative of row i (weighted) at th along v
template <class M, int JAC>
GSLNLS_HD double row_fvv(const double *th, const double *v, const double *delta, double h_fvv, bool analytic,
                         const double *xr, double y, double sw, double *Jrow, double *nbad)
{
    constexpr int P = M::P;
    double jb = 0.0;
    const double f = row_fj<M, JAC>(th, delta, xr, y, sw, Jrow, &jb);
    if (analytic)
    {
        const double r = M::fvv(th, v, xr);
        *nbad = fma(r, 0.0, *nbad);
        return r * sw;
    }
    double tp[P];
    double u = 0.0;
#pragma unroll
    for (int j = 0; j < P; ++j)
    {
        tp[j] = th[j] + h_fvv * v[j];
        u += Jrow[j] * v[j];
    }
    const double fip = row_resid<M>(tp, xr, y, sw);
    const double hinv = 1.0 / h_fvv;
    return (2.0 * hinv) * ((fip - f) * hinv - u);
}

template <int P>
GSLNLS_HD void pass_zero(PassSums<P> &a)
{
    a.ssr = 0.0;
    a.badj = 0.0;
#pragma unroll
    for (int k = 0; k < PassSums<P>::NA; ++k)
        a.A[k] = 0.0;
#pragma unroll
    for (int k = 0; k < P; ++k)
        a.g[k] = 0.0;
}

// flat view used by the reductions: v in [0, NV)
template <int P>
GSLNLS_HD double &pass_slot(PassSums<P> &a, int v)
{
    return reinterpret_cast<double *>(&a)[v];
}

} // namespace gslnls
