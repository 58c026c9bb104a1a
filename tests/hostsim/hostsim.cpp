// hostsim.cpp -- TEST-ONLY harness (never linked into libgslnls_hip.so).
//
// Compiles the SAME headers the device kernels use (lm_core.hpp, rowops.hpp, models.hpp)
// with g++ and replaces the GPU's parallel pass by a serial loop, so the device state
// machine and row functions can be checked against the oracle on a CPU-only box.  This
// exercises "host logic" only; GPU parity proper is tests/test_gpu_*.py through the C ABI.
#include <math.h>
#include <string.h>
#include "lm_core.hpp"
#include "models.hpp"
#include "rowops.hpp"

using namespace gslnls;

template <class M, int JAC>
static void pass(const LmState<M::P> &s, const LmParams &prm, int n, const double *x, const double *y,
                 const double *sw, PassSums<M::P> &acc)
{
    constexpr int P = M::P;
    double th[P], delta[P];
    for (int k = 0; k < P; ++k)
        th[k] = (s.phase == PH_FVV) ? s.x[k] : s.xt[k];
    fd_deltas<P>(th, prm.h_df, delta);
    pass_zero<P>(acc);
    for (int i = 0; i < n; ++i)
    {
        double xr[M::NX];
        for (int c = 0; c < M::NX; ++c)
            xr[c] = x[i + (size_t)n * c];
        double Jrow[P];
        const double w = sw ? sw[i] : 1.0;
        if (s.phase == PH_FVV)
        {
            const double fv = row_fvv<M, JAC>(th, s.vel, delta, prm.h_fvv, prm.fvv_analytic != 0, xr, y[i], w, Jrow,
                                              &acc.badj);
            for (int k = 0; k < P; ++k)
                acc.g[k] += Jrow[k] * fv;
        }
        else
        {
            const double f = row_fj<M, JAC>(th, delta, xr, y[i], w, Jrow, &acc.badj);
            acc_fj<P>(acc, f, Jrow);
        }
    }
}

template <class M>
static int fit(int n, const double *x, const double *y, const double *sw, const double *start, const double *lupars,
               const int *ci, const double *cd, int jac, int fvv, double *par, int *ints, double *dbls, double *covar,
               double *ssrtrace, double *partrace)
{
    constexpr int P = M::P;
    LmParams prm;
    prm.maxiter = ci[0];
    prm.trs = (ci[2] == 1) ? 1 : 0;
    prm.scale = ci[3];
    prm.fdtype = ci[5] ? 1 : 0;
    prm.jac_analytic = jac;
    prm.fvv_analytic = fvv;
    prm.has_bounds = lupars != nullptr;
    prm.has_weights = sw != nullptr;
    prm.bench_hold = 0;
    prm.factor_up = cd[0];
    prm.factor_down = cd[1];
    prm.avmax = cd[2];
    prm.h_df = cd[3];
    prm.h_fvv = cd[4];
    prm.xtol = cd[5];
    prm.ftol = cd[6];
    prm.gtol = cd[7];
    LmState<P> s;
    lm_state_reset<P>(s, start, lupars);
    const int jacmode = jac ? JAC_ANALYTIC : (ci[5] ? JAC_CENTER : JAC_FORWARD);
    long launches = 0;
    while (s.phase != PH_DONE && launches < 1000000)
    {
        PassSums<P> acc;
        if (jacmode == JAC_ANALYTIC)
            pass<M, JAC_ANALYTIC>(s, prm, n, x, y, sw, acc);
        else if (jacmode == JAC_FORWARD)
            pass<M, JAC_FORWARD>(s, prm, n, x, y, sw, acc);
        else
            pass<M, JAC_CENTER>(s, prm, n, x, y, sw, acc);
        const int nb = s.niter, pb = s.phase;
        lm_advance<P>(s, acc, prm);
        if (ssrtrace)
        {
            if (pb == PH_INIT)
            {
                ssrtrace[0] = s.chisq_init;
                for (int k = 0; k < P; ++k)
                    partrace[(size_t)(prm.maxiter + 1) * k] = s.x[k];
            }
            else if (s.niter != nb && s.status != ST_EBADFUNC && !(s.status == ST_ENOPROG && nb == 0))
            {
                ssrtrace[s.niter] = s.chisq1;
                for (int k = 0; k < P; ++k)
                    partrace[s.niter + (size_t)(prm.maxiter + 1) * k] = s.x[k];
            }
        }
        ++launches;
    }
    for (int k = 0; k < P; ++k)
        par[k] = s.x[k];
    ints[0] = s.niter;
    ints[1] = s.status;
    ints[2] = s.info;
    ints[3] = s.nevalf;
    ints[4] = s.nevaldf;
    ints[5] = s.nevalfvv;
    ints[6] = (int)launches;
    dbls[0] = s.chisq1;
    dbls[1] = s.chisq0 - s.chisq1;
    dbls[2] = s.chisq_init;
    dbls[3] = s.mu;
    dbls[4] = s.delta;
    dbls[5] = det_cholesky<P>(s.A);
    if (covar)
        covar_from_jtj<P>(s.A, covar);
    return s.status;
}

extern "C" int hostsim_fit(int model, int n, const double *x, const double *y, const double *sw, const double *start,
                           const double *lupars, const int *ci, const double *cd, int jac, int fvv, double *par,
                           int *ints, double *dbls, double *covar, double *ssrtrace, double *partrace)
{
    switch (model)
    {
    case 1:
        return fit<ModelExpDecay>(n, x, y, sw, start, lupars, ci, cd, jac, fvv, par, ints, dbls, covar, ssrtrace, partrace);
    case 2:
        return fit<ModelMisra1a>(n, x, y, sw, start, lupars, ci, cd, jac, fvv, par, ints, dbls, covar, ssrtrace, partrace);
    case 3:
        return fit<ModelGaussPeak>(n, x, y, sw, start, lupars, ci, cd, jac, fvv, par, ints, dbls, covar, ssrtrace, partrace);
    case 4:
        return fit<ModelGauss1>(n, x, y, sw, start, lupars, ci, cd, jac, fvv, par, ints, dbls, covar, ssrtrace, partrace);
    default:
        return -101;
    }
}

// direct access to the device linear algebra for unit tests
extern "C" void hostsim_lm_solve3(const double *Ap, const double *diag, double mu, const double *rhs, double *sol)
{
    lm_solve<3>(Ap, diag, mu, rhs, sol);
}
extern "C" void hostsim_lm_solve8(const double *Ap, const double *diag, double mu, const double *rhs, double *sol)
{
    lm_solve<8>(Ap, diag, mu, rhs, sol);
}
