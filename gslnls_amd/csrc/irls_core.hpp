// irls_core.hpp -- robust-loss psi / psi' families evaluated on device (host+device inline).
//
// Device twin of psi() / psip() in src/nls_irls.c:10-341 (formulations from robustbase's lmrob.c).
// Index = loss_config$rho - 0: 1 huber, 2 barron, 3 bisquare, 4 welsh, 5 optimal, 6 hampel, 7 ggw,
// 8 lqq (R/nls.R:663).  cc[] are the tuning constants of gsl_nls_loss() (R/nls_rho.R:111-121).
#pragma once
#include "lm_core.hpp"

namespace gslnls
{

struct LossCfg
{
    int rho;      // 1..8
    double cc[3]; // tuning constants (unused entries 0)
};

GSLNLS_HD double irls_psi(double x, const LossCfg &L)
{
    const double *c = L.cc;
    switch (L.rho)
    {
    case 2: // Barron family (alpha, c)
    {
        const double alpha = c[0], c2 = c[1] * c[1], z = (x * x) / c2;
        if (fabs(alpha - 2.0) < 1.4901161193847656e-08)
            return x / c2;
        if (fabs(alpha) < 1.4901161193847656e-08)
            return 2.0 * x / (x * x + 2 * c2);
        if (alpha > -1e8)
            return x / c2 * pow((z / fabs(alpha - 2.0) + 1), 0.5 * alpha - 1.0);
        return x / c2 * exp(-0.5 * z);
    }
    case 3: // Tukey bisquare
    {
        if (fabs(x) > c[0])
            return 0.;
        const double a = x / c[0], u = 1. - a * a;
        return x * u * u;
    }
    case 4: // Welsh
    {
        const double a = x / c[0];
        return fabs(a) > 37.7 ? 0. : x * exp(-(a * a) / 2);
    }
    case 5: // optimal
    {
        const double ac = x / c[0], ax = fabs(ac);
        if (ax > 3.)
            return 0.;
        if (ax > 2.)
        {
            const double a2 = ac * ac;
            const double poly = c[0] * ((((0.016 * a2 + -0.312) * a2 + 1.728) * a2 + -1.944) * ac);
            return ac > 0. ? fmax(0., poly) : -fabs(poly);
        }
        return x;
    }
    case 6: // Hampel (1.5, 3.5, 8) k
    {
        const double a = 1.5 * c[0], b = 3.5 * c[0], r = 8.0 * c[0];
        const double sgn = x < 0 ? -1. : 1., u = fabs(x);
        if (u <= a)
            return x;
        if (u <= b)
            return sgn * a;
        if (u <= r)
            return sgn * a * (r - u) / (r - b);
        return 0.;
    }
    case 7: // GGW (a, b, c)
    {
        const double ax = fabs(x);
        if (ax < c[2])
            return x;
        const double e = -pow(ax - c[2], c[1]) / 2 / c[0];
        return e < -708.4 ? 0. : x * exp(e);
    }
    case 8: // LQQ (b, c, s)
    {
        const double ax = fabs(x);
        if (ax <= c[1])
            return x;
        const double k01 = c[0] + c[1];
        if (ax <= k01)
            return (double)(x > 0 ? 1 : (x < 0 ? -1 : 0)) * (ax - c[2] * pow(ax - c[1], 2.) / c[0] / 2.);
        const double s5 = c[2] - 1., s6 = -2 * k01 + c[0] * c[2];
        if (ax < k01 - s6 / s5)
            return (double)(x > 0 ? 1 : -1) *
                   (-s6 / 2. - pow(s5, 2.) / s6 * (pow(ax - k01, 2.) / 2. + s6 / s5 * (ax - k01)));
        return 0.;
    }
    default: // Huber
        return x <= -c[0] ? -c[0] : (x < c[0] ? x : c[0]);
    }
}

GSLNLS_HD double irls_psip(double x, const LossCfg &L)
{
    const double *c = L.cc;
    switch (L.rho)
    {
    case 2:
    {
        const double alpha = c[0], c2 = c[1] * c[1], x2 = x * x;
        if (fabs(alpha - 2.0) < 1.4901161193847656e-08)
            return 1.0 / c2;
        if (fabs(alpha) < 1.4901161193847656e-08)
            return -2. * (x2 - 2. * c2) / ((2. * c2 + x2) * (2. * c2 + x2));
        if (alpha > -1e8)
        {
            const double den = x2 - (alpha - 2.) * c2;
            return (alpha - 2.) * ((alpha - 2.) * c2 - (alpha - 1.) * x2) *
                   pow(1. - x2 / ((alpha - 2.) * c2), 0.5 * alpha) / (den * den);
        }
        return exp(-x2 / (2. * c2)) * (c2 - x2) / (c2 * c2);
    }
    case 3:
    {
        if (fabs(x) > c[0])
            return 0.;
        const double a = x / c[0], a2 = a * a;
        return (1. - a2) * (1 - 5 * a2);
    }
    case 4:
    {
        const double a = x / c[0];
        if (fabs(a) > 37.7)
            return 0.;
        const double a2 = a * a;
        return exp(-a2 / 2) * (1. - a2);
    }
    case 5:
    {
        double ax = fabs(x / c[0]);
        if (ax > 3.)
            return 0.;
        if (ax > 2.)
        {
            ax *= ax;
            return -1.944 + ax * (3 * 1.728 + ax * (5 * -0.312 + ax * 7 * 0.016));
        }
        return 1.;
    }
    case 6:
    {
        const double a = 1.5 * c[0], b = 3.5 * c[0], r = 8.0 * c[0], u = fabs(x);
        if (u <= a)
            return 1.;
        if (u <= b)
            return 0.;
        if (u <= r)
            return a / (b - r);
        return 0.;
    }
    case 7:
    {
        const double ax = fabs(x);
        if (ax < c[2])
            return 1.;
        const double a = 2 * c[0], b = c[1], cc = c[2];
        const double e = -pow(ax - cc, b) / a;
        return e < -708.4 ? 0. : exp(e) * (1 - b / a * ax * pow(ax - cc, b - 1));
    }
    case 8:
    {
        const double ax = fabs(x);
        if (ax <= c[1])
            return 1.;
        const double k01 = c[0] + c[1];
        if (ax <= k01)
            return 1. - c[2] / c[0] * (ax - c[1]);
        const double s5 = 1. - c[2], a = (c[0] * c[2] - 2 * k01) / s5;
        if (ax < k01 + a)
            return -s5 * ((ax - k01) / a - 1.);
        return 0.;
    }
    default:
        return fabs(x) >= c[0] ? 0. : 1.;
    }
}

} // namespace gslnls
