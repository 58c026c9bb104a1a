// lm_decide.hpp -- the scalar decisions of one trust-region trial, ONCE.
//
// Round 4 ended with the same Levenberg-Marquardt state machine written four times -- lm_core.hpp (template p <= 9, one
// wavefront, the headline kernel), wide_core.hpp (run-time p <= 64, lane = component), bd_host.hpp (host p-vectors, the
// Jacobian a matrix in HBM) and large_host.hpp (gsl_multilarge_nlinear) -- plus tests/hostsim, which compiles lm_core.hpp
// for the CPU.  The vector algebra of the four differs by construction (registers / lanes / device kernels / operators);
// what must NOT differ is the handful of scalar rules every variant applies to the outcome of a trial.  They live here,
// as inline functions of plain doubles that compile to the same instructions wherever they are inlined:
//
//   lmd_rho_of            trust_calc_rho with GSL's lm_preduction           src/trust.c:67-118
//   lmd_step_found        rho > 0, and the avmax gate of geodesic accel.    src/trust.c:474-486
//   lmd_radius            delta *= factor_up / delta /= factor_down         src/trust.c:488-496
//   lmd_nielsen_accept    mu *= max(1/3, 1 - (2 rho - 1)^3), nu = 2        src/trust.c:175-188
//   lmd_nielsen_reject    mu *= nu, nu *= 2; LMD_MAX_REJECTS: the 16th
//                         rejection in a row is "no progress"               src/trust.c:190-199, :530-545
//
// The expressions are the ones lm_core.hpp has carried since round 1 (same operations, same association): moving them
// here changes no bit anywhere -- checked by the ISA of lm_step_kernel<ModelExpDecay, 0, 512> before and after (no
// instruction differs) and by the bit-identity tests of the narrow, wide and matrix paths.
#pragma once
#if !defined(__HIPCC_RTC__)
#include <math.h>
#endif
#if !defined(GSLNLS_HD)
#if defined(__HIPCC__)
#define GSLNLS_HD __host__ __device__ __forceinline__
#else
#define GSLNLS_HD inline
#endif
#endif

namespace gslnls
{

// rho of a trial whose ||f_trial||^2 = ssr_t is below ||f||^2 = fnorm2: actual over predicted reduction, the prediction
// from vAv = v^T (J^T J) v and Dv2 = ||D v||^2 of the velocity v (lm_preduction: (||J v|| / ||f||)^2 + 2 mu (||D v|| / ||f||)^2)
GSLNLS_HD double lmd_rho_of(double ssr_t, double fnorm2, double vAv, double Dv2, double mu)
{
    const double finv = 1.0 / fnorm2;
    const double ared = 1.0 - ssr_t * finv;
    const double pred = vAv * finv + 2.0 * mu * (Dv2 * finv);
    return (pred > 0.0) ? ared / pred : -1.0;
}

// a step is taken when rho > 0 -- unless geodesic acceleration is on and |a| / |v| exceeds avmax
GSLNLS_HD bool lmd_step_found(double rho, int trs, double avratio, double avmax)
{
    bool found = rho > 0.0;
    if (trs == 1 && avratio > avmax)
        found = false;
    return found;
}

// the trust-region radius after a trial
GSLNLS_HD void lmd_radius(double rho, double factor_up, double factor_down, double &delta)
{
    if (rho > 0.75)
        delta *= factor_up;
    else if (rho < 0.25)
        delta /= factor_down;
}

// Nielsen's damping schedule: an accepted step ...
GSLNLS_HD void lmd_nielsen_accept(double rho, double &mu, double &nu)
{
    double b = 2.0 * rho - 1.0;
    b = 1.0 - b * b * b;
    nu = 2.0;
    mu *= fmax(0.333333333333333, b);
}

// ... and a rejected one.  The rejection that makes it more than LMD_MAX_REJECTS in a row ends the iteration with "no
// progress": `++bad_steps > LMD_MAX_REJECTS` stays spelled out at the call sites (as a function returning bool it cost the
// headline kernel five instructions of control flow -- ISA diffed, round 5)
constexpr int LMD_MAX_REJECTS = 15;
GSLNLS_HD void lmd_nielsen_reject(double &mu, double &nu)
{
    mu *= nu;
    nu *= 2.0;
}

} // namespace gslnls
