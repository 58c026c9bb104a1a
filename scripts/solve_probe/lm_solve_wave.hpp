// lm_solve_wave<P>, 5 <= P <= 8: the damped modified-Cholesky solve of lm_solve (lm_core.hpp) with the p x p working
// matrix spread over the 64 lanes of ONE fully active wavefront whose lanes all hold the same inputs (the wavefront
// that advances the state machine of a grid-per-fit or workgroup-per-data-set kernel).
//
// lm_solve keeps the matrix in registers of every lane and has to write the symmetric pivot interchange as predicated
// swaps for every candidate row -- O(p^3) selects, ~9.5 k cycles of the ~10.7 k of one p = 8 step.  Here lane
// l = 8 i + k owns M[i][k] (both triangles, kept identical), an interchange is one cross-lane gather, the rank-one
// update of step j is ONE multiply-add in every lane, and what lm_solve reads with a static index is read with
// v_readlane (lane numbers are compile-time constants: the loop over j is unrolled).  b, perm and the two
// substitutions stay replicated scalars as in lm_solve (they are O(p^2)).
//
// Same operations on the same values in the same order as lm_solve -> the same bits
// (tests/test_gpu_dense.py::test_wave_solve_bits).
#pragma once

namespace gslnls
{
#if defined(__HIP_DEVICE_COMPILE__)

__device__ __forceinline__ double lsw_readlane(double v, int lane) // lane: wave-uniform
{
    const long long bits = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(bits & 0xffffffffll), lane);
    const int hi = __builtin_amdgcn_readlane((int)(bits >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

__device__ __forceinline__ double lsw_gather(double v, int src_lane) // src_lane: per lane
{
    const long long bits = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(bits & 0xffffffffll));
    const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(bits >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

template <int CTRL>
__device__ __forceinline__ double lsw_dpp(double v)
{
    const long long bits = __double_as_longlong(v);
    int lo = (int)(bits & 0xffffffffll), hi = (int)(bits >> 32);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// maximum over the wavefront of non-negative, possibly NaN values (fmax drops NaNs, so the order does not matter)
__device__ __forceinline__ double lsw_max(double v)
{
    v = fmax(v, lsw_dpp<0xB1>(v));
    v = fmax(v, lsw_dpp<0x4E>(v));
    v = fmax(v, lsw_dpp<0x141>(v));
    v = fmax(v, lsw_dpp<0x140>(v));
    return fmax(fmax(lsw_readlane(v, 0), lsw_readlane(v, 16)), fmax(lsw_readlane(v, 32), lsw_readlane(v, 48)));
}

template <int P>
__device__ void lm_solve_wave(const double *Ap, const double *diag, double mu, const double *rhs, double *sol)
{
    static_assert(P >= 2 && P <= 8, "one matrix element per lane");
    const int lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int li = lane >> 3, lk = lane & 7;
    const int hi = li > lk ? li : lk, lo = li > lk ? lk : li;
    double m = 0.0;
    int perm[P];
    double b[P], dinv[P];
#pragma unroll
    for (int i = 0; i < P; ++i)
    {
        perm[i] = i;
        b[i] = rhs[i];
#pragma unroll
        for (int j = 0; j <= i; ++j)
        {
            double v = Ap[tri(i, j)];
            if (i == j)
                v += mu * diag[i] * diag[i];
            m = (hi == i && lo == j) ? v : m;
        }
    }
    const double am = fabs(m);
    const double gamma = lsw_max(li == lk ? am : 0.0); // lanes outside the p x p block hold 0
    const double xi = lsw_max(li == lk ? 0.0 : am);
    const double beta = fmax(fmax(gamma, xi / sqrt((double)P * P - 1.0)), DBL_EPSILON);
    const double betainv = 1.0 / sqrt(beta);
#pragma unroll
    for (int j = 0; j < P; ++j)
    {
        int q = j;
        double maxd = fabs(lsw_readlane(m, 9 * j));
#pragma unroll
        for (int i = j + 1; i < P; ++i)
        {
            const double d = fabs(lsw_readlane(m, 9 * i));
            if (d > maxd)
            {
                maxd = d;
                q = i;
            }
        }
        q = __builtin_amdgcn_readfirstlane(q);
        if (q != j)
        {
            // symmetric interchange of rows/columns j and q: new M[i][k] = old M[s(i)][s(k)], s = (j q)
            const int si = li == j ? q : (li == q ? j : li);
            const int sk = lk == j ? q : (lk == q ? j : lk);
            m = lsw_gather(m, 8 * si + sk);
        }
#pragma unroll
        for (int i = j + 1; i < P; ++i)
        {
            if (i == q)
            {
                const double tb = b[j];
                b[j] = b[i];
                b[i] = tb;
                const int tp = perm[j];
                perm[j] = perm[i];
                perm[i] = tp;
            }
        }
        // column j below the diagonal, as wave-uniform values
        double c[P];
        double theta = 0.0;
#pragma unroll
        for (int i = j + 1; i < P; ++i)
        {
            c[i] = lsw_readlane(m, 8 * i + j);
            theta = fmax(theta, fabs(c[i]));
        }
        const double u = theta * betainv;
        const double alpha = fmax(fmax(DBL_EPSILON, fabs(lsw_readlane(m, 9 * j))), u * u);
        const double ainv = 1.0 / alpha;
        dinv[j] = ainv;
        if (j + 1 < P)
        {
            // M[i][k] -= ainv * M[i][j] * M[k][j] for j < k <= i, mirrored for k > i with the same operand order
            double chi = c[P - 1], clo = c[P - 1];
#pragma unroll
            for (int i = j + 1; i < P - 1; ++i)
            {
                chi = (hi == i) ? c[i] : chi;
                clo = (lo == i) ? c[i] : clo;
            }
            const double upd = m - ainv * chi * clo;
            const double scl = m * ainv;
            const bool inblock = hi < P;
            m = (inblock && lo > j) ? upd : m;
            m = (inblock && lo == j && hi > j) ? scl : m;
        }
        m = (lane == 9 * j) ? alpha : m;
    }
    // forward substitution, D^-1, back substitution: L[i][j] read from its lane
#pragma unroll
    for (int i = 0; i < P; ++i)
    {
#pragma unroll
        for (int j = 0; j < i; ++j)
            b[i] -= lsw_readlane(m, 8 * i + j) * b[j];
    }
#pragma unroll
    for (int i = 0; i < P; ++i)
        b[i] *= dinv[i];
#pragma unroll
    for (int i = P - 1; i >= 0; --i)
    {
#pragma unroll
        for (int j = i + 1; j < P; ++j)
            b[i] -= lsw_readlane(m, 8 * j + i) * b[j];
    }
#pragma unroll
    for (int i = 0; i < P; ++i)
    {
#pragma unroll
        for (int k = 0; k < P; ++k)
            if (perm[i] == k)
                sol[k] = b[i];
    }
}

#endif
} // namespace gslnls
