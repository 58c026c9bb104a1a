// sobol.hpp -- quasi-random start points for multi-start, index-addressable.
//
// The reference draws points one after another from gsl_qrng_sobol (p < 41) or
// gsl_qrng_halton (src/nls.c:277-280, src/nls_mstart.c:48; re-initialised at src/nls.c:447).
// GSL's Sobol generator is the Antonov-Saleev Gray-code recurrence
//     x_{k+1} = x_k XOR v[c_k],   c_k = index of the lowest zero bit of k,
// with the Bratley-Fox (ACM TOMS 659) direction numbers, 30 bits, and it returns
// x_1, x_2, ... (first returned point = 0.5 in every dimension).  Because
// x_k = XOR of v[b] over the set bits b of gray(k) = k ^ (k >> 1), the k-th draw can be
// computed directly from its index -- which is what lets every GPU lane (and every rank
// of a multi-GPU job) generate its own block of the sequence without a serial generator.
#pragma once
#include "lm_core.hpp"

namespace gslnls
{

constexpr int SOBOL_MAX_DIM = 40;
constexpr int SOBOL_BITS = 30;

struct SobolTable
{
    int dim;
    int halton;                         // p > 40: Halton radical inverses instead
    unsigned int v[SOBOL_BITS][SOBOL_MAX_DIM]; // direction numbers scaled to 2^30
};

// direction-number construction (host side, once per fit)
inline void sobol_build(SobolTable &t, int dim)
{
    static const int poly[SOBOL_MAX_DIM] = {1,   3,   7,   11,  13,  19,  25,  37,  59,  47,  61,  55,  41,  67,
                                            97,  91,  109, 103, 115, 131, 193, 137, 145, 143, 241, 157, 185, 167,
                                            229, 171, 213, 191, 253, 203, 211, 239, 247, 285, 369, 299};
    static const int deg[SOBOL_MAX_DIM] = {0, 1, 2, 3, 3, 4, 4, 5, 5, 5, 5, 5, 5, 6, 6, 6, 6, 6, 6, 7,
                                           7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 8, 8, 8};
    static const int vinit[8][SOBOL_MAX_DIM] = {
        {0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
         1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1},
        {0, 0, 1, 3, 1, 3, 1, 3, 3, 1, 3, 1, 3, 1, 3, 1, 1, 3, 1, 3,
         1, 3, 1, 3, 3, 1, 3, 1, 3, 1, 3, 1, 1, 3, 1, 3, 1, 3, 1, 3},
        {0, 0, 0, 7, 5, 1, 3, 3, 7, 5, 5, 7, 7, 1, 3, 3, 7, 5, 1, 1,
         5, 3, 3, 1, 7, 5, 1, 3, 3, 7, 5, 1, 1, 5, 7, 7, 5, 1, 3, 3},
        {0,  0, 0,  0, 0, 1,  7, 9, 13, 11, 1, 3,  7, 9,  5,  13, 13, 11, 3, 15,
         5,  3, 15, 7, 9, 13, 9, 1, 11, 7,  5, 15, 1, 15, 11, 5,  3,  1,  7, 9},
        {0, 0,  0,  0, 0,  0,  0, 9,  3,  27, 15, 29, 21, 23, 19, 11, 25, 7,  13, 17,
         1, 25, 29, 3, 31, 11, 5, 23, 27, 19, 21, 5,  1,  17, 13, 7,  15, 9,  31, 9},
        {0,  0,  0,  0,  0,  0, 0,  0,  0,  0,  0, 0,  0,  37, 33, 7,  5,  11, 39, 63,
         27, 17, 15, 23, 29, 3, 21, 13, 31, 25, 9, 49, 33, 19, 29, 11, 19, 27, 15, 25},
        {0,  0,   0,  0,  0,  0,  0,   0,  0,  0,   0, 0,  0,  0,  0, 0,   0,  0,  0,  13,
         33, 115, 41, 79, 17, 29, 119, 75, 73, 105, 7, 59, 65, 21, 3, 113, 61, 89, 45, 107},
        {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
         0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 7, 23, 39}};
    t.dim = dim;
    t.halton = dim > SOBOL_MAX_DIM;
    if (t.halton)
        return;
    for (int k = 0; k < SOBOL_BITS; ++k)
        t.v[k][0] = 1;
    for (int d = 1; d < dim; ++d)
    {
        const int m = deg[d];
        int incl[8], pp = poly[d];
        for (int k = m - 1; k >= 0; --k)
        {
            incl[k] = (pp % 2) == 1;
            pp /= 2;
        }
        for (int j = 0; j < m; ++j)
            t.v[j][d] = (unsigned int)vinit[j][d];
        for (int j = m; j < SOBOL_BITS; ++j)
        {
            unsigned int nv = t.v[j - m][d];
            unsigned int ell = 1;
            for (int k = 0; k < m; ++k)
            {
                ell *= 2;
                if (incl[k])
                    nv ^= ell * t.v[j - k - 1][d];
            }
            t.v[j][d] = nv;
        }
    }
    unsigned int ell = 1;
    for (int j = SOBOL_BITS - 2; j >= 0; --j)
    {
        ell *= 2;
        for (int d = 0; d < dim; ++d)
            t.v[j][d] *= ell;
    }
}

// coordinate `d` of the draw with 0-based index `draw` (the (draw+1)-th Gray-code point)
GSLNLS_HD double sobol_coord(const SobolTable &t, unsigned int draw, int d)
{
    if (t.halton)
    {
        // GSL halton: radical inverse of (draw + 1) in the d-th prime base
        unsigned int prime = 2, found = 0;
        for (unsigned int cand = 2;; ++cand)
        {
            bool is = true;
            for (unsigned int q = 2; q * q <= cand; ++q)
                if (cand % q == 0)
                {
                    is = false;
                    break;
                }
            if (is)
            {
                if ((int)found == d)
                {
                    prime = cand;
                    break;
                }
                ++found;
            }
        }
        unsigned int k = draw + 1;
        const double binv = 1.0 / prime;
        double r = 0.0, f = binv;
        while (k > 0)
        {
            r += f * (double)(k % prime);
            k /= prime;
            f *= binv;
        }
        return r;
    }
    const unsigned int k = draw + 1;
    unsigned int g = k ^ (k >> 1), num = 0;
#pragma unroll 1
    for (int b = 0; b < SOBOL_BITS && g; ++b, g >>= 1)
        if (g & 1u)
            num ^= t.v[b][d];
    return (double)num * (1.0 / 1073741824.0); // 2^-30
}

// map a uniform draw to the sampling range with exponent kd (src/nls_mstart.c:49-70)
GSLNLS_HD double sobol_to_range(double u, double l0, double l1, double kd)
{
    if (!(l1 > l0))
        return l0;
    const double q = l0 + (l1 - l0) * u;
    if (l0 > 0.0)
        return (pow(q - l0 + 1.0, kd) - 1.0) / kd + l0;
    if (l1 < 0.0)
        return -(pow(-q + l1 + 1.0, kd) - 1.0) / kd + l1;
    if (q > 0.0)
        return (pow(q + 1.0, kd) - 1.0) / kd;
    return -(pow(-q + 1.0, kd) - 1.0) / kd;
}

} // namespace gslnls
