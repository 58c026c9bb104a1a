// irls_kernels.hpp -- device side of one IRLS re-weighting (src/nls_irls.c:486-514):
//   unweighted residuals at the fitted point            (:487-492)
//   sigma = 1.482602218505602 * median |r|               (:493; gsl_median src/nls_utils.c:162-189
//           full-sorts with R_orderVector1 -- here an 8-pass radix SELECT on the IEEE bit pattern,
//           histogram in LDS, bin choice by a one-workgroup kernel, no host round trip)
//   w_i = max(psi(r_i/sigma)/(r_i/sigma), eps), psi, psi' (:495-505), normalised to sum n (:507) and
//           multiplied by the user weights (:510-514)
// Counting is integer arithmetic and every floating-point sum has a fixed shape: deterministic.
#pragma once
#include <hip/hip_runtime.h>
#include "dense_kernels.hpp"
#include "irls_core.hpp"

namespace gslnls
{

// r_i = model(theta, x_i) - y_i (unweighted; +Inf for a non-finite model value), key_i = bits(|r_i|)
template <class M, int T>
__global__ __launch_bounds__(T) void irls_resid_kernel(DenseCtx<M::P> ctx, int parity, double *r,
                                                       unsigned long long *keys)
{
    constexpr int P = M::P, NX = M::NX;
    const LmState<P> *s = ctx.state[parity];
    double th[P];
#pragma unroll
    for (int k = 0; k < P; ++k)
        th[k] = s->x[k];
    const long long n = ctx.n, stride = (long long)gridDim.x * T;
    for (long long i = (long long)blockIdx.x * T + threadIdx.x; i < n; i += stride)
    {
        double xr[NX];
#pragma unroll
        for (int c = 0; c < NX; ++c)
            xr[c] = ctx.x[c][i];
        const double v = row_resid<M>(th, xr, ctx.y[i], 1.0);
        r[i] = v;
        keys[i] = (unsigned long long)__double_as_longlong(fabs(v));
    }
}

struct SelectState
{
    unsigned long long prefix, mask;
    unsigned long long k; // rank still to find inside the current prefix class (0-based)
    unsigned int hist[256];
    double value; // result
};

// histogram of byte `pass` over the keys whose already-fixed high bytes equal the prefix
static __global__ __launch_bounds__(256) void select_hist_kernel(const unsigned long long *keys, long long n, int pass,
                                                          SelectState *st)
{
    __shared__ unsigned int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const unsigned long long prefix = st->prefix, mask = st->mask;
    const int shift = 8 * pass;
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
    {
        const unsigned long long key = keys[i];
        if ((key & mask) == prefix)
            atomicAdd(&h[(key >> shift) & 255ull], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x])
        atomicAdd(&st->hist[threadIdx.x], h[threadIdx.x]);
}

// choose the bin that holds rank k, descend into it, clear the histogram for the next pass
static __global__ void select_pick_kernel(int pass, SelectState *st)
{
    if (threadIdx.x != 0 || blockIdx.x != 0)
        return;
    unsigned long long k = st->k, cum = 0;
    int bin = 255;
    for (int b = 0; b < 256; ++b)
    {
        const unsigned long long c = st->hist[b];
        if (k < cum + c)
        {
            bin = b;
            break;
        }
        cum += c;
    }
    st->k = k - cum;
    st->prefix |= (unsigned long long)bin << (8 * pass);
    st->mask |= 255ull << (8 * pass);
    for (int b = 0; b < 256; ++b)
        st->hist[b] = 0;
    if (pass == 0)
        st->value = __longlong_as_double((long long)st->prefix);
}

static __global__ void select_init_kernel(SelectState *st, unsigned long long k)
{
    if (threadIdx.x != 0 || blockIdx.x != 0)
        return;
    st->prefix = 0;
    st->mask = 0;
    st->k = k;
    for (int b = 0; b < 256; ++b)
        st->hist[b] = 0;
    st->value = 0.0;
}

struct IrlsScalars
{
    double sigma;    // 1.4826 * median |r|
    double sum_wts;  // sum of raw weights
    double scale;    // n / sum_wts
};

// sigma from the one or two middle order statistics (n odd / even), src/nls_utils.c:177-186
static __global__ void irls_sigma_kernel(const SelectState *lo, const SelectState *hi, IrlsScalars *sc)
{
    if (threadIdx.x == 0 && blockIdx.x == 0)
        sc->sigma = 1.482602218505602 * ((lo == hi) ? lo->value : (lo->value + hi->value) / 2.0);
}

// raw IRLS weights, psi, psi' + per-workgroup partial sums of the weights
template <int T>
__global__ __launch_bounds__(T) void irls_weight_kernel(const double *r, long long n, LossCfg L, const IrlsScalars *sc,
                                                        double *wt, double *psi, double *psip, double *partial)
{
    __shared__ double lds[T / 64];
    const double sigma = sc->sigma;
    double acc = 0.0;
    const long long stride = (long long)gridDim.x * T;
    for (long long i = (long long)blockIdx.x * T + threadIdx.x; i < n; i += stride)
    {
        const double rs = r[i] / sigma;
        const double ps = irls_psi(rs, L);
        const double w = fmax(ps / rs, DBL_EPSILON);
        wt[i] = w;
        psi[i] = ps;
        psip[i] = irls_psip(rs, L);
        acc += w;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0)
        lds[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0)
    {
        double t = 0.0;
        for (int w = 0; w < T / 64; ++w)
            t += lds[w];
        partial[blockIdx.x] = t;
    }
}

static __global__ void irls_scale_kernel(const double *partial, int nblk, long long n, IrlsScalars *sc)
{
    if (threadIdx.x != 0 || blockIdx.x != 0)
        return;
    double t = 0.0;
    for (int b = 0; b < nblk; ++b)
        t += partial[b];
    sc->sum_wts = t;
    sc->scale = (double)n / t;
}

// workn_i = w_i * n / sum(w) [* user weight]; sqrt goes to the solver (gsl_multifit_nlinear_winit)
template <int T>
__global__ __launch_bounds__(T) void irls_apply_kernel(double *wt, long long n, const IrlsScalars *sc,
                                                       const double *user_sw, double *sw_out)
{
    const double scale = sc->scale;
    const long long stride = (long long)gridDim.x * T;
    for (long long i = (long long)blockIdx.x * T + threadIdx.x; i < n; i += stride)
    {
        double w = wt[i] * scale;
        if (user_sw)
            w = (user_sw[i] * user_sw[i]) * w;
        wt[i] = w;
        sw_out[i] = sqrt(w);
    }
}

} // namespace gslnls
