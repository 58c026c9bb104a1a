// bd_model_kernels.hpp -- a formula with more than 64 parameters as a MATRIX producer: the rows of the model value, of its
// Jacobian (n x p column-major in HBM, the layout C_nls returns as `grad`, src/nls.c:718) and of the second directional
// derivative, from the row model the in-process compiler generates for the formula (rtc_host.hpp: struct ModelJit --
// value, symbolic gradient = stats::deriv, R/nls.R:588-599, D^2 f[v, v] = deriv(hessian = TRUE), R/nls.R:600-640).
// The fit around them is csrc/bd_host.hpp.  Compiled in process only (M = ModelJit).
#pragma once
#if !defined(__HIPCC_RTC__)
#include <hip/hip_runtime.h>
#endif
#include "lm_core.hpp"
#include "devmath.hpp"

namespace gslnls
{

struct BdTheta
{
    const double *p;
    __device__ __forceinline__ double operator[](int k) const { return p[k]; }
};
struct BdPreOut
{
    double *p;
    __device__ __forceinline__ void set(int k, double v) { p[k] = v; }
};
// gradient entry k of row i -> J[i + n k]: for every k the lanes of a wavefront write consecutive doubles
struct BdJacSink
{
    double *dst;
    long long n;
    __device__ __forceinline__ void set(int k, double v) { dst[(size_t)n * k] = v; }
};

// the same with the row's sqrt(w) applied and the non-finite flag of bd_weight_kernel taken on the way (mode 3, round 5: one
// kernel instead of two per Jacobian; v * sw and fma(v, 0, bad) are that kernel's operations on the same operands)
struct BdJacSinkW
{
    double *dst;
    long long n;
    double sw, bad;
    __device__ __forceinline__ void set(int k, double v)
    {
        bad = fma(v, 0.0, bad);
        dst[(size_t)n * k] = v * sw;
    }
};

// bd_block_sum of bd_kernels.hpp (not visible to the in-process compiler): wavefront xor butterflies, then the four wave sums
// in wave order -- the same additions in the same order, so that mode 4 below leaves the partial sums bd_resid_kernel leaves
__device__ __forceinline__ double bd_model_block_sum(double v, double *red_s)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
    {
        const long long bits = __double_as_longlong(v);
        const int lo = __builtin_amdgcn_ds_bpermute((lane ^ m) << 2, (int)(bits & 0xffffffffll));
        const int hi = __builtin_amdgcn_ds_bpermute((lane ^ m) << 2, (int)(bits >> 32));
        v += __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
    }
    __syncthreads();
    if (lane == 0)
        red_s[wave] = v;
    __syncthreads();
    double s = red_s[0];
    for (int w = 1; w < 4; ++w)
        s += red_s[w];
    return s;
}

// mode 0: fval[i] = m_i(theta); 1: + J (unweighted); 2: out[i] = D^2 m_i[v, v]; 3: as 1 with J's rows scaled by sw (when
// given) and part[block] = NaN when an entry of this block's rows is not finite, else 0; 4: the weighted residual
// fval[i] = sqrt(w_i) (m_i - y_i) (+Inf where m_i is not finite, src/nls.c:843-849; y comes in the J slot) and
// part[block] = the block's sum of squares -- bd_resid_kernel's work on the model value, in the kernel that produces it
template <class M, int MODE>
__global__ __launch_bounds__(256) void bd_model_kernel(const double *theta, const double *dir, const double *x, long long n,
                                                       double *fval, double *J, const double *sw, double *part)
{
    constexpr int P = M::P, NX = M::NX;
    constexpr bool GRAD = MODE == 1 || MODE == 3;
    __shared__ double th_s[P], dir_s[MODE == 2 ? P : 1];
    __shared__ double pre_s[(GRAD && M::NPRE > 0) ? M::NPRE : 1];
    __shared__ int bad_s;
    __shared__ double red_s[4];
    if (MODE == 3 && threadIdx.x == 0)
        bad_s = 0;
    for (int k = threadIdx.x; k < P; k += 256)
    {
        th_s[k] = theta[k];
        if constexpr (MODE == 2)
            dir_s[k] = dir[k];
    }
    __syncthreads();
    const BdTheta th{th_s};
    if constexpr (GRAD && M::NPRE > 0)
    {
        // what depends on the parameters alone, once per workgroup (rtc_host.hpp: prologue / value_grad_sink_pre, round 5);
        // every thread stores the same values
        BdPreOut po{pre_s};
        M::prologue(th, po);
        __syncthreads();
    }
    const BdTheta pre{pre_s};
    double bad = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    {
        double xr[NX];
#pragma unroll
        for (int c = 0; c < NX; ++c)
            xr[c] = x[(size_t)c * n + i];
        if constexpr (MODE == 0)
            fval[i] = M::value(th, xr);
        else if constexpr (MODE == 4)
        {
            const double m = M::value(th, xr);
            const double *yv = J;
            double r = isfinite(m) ? m - yv[i] : __builtin_inf();
            if (sw)
                r *= sw[i];
            fval[i] = r;
            bad += r * r;
        }
        else if constexpr (MODE == 1)
        {
            BdJacSink sink{J + i, n};
            fval[i] = M::value_grad_sink_pre(th, pre, xr, sink);
        }
        else if constexpr (MODE == 3)
        {
            BdJacSinkW sink{J + i, n, sw ? sw[i] : 1.0, bad};
            fval[i] = M::value_grad_sink_pre(th, pre, xr, sink);
            bad = sink.bad;
        }
        else
            fval[i] = M::fvv(th, BdTheta{dir_s}, xr);
    }
    if constexpr (MODE == 4)
    {
        const double tot = bd_model_block_sum(bad, red_s);
        if (threadIdx.x == 0 && part)
            part[blockIdx.x] = tot;
    }
    if constexpr (MODE == 3)
    {
        // (bad is 0 or NaN: any thread that saw a non-finite entry raises the block's flag; all store the same word)
        __syncthreads();
        if (bad != 0.0)
            bad_s = 1;
        __syncthreads();
        if (threadIdx.x == 0)
            part[blockIdx.x] = bad_s ? __builtin_nan("") : 0.0;
    }
}

} // namespace gslnls
