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

// mode 0: fval[i] = m_i(theta); 1: + J (unweighted); 2: out[i] = D^2 m_i[v, v]
template <class M, int MODE>
__global__ __launch_bounds__(256) void bd_model_kernel(const double *theta, const double *dir, const double *x, long long n,
                                                       double *fval, double *J)
{
    constexpr int P = M::P, NX = M::NX;
    __shared__ double th_s[P], dir_s[MODE == 2 ? P : 1];
    __shared__ double pre_s[(MODE == 1 && M::NPRE > 0) ? M::NPRE : 1];
    for (int k = threadIdx.x; k < P; k += 256)
    {
        th_s[k] = theta[k];
        if constexpr (MODE == 2)
            dir_s[k] = dir[k];
    }
    __syncthreads();
    const BdTheta th{th_s};
    if constexpr (MODE == 1 && M::NPRE > 0)
    {
        // what depends on the parameters alone, once per workgroup (rtc_host.hpp: prologue / value_grad_sink_pre, round 5);
        // every thread stores the same values
        BdPreOut po{pre_s};
        M::prologue(th, po);
        __syncthreads();
    }
    const BdTheta pre{pre_s};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    {
        double xr[NX];
#pragma unroll
        for (int c = 0; c < NX; ++c)
            xr[c] = x[(size_t)c * n + i];
        if constexpr (MODE == 0)
            fval[i] = M::value(th, xr);
        else if constexpr (MODE == 1)
        {
            BdJacSink sink{J + i, n};
            fval[i] = M::value_grad_sink_pre(th, pre, xr, sink);
        }
        else
            fval[i] = M::fvv(th, BdTheta{dir_s}, xr);
    }
}

} // namespace gslnls
