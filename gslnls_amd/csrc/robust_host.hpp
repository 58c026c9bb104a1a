// robust_host.hpp -- the robust second pass of multi-start (src/nls.c:401-509) on device.
//
// When a robust loss is requested together with multi-start, the reference flags outliers at the
// first-pass optimum by Cook's distance, D_i = e_i^2/(p s^2) * h_i/(1-h_i)^2 with hat values
// h_i = J_i (J^T J)^-1 J_i^T (src/nls_utils.c:88-150), zero-weights every observation with
// D_i > min(4/n, 5 MAD(D)) (src/nls.c:424-443; gsl_mad src/nls_utils.c:201-217), and repeats the whole
// multi-start with those weights.  Here: one fused pass for J^T J and ssr, a per-row kernel for D_i,
// two radix selects for the MAD, one kernel for the 0/1 weights -- nothing n-sized leaves HBM.
#pragma once
#include "dense_host.hpp"
#include "irls_kernels.hpp"
#include "large_host.hpp"

namespace gslnls
{

template <int P>
struct CooksArgs
{
    double theta[P];
    double Cinv[P * P];
    double s2;
};

template <class M, int JAC, int T>
__global__ __launch_bounds__(T) void cooks_kernel(DenseCtx<M::P> ctx, CooksArgs<M::P> a, double *d,
                                                  unsigned long long *keys, double *hat = nullptr)
{
    constexpr int P = M::P, NX = M::NX;
    double th[P], delta[P];
#pragma unroll
    for (int k = 0; k < P; ++k)
        th[k] = a.theta[k];
    fd_deltas<P>(th, ctx.prm.h_df, delta);
    const long long n = ctx.n, stride = (long long)gridDim.x * T;
    for (long long i = (long long)blockIdx.x * T + threadIdx.x; i < n; i += stride)
    {
        double xr[NX];
#pragma unroll
        for (int c = 0; c < NX; ++c)
            xr[c] = ctx.x[c][i];
        double Jrow[P], nb = 0.0;
        const double e = row_fj<M, JAC>(th, delta, xr, ctx.y[i], ctx.sw ? ctx.sw[i] : 1.0, Jrow, &nb);
        double h = 0.0;
#pragma unroll
        for (int j = 0; j < P; ++j)
        {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < P; ++k)
                s += Jrow[k] * a.Cinv[k * P + j];
            h += s * Jrow[j];
        }
        const double di = (e * e) / (P * a.s2) * (h / ((1 - h) * (1 - h)));
        d[i] = di;
        if (keys)
            keys[i] = (unsigned long long)__double_as_longlong(fabs(di));
        if (hat)
            hat[i] = h;
    }
}

static __global__ void absdev_keys_kernel(const double *d, long long n, double med, unsigned long long *keys)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        keys[i] = (unsigned long long)__double_as_longlong(fabs(d[i] - med));
}

static __global__ void outlier_weights_kernel(const double *d, long long n, double thresh, const double *user_sw,
                                       double *sw_out, int *noutlier)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    int cnt = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    {
        const bool out = d[i] > thresh;
        sw_out[i] = out ? 0.0 : (user_sw ? user_sw[i] : 1.0);
        cnt += out ? 1 : 0;
    }
    if (cnt)
        atomicAdd(noutlier, cnt);
}

// median of n non-negative doubles given as IEEE bit patterns (gsl_median, src/nls_utils.c:162-189)
inline int device_median(hipStream_t st, const unsigned long long *d_keys, long long n, SelectState *d_sel,
                         double *median)
{
    const unsigned long long k_lo = (unsigned long long)((n - 1) / 2), k_hi = (unsigned long long)(n / 2);
    const int nsel = (k_lo == k_hi) ? 1 : 2;
    for (int which = 0; which < nsel; ++which)
    {
        SelectState *s = d_sel + which;
        hipLaunchKernelGGL(select_init_kernel, dim3(1), dim3(1), 0, st, s, which == 0 ? k_lo : k_hi);
        for (int pass = 7; pass >= 0; --pass)
        {
            hipLaunchKernelGGL(select_hist_kernel, dim3(std::min<long long>(1024, (n + 255) / 256)), dim3(256), 0, st,
                               d_keys, n, pass, s);
            hipLaunchKernelGGL(select_pick_kernel, dim3(1), dim3(1), 0, st, pass, s);
        }
    }
    SelectState h[2];
    GSLNLS_HIP_OK(hipMemcpyAsync(h, d_sel, sizeof(SelectState) * nsel, hipMemcpyDeviceToHost, st));
    GSLNLS_HIP_OK(hipStreamSynchronize(st));
    *median = (nsel == 1) ? h[0].value : (h[0].value + h[1].value) / 2.0;
    return 0;
}

// ssr, J^T J, J^T f at theta through the regular pass kernel (fresh state => no advance)
template <class M>
int DenseFit<M>::sums_at(const double *theta, int jacmode, PassSums<P> &out)
{
    StartArgs<P> sa;
    for (int k = 0; k < P; ++k)
    {
        sa.start[k] = theta[k];
        sa.lo[k] = -INFINITY;
        sa.up[k] = INFINITY;
    }
    const unsigned int seq_keep = ctx.seq;
    ctx.sa = sa;
    launch_step(jacmode, 0, true);
    double *d_tot = nullptr;
    GSLNLS_HIP_OK(hipMalloc(&d_tot, sizeof(double) * NV));
    hipLaunchKernelGGL(large_reduce_kernel, dim3(NV), dim3(64), 0, stream, ctx.partials[0], NV, ctx.G, d_tot);
    GSLNLS_HIP_OK(hipMemcpyAsync(&out, d_tot, sizeof(double) * NV, hipMemcpyDeviceToHost, stream));
    GSLNLS_HIP_OK(hipStreamSynchronize(stream));
    hipFree(d_tot);
    ctx.seq = seq_keep;
    return 0;
}

// returns 1 when a second pass has to run (sw_robust filled), 0 when not, < 0 on error
template <class M>
int DenseFit<M>::robust_weights(int jacmode, const double *mpopt, double *d_sw_robust)
{
    PassSums<P> s;
    int rc = sums_at(mpopt, jacmode, s);
    if (rc)
        return rc;
    std::vector<double> A((size_t)P * P);
    for (int i = 0; i < P; ++i)
        for (int j = 0; j <= i; ++j)
            A[i * P + j] = A[j * P + i] = s.A[tri(i, j)];
    if (!lg_chol(P, A)) // cooks_d -> hat_values fails: no second pass (src/nls.c:419-421)
        return 0;
    lg_chol_invert(P, A);
    CooksArgs<P> ca;
    for (int k = 0; k < P; ++k)
        ca.theta[k] = mpopt[k];
    for (int k = 0; k < P * P; ++k)
        ca.Cinv[k] = A[k];
    ca.s2 = s.ssr / (n - P);
    double *d_d = nullptr;
    unsigned long long *d_keys = nullptr;
    SelectState *d_sel = nullptr;
    int *d_cnt = nullptr;
    GSLNLS_HIP_OK(hipMalloc(&d_d, sizeof(double) * (size_t)n));
    GSLNLS_HIP_OK(hipMalloc(&d_keys, sizeof(unsigned long long) * (size_t)n));
    GSLNLS_HIP_OK(hipMalloc(&d_sel, sizeof(SelectState) * 2));
    GSLNLS_HIP_OK(hipMalloc(&d_cnt, sizeof(int)));
    GSLNLS_HIP_OK(hipMemsetAsync(d_cnt, 0, sizeof(int), stream));
    const int Gf = std::min(2048, (int)((n + T - 1) / T));
    switch (jacmode)
    {
    case JAC_ANALYTIC:
        hipLaunchKernelGGL((cooks_kernel<M, JAC_ANALYTIC, T>), dim3(Gf), dim3(T), 0, stream, ctx, ca, d_d, d_keys);
        break;
    case JAC_FORWARD:
        hipLaunchKernelGGL((cooks_kernel<M, JAC_FORWARD, T>), dim3(Gf), dim3(T), 0, stream, ctx, ca, d_d, d_keys);
        break;
    default:
        hipLaunchKernelGGL((cooks_kernel<M, JAC_CENTER, T>), dim3(Gf), dim3(T), 0, stream, ctx, ca, d_d, d_keys);
        break;
    }
    double med = 0.0, med2 = 0.0;
    rc = device_median(stream, d_keys, n, d_sel, &med);
    if (!rc)
    {
        hipLaunchKernelGGL(absdev_keys_kernel, dim3(Gf), dim3(256), 0, stream, d_d, (long long)n, med, d_keys);
        rc = device_median(stream, d_keys, n, d_sel, &med2);
    }
    int noutlier = 0;
    if (!rc)
    {
        const double mad = 1.482602218505602 * med2;
        const double thresh = fmin(4.0 / n, 5 * mad);
        hipLaunchKernelGGL(outlier_weights_kernel, dim3(Gf), dim3(256), 0, stream, d_d, (long long)n, thresh, ctx.sw,
                           d_sw_robust, d_cnt);
        hipMemcpyAsync(&noutlier, d_cnt, sizeof(int), hipMemcpyDeviceToHost, stream);
        hipStreamSynchronize(stream);
    }
    hipFree(d_d);
    hipFree(d_keys);
    hipFree(d_sel);
    hipFree(d_cnt);
    if (rc)
        return rc;
    return (noutlier > 0 && noutlier < (n - P)) ? 1 : 0;
}

// hat values h_i = J_i (J^T J)^-1 J_i^T and Cook's distances at `theta` (src/nls_utils.c:88-150; what the S3
// methods hatvalues() / cooks.distance() of the reference compute from the n x p gradient on the host)
template <class M>
int DenseFit<M>::diagnostics(int jac, const double *theta, const int *ci, const double *cd, double *hat, double *cooks)
{
    ctx.prm = make_params(ci, cd, jac, 0, false, ctx.sw != nullptr);
    const int jacmode = jac ? JAC_ANALYTIC : (ci[5] ? JAC_CENTER : JAC_FORWARD);
    PassSums<P> s;
    int rc = sums_at(theta, jacmode, s);
    if (rc)
        return rc;
    std::vector<double> A((size_t)P * P);
    for (int i = 0; i < P; ++i)
        for (int j = 0; j <= i; ++j)
            A[i * P + j] = A[j * P + i] = s.A[tri(i, j)];
    if (!lg_chol(P, A))
        return GSLNLS_EINVAL; // singular J^T J: hat_values fails in the reference as well
    lg_chol_invert(P, A);
    CooksArgs<P> ca;
    for (int k = 0; k < P; ++k)
        ca.theta[k] = theta[k];
    for (int k = 0; k < P * P; ++k)
        ca.Cinv[k] = A[k];
    ca.s2 = s.ssr / (n - P);
    double *d_d = nullptr, *d_h = nullptr;
    GSLNLS_HIP_OK(hipMalloc(&d_d, sizeof(double) * (size_t)n));
    GSLNLS_HIP_OK(hipMalloc(&d_h, sizeof(double) * (size_t)n));
    const int Gf = std::min(2048, (int)((n + T - 1) / T));
    switch (jacmode)
    {
    case JAC_ANALYTIC:
        hipLaunchKernelGGL((cooks_kernel<M, JAC_ANALYTIC, T>), dim3(Gf), dim3(T), 0, stream, ctx, ca, d_d,
                           (unsigned long long *)nullptr, d_h);
        break;
    case JAC_FORWARD:
        hipLaunchKernelGGL((cooks_kernel<M, JAC_FORWARD, T>), dim3(Gf), dim3(T), 0, stream, ctx, ca, d_d,
                           (unsigned long long *)nullptr, d_h);
        break;
    default:
        hipLaunchKernelGGL((cooks_kernel<M, JAC_CENTER, T>), dim3(Gf), dim3(T), 0, stream, ctx, ca, d_d,
                           (unsigned long long *)nullptr, d_h);
        break;
    }
    hipError_t e = hipSuccess;
    if (hat)
        e = hipMemcpyAsync(hat, d_h, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, stream);
    if (cooks && e == hipSuccess)
        e = hipMemcpyAsync(cooks, d_d, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess)
        e = hipStreamSynchronize(stream);
    (void)hipFree(d_d);
    (void)hipFree(d_h);
    return e == hipSuccess ? GSLNLS_SUCCESS : GSLNLS_E_NODEVICE;
}

} // namespace gslnls
