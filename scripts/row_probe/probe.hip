// How fast can the C2 row pass go?  One pass = every thread evaluates its R resident rows of ModelExpDecay
// (f, analytic Jacobian row, 11 accumulators), exactly the arithmetic of lm_step_kernel / lm_fit_kernel, repeated
// `iters` times inside one launch so that only fp64 issue is measured.  Variants: threads per workgroup x rows per
// thread (same 1,048,576 rows per pass), with and without the non-finite / dead-row guards.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../gslnls_amd/csrc/models.hpp"
#include "../../gslnls_amd/csrc/rowops.hpp"
using namespace gslnls;

template <int T, int R, int MODE>
__global__ __launch_bounds__(T) void pass_kernel(const double *x, const double *y, long long n, int iters, double *out)
{
    using M = ModelExpDecay;
    const long long stride = (long long)gridDim.x * T;
    const long long i0 = (long long)blockIdx.x * T + threadIdx.x;
    double px[R], py[R], pw[R];
#pragma unroll
    for (int k = 0; k < R; ++k)
    {
        const long long i = i0 + k * stride;
        const long long ic = i < n ? i : n - 1;
        px[k] = x[ic];
        py[k] = y[ic];
        pw[k] = i < n ? 1.0 : 0.0;
    }
    double th[3] = {4.0, 1.2, 0.8}, delta[3] = {0, 0, 0};
    PassSums<3> tot;
    pass_zero<3>(tot);
    for (int it = 0; it < iters; ++it)
    {
        PassSums<3> acc;
        pass_zero<3>(acc);
#pragma unroll
        for (int k = 0; k < R; ++k)
        {
            double Jrow[3];
            if (MODE == 0)
            {
                double f = row_fj<M, JAC_ANALYTIC>(th, delta, &px[k], py[k], pw[k], Jrow, &acc.badj);
                f = (pw[k] != 0.0) ? f : 0.0;
                acc_fj<3>(acc, f, Jrow);
            }
            else
            {
                // no guards: plain exp polynomial, no isfinite / NaN restore / dead-row select
                const double e = gexp(-th[1] * px[k]);
                const double f = (th[0] * e + th[2] - py[k]) * pw[k];
                Jrow[0] = e * pw[k];
                Jrow[1] = -th[0] * px[k] * e * pw[k];
                Jrow[2] = pw[k];
                acc_fj<3>(acc, f, Jrow);
            }
        }
        // keep every pass alive and dependent on the previous one through the point
        double *a = reinterpret_cast<double *>(&acc), *t = reinterpret_cast<double *>(&tot);
#pragma unroll
        for (int v = 0; v < PassSums<3>::NV; ++v)
            t[v] += a[v];
        th[1] += 1e-12 * (acc.ssr > 1e300 ? 1.0 : 0.0) + 1e-9;
    }
    double s = 0.0;
    const double *t = reinterpret_cast<const double *>(&tot);
#pragma unroll
    for (int v = 0; v < PassSums<3>::NV; ++v)
        s += t[v];
    out[(size_t)blockIdx.x * T + threadIdx.x] = s;
}

template <int T, int R, int MODE>
void run(const double *dx, const double *dy, long long n, double *out, const char *label)
{
    const int G = 256 * 512 / T * (8 / R) > 0 ? (int)((n + (long long)T * R - 1) / ((long long)T * R)) : 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 2000;
    hipLaunchKernelGGL((pass_kernel<T, R, MODE>), dim3(G), dim3(T), 0, 0, dx, dy, n, 50, out);
    hipEventRecord(e0);
    hipLaunchKernelGGL((pass_kernel<T, R, MODE>), dim3(G), dim3(T), 0, 0, dx, dy, n, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s G=%4d T=%4d R=%2d : %.3f us per pass over %lld rows\n", label, G, T, R, ms * 1e3 / iters, n);
}

int main()
{
    const long long n = 1000000;
    std::vector<double> x(n), y(n);
    for (long long i = 0; i < n; ++i)
    {
        x[i] = 3.0 * i / (n - 1);
        y[i] = 5.0 * exp(-1.5 * x[i]) + 1.0 + 0.01 * ((i * 2654435761u) % 1000) / 1000.0;
    }
    double *dx, *dy, *out;
    hipMalloc(&dx, n * 8);
    hipMalloc(&dy, n * 8);
    hipMalloc(&out, 8 * 1024 * 1024);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipMemcpy(dy, y.data(), n * 8, hipMemcpyHostToDevice);
    run<512, 8, 0>(dx, dy, n, out, "guards, 2 waves/SIMD x 8 rows");
    run<512, 8, 1>(dx, dy, n, out, "no guards, 2 waves/SIMD x 8 rows");
    run<1024, 4, 0>(dx, dy, n, out, "guards, 4 waves/SIMD x 4 rows");
    run<1024, 4, 1>(dx, dy, n, out, "no guards, 4 waves/SIMD x 4 rows");
    run<256, 16, 0>(dx, dy, n, out, "guards, 1 wave/SIMD x 16 rows");
    run<256, 16, 1>(dx, dy, n, out, "no guards, 1 wave/SIMD x 16 rows");
    run<512, 4, 0>(dx, dy, n, out, "guards, 2 waves/SIMD x 4 rows, 512 WGs");
    return 0;
}
