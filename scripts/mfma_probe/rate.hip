// issue rate of v_mfma_f64_16x16x4_f64 on gfx950: one wave per SIMD, 10 independent accumulators, back to back
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64_t __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void rate(double *out, int iters, unsigned long long *cyc)
{
    v4f64_t acc[NACC];
    for (int q = 0; q < NACC; ++q)
        acc[q] = (v4f64_t){0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it)
    {
#pragma unroll
        for (int q = 0; q < NACC; ++q)
            acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int q = 0; q < NACC; ++q)
        s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0)
        cyc[blockIdx.x] = t1 - t0;
}
int main()
{
    double *out;
    unsigned long long *cyc, h[1024];
    hipMalloc(&out, 1024 * 256 * 8);
    hipMalloc(&cyc, 1024 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    for (int wg = 256; wg <= 1024; wg *= 2)
    {
        hipLaunchKernelGGL((rate<10>), dim3(wg), dim3(256), 0, 0, out, 100, cyc);
        hipEventRecord(e0);
        hipLaunchKernelGGL((rate<10>), dim3(wg), dim3(256), 0, 0, out, iters, cyc);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, cyc, wg * 8, hipMemcpyDeviceToHost);
        const double nm = (double)wg * 4 * iters * 10;
        printf("%4d WGs x 4 waves: %.3f ms, %.1f TFLOP/s fp64, %.1f shader cycles per MFMA per wave (s_memtime)\n", wg, ms,
               nm * 2048 / ms / 1e9, (double)h[0] / (iters * 10.0));
    }
    return 0;
}
