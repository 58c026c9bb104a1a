// issue rate of v_mfma_f64_16x16x4_f64 on gfx950: one wave per SIMD, 10 independent accumulators, back to back.
// ROUND 5 CORRECTION: the compiler keeps the 10 accumulators of this loop in AGPRs and moves all 80 registers to VGPRs and
// back in EVERY iteration (80 v_accvgpr_write + 10 MFMA + 80 v_accvgpr_read per trip: see the ISA), so the 47.7 TFLOP/s
// this probe reported in round 3 is the rate of that loop, not of the instruction.  overlap.hip (case A: four accumulators
// that stay in VGPRs, no moves -- checked in its ISA) shows the instruction's own rate: 29 ns per MFMA per SIMD = 64
// cycles at the 2.2 GHz the device holds under this load = 72 TFLOP/s.  This file is kept as the record of the old number
// and for the vector pipe's rate below.
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
// the vector pipe's fp64 FMA rate on the same device, same launch shape: 16 independent chains per lane.  Its data-sheet
// peak is the same 78.6 TFLOP/s (256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz), so it calibrates the clock
__global__ __launch_bounds__(256) void valu_rate(double *out, int iters)
{
    double a[16];
    for (int q = 0; q < 16; ++q)
        a[q] = threadIdx.x * 1e-3 + q;
    const double b = 1.0 + threadIdx.x * 1e-9, c = 1e-7;
    for (int it = 0; it < iters; ++it)
    {
#pragma unroll
        for (int q = 0; q < 16; ++q)
            a[q] = fma(a[q], b, c);
    }
    double s = 0;
    for (int q = 0; q < 16; ++q)
        s += a[q];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    double *out;
    unsigned long long *cyc, h[1024];
    hipMalloc(&out, 2048 * 256 * 8);
    hipMalloc(&cyc, 1024 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    // what a "cycle" is: the shader clock the runtime reports (the data sheet's 78.6 TFLOP/s is 256 CUs x 4 SIMDs x
    // 32 flop per cycle at 2.4 GHz), and the fixed-rate counter s_memtime reads (NOT the shader clock)
    int sclk_khz = 0, wall_khz = 0, cus = 0;
    hipDeviceGetAttribute(&sclk_khz, hipDeviceAttributeClockRate, 0);
    hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    printf("device: %d CUs, peak shader clock %.0f MHz (hipDeviceAttributeClockRate), s_memtime / wall clock counter %.0f MHz\n",
           cus, sclk_khz / 1e3, wall_khz / 1e3);
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
        // MFMAs a SIMD executed: (wg * 4 waves / (cus * 4 SIMDs)) waves per SIMD x iters x 10; time per instruction at the
        // reported peak clock -- 64 cycles is what 78.6 TFLOP/s would need (2048 flop at 32 flop / cycle / SIMD)
        const double per_simd = (double)wg * 4 / (cus * 4.0) * iters * 10;
        printf("%4d WGs x 4 waves: %.3f ms, %.1f TFLOP/s fp64, %.1f ns = %.1f cycles at %.0f MHz per MFMA per SIMD (64 would be the "
               "data sheet's rate), %.1f s_memtime ticks per MFMA per wave\n",
               wg, ms, nm * 2048 / ms / 1e9, ms * 1e6 / per_simd, ms * 1e-3 * sclk_khz * 1e3 / per_simd, sclk_khz / 1e3,
               (double)h[0] / (iters * 10.0));
    }
    for (int wg = 512; wg <= 2048; wg *= 2)
    {
        const int vit = 20000;
        hipLaunchKernelGGL(valu_rate, dim3(wg), dim3(256), 0, 0, out, 100);
        hipEventRecord(e0);
        hipLaunchKernelGGL(valu_rate, dim3(wg), dim3(256), 0, 0, out, vit);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)wg * 256 * vit * 16 * 2;
        printf("v_fma_f64, %4d WGs x 4 waves: %.3f ms, %.1f TFLOP/s fp64 => the clock under vector fp64 load is about %.0f MHz "
               "(78.6 TFLOP/s at 2400)\n", wg, ms, flop / ms / 1e9, flop / ms / 1e9 / 78.6 * 2400.0);
    }
    return 0;
}
