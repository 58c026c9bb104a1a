// do v_mfma_f64_16x16x4_f64 and the vector pipe's fp64 FMAs overlap on a SIMD of gfx950?  (round 5: the wide pass has 48 of
// the former and ~600 of the latter per 64-row tile and runs at the SUM of their issue times, not the maximum)
//   A  MFMA only          : 1 wave per SIMD, 12 MFMAs per iteration (4 accumulators x 3)
//   B  FMA only           : 1 wave per SIMD, 192 v_fma_f64 per iteration (16 chains x 12)
//   C  both, one wave     : the 12 MFMAs and the 192 FMAs of an iteration in one instruction stream
//   D  both, two waves    : 2 waves per SIMD, waves 0-3 of a workgroup run A's loop, waves 4-7 run B's
// 12 MFMAs x 64 cycles = 768 and 192 FMAs x 4 cycles = 768: equal issue times; overlap would give C = D = A = B.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64_t __attribute__((ext_vector_type(4)));

template <bool MF, bool FM>
__device__ __forceinline__ double body(int iters, double seed)
{
    v4f64_t acc[4];
    for (int q = 0; q < 4; ++q)
        acc[q] = (v4f64_t){0, 0, 0, 0};
    double a[16];
    for (int q = 0; q < 16; ++q)
        a[q] = seed + q;
    const double ma = seed * 1e-3, mb = 1.0 + seed * 1e-4, b = 1.0 + seed * 1e-9, c = 1e-7;
    for (int it = 0; it < iters; ++it)
    {
#pragma unroll
        for (int r = 0; r < 3; ++r)
        {
#pragma unroll
            for (int q = 0; q < 4; ++q)
            {
                if (MF)
                    acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, acc[q], 0, 0, 0);
                if (FM)
                {
#pragma unroll
                    for (int k = 0; k < 16; ++k)
                        a[k] = fma(a[k], b, c);
                }
            }
        }
        if (MF && FM)
        {
            // one MFMA, then the sixteen FMAs it is written with -- the order of the source, pinned
#pragma unroll
            for (int g = 0; g < 12; ++g)
            {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);
            }
        }
    }
    double s = 0;
    for (int q = 0; q < 4; ++q)
        s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    for (int q = 0; q < 16; ++q)
        s += a[q];
    return s;
}

template <int MODE>
__global__ __launch_bounds__(512) void probe(double *out, int iters)
{
    const double seed = threadIdx.x * 1e-3;
    double s;
    if (MODE == 0)
        s = body<true, false>(iters, seed);
    else if (MODE == 1)
        s = body<false, true>(iters, seed);
    else if (MODE == 2)
        s = body<true, true>(iters, seed);
    else
        s = (threadIdx.x < 256) ? body<true, false>(iters, seed) : body<false, true>(iters, seed);
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE>
static float run(double *out, int wg, int threads, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<MODE>), dim3(wg), dim3(threads), 0, 0, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE>), dim3(wg), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    double *out;
    hipMalloc(&out, 4096 * 512 * 8);
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int iters = 20000;
    const float a = run<0>(out, cus, 256, iters), b = run<1>(out, cus, 256, iters), c = run<2>(out, cus, 256, iters), d = run<3>(out, cus, 512, iters);
    printf("%d CUs, one workgroup per CU, %d iterations of (12 MFMA f64 16x16x4 | 192 v_fma_f64)\n", cus, iters);
    printf("A  MFMA only, 1 wave / SIMD            : %.3f ms  (%.1f ns per MFMA per SIMD = %.1f TFLOP/s on %d SIMDs)\n", a, a * 1e6 / (iters * 12.0),
           (double)cus * 4 * iters * 12.0 * 2048 / a / 1e9, cus * 4);
    printf("B  FMA only, 1 wave / SIMD             : %.3f ms  (%.2f ns per v_fma_f64 = %.1f TFLOP/s)\n", b, b * 1e6 / (iters * 192.0),
           (double)cus * 4 * iters * 192.0 * 128 / b / 1e9);
    printf("C  both in one instruction stream      : %.3f ms  (A + B = %.3f, max = %.3f)\n", c, a + b, a > b ? a : b);
    printf("D  MFMA wave + FMA wave on each SIMD   : %.3f ms  (A + B = %.3f, max = %.3f)\n", d, a + b, a > b ? a : b);
    return 0;
}
