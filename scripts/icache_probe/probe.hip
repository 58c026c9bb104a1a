// Is the instruction cache cold at every kernel launch?  One wave per block runs a long straight-line block
// of independent fp64 FMAs twice inside the same launch; s_memtime around each trip.  Launched repeatedly.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define R8(x) x x x x x x x x
#define BODY R8(R8(R8(asm volatile("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)))
__global__ void probe(double *out, unsigned long long *stamps, int trips)
{
    double a = threadIdx.x, b = 1.0, c = 2.0, d = 3.0, e = 1.0000001;
    for (int t = 0; t < trips; ++t)
    {
        unsigned long long t0, t1;
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        BODY  // 512 x 4 = 2048 FMAs = 16 KB of code
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        if (threadIdx.x == 0)
            stamps[blockIdx.x * 8 + t] = t1 - t0;
    }
    out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d;
}
int main()
{
    double *out;
    unsigned long long *st;
    const int G = 256;
    hipMalloc(&out, G * 64 * 8);
    hipMalloc(&st, G * 8 * 8);
    std::vector<unsigned long long> h(G * 8);
    for (int launch = 0; launch < 6; ++launch)
    {
        hipLaunchKernelGGL(probe, dim3(G), dim3(64), 0, 0, out, st, 3);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), st, G * 8 * 8, hipMemcpyDeviceToHost);
        double m[3] = {0, 0, 0};
        for (int b = 0; b < G; ++b)
            for (int t = 0; t < 3; ++t)
                m[t] += (double)h[b * 8 + t] / G;
        printf("launch %d: mean cycles trip0 %.0f trip1 %.0f trip2 %.0f (2048 fp64 FMAs, 16 KB code; issue-bound = 8192)\n", launch, m[0], m[1], m[2]);
    }
    return 0;
}
