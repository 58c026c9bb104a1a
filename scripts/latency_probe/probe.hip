// Dependent-issue latency of the instructions lm_advance is made of: one wave per block runs a chain of
// 1024 dependent instructions of one kind (and the same with 2 and 4 independent chains); s_memtime around it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define R4(x) x x x x
#define R256(x) R4(R4(R4(R4(x))))
#define STAMP(t) asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")
__global__ void probe(double *out, unsigned long long *st)
{
    double a = threadIdx.x + 1.5, b = 1.25, c = 2.5, d = 3.5, e = 1.0000001;
    unsigned long long t0, t1;
    int k = 0;
#define RUN(code)                                    \
    STAMP(t0);                                       \
    R256(asm volatile(code : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));) \
    STAMP(t1);                                       \
    if (threadIdx.x == 0) st[blockIdx.x * 16 + k] = t1 - t0; \
    ++k;
    RUN("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %0, %0, %4, %4")       // 0: 1 chain
    RUN("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4")       // 1: 2 chains
    RUN("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4")       // 2: 4 chains
    RUN("v_mul_f64 %0, %0, %4\n v_mul_f64 %0, %0, %4\n v_mul_f64 %0, %0, %4\n v_mul_f64 %0, %0, %4")                       // 3
    RUN("v_add_f64 %0, %0, %4\n v_add_f64 %0, %0, %4\n v_add_f64 %0, %0, %4\n v_add_f64 %0, %0, %4")                       // 4
    RUN("v_rcp_f64 %0, %0\n v_rcp_f64 %0, %0\n v_rcp_f64 %0, %0\n v_rcp_f64 %0, %0")                                       // 5
    RUN("v_rsq_f64 %0, %0\n v_rsq_f64 %0, %0\n v_rsq_f64 %0, %0\n v_rsq_f64 %0, %0")                                       // 6
    RUN("v_max_f64 %0, %0, %4\n v_max_f64 %0, %0, %4\n v_max_f64 %0, %0, %4\n v_max_f64 %0, %0, %4")                       // 7
    RUN("v_min_f64 %0, %0, %4\n v_min_f64 %0, %0, %4\n v_min_f64 %0, %0, %4\n v_min_f64 %0, %0, %4") // 8
    RUN("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3")                                       // 9: 4 indep rcp
    RUN("v_fma_f64 %0, %0, %4, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4")                               // 10: fma chain with 3 independent fillers
    out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d;
}
int main()
{
    double *out; unsigned long long *st;
    const int G = 8;
    hipMalloc(&out, G * 64 * 8); hipMalloc(&st, G * 16 * 8);
    std::vector<unsigned long long> h(G * 16);
    for (int launch = 0; launch < 3; ++launch)
    {
        hipLaunchKernelGGL(probe, dim3(G), dim3(64), 0, 0, out, st);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), st, G * 16 * 8, hipMemcpyDeviceToHost);
    }
    const char *nm[] = {"fma64 1 chain", "fma64 2 chains", "fma64 4 chains", "mul64 dep", "add64 dep", "rcp64 dep", "rsq64 dep", "max64 dep", "min64 dep", "rcp64 4 indep", "fma64 dep + 3 indep add"};
    for (int k = 0; k < 11; ++k)
        printf("%-22s %.2f cycles per instruction (1024 instructions)\n", nm[k], (double)h[k] / 1024.0);
    return 0;
}
