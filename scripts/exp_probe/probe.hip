// gexp (csrc/devmath.hpp) vs the device library's exp: bitwise comparison + VALU instruction timing
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
#include <random>
#include "../../gslnls_amd/csrc/devmath.hpp"
using namespace gslnls;
__global__ void cmp(const double *x, long long n, unsigned long long *ndiff, double *worst)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double a = exp(x[i]), b = gexp(x[i]);
    if (__double_as_longlong(a) != __double_as_longlong(b) && !(a != a && b != b))
    {
        atomicAdd(ndiff, 1ull);
        worst[0] = x[i]; worst[1] = a; worst[2] = b;
    }
}
template <int WHICH>
__global__ void timeit(double *out, double x0, int reps)
{
    double s = 0.0, x = x0 + threadIdx.x * 1e-3;
    for (int r = 0; r < reps; ++r)
    {
        s += WHICH ? gexp(x) : exp(x);
        x += 1e-6;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main()
{
    const long long n = 1 << 24;
    std::vector<double> h(n);
    std::mt19937_64 g(1);
    std::uniform_real_distribution<double> u(-760.0, 720.0), v(-5.0, 5.0);
    for (long long i = 0; i < n; ++i) h[i] = (i & 1) ? u(g) : v(g);
    const double sp[] = {0.0, -0.0, INFINITY, -INFINITY, NAN, 709.78, 709.79, 710.0, -745.13, -745.14, -708.4, -1e300, 1e300, 1e-320, 1024.0, -1075.0, 1099.0, 1101.0};
    for (size_t k = 0; k < sizeof(sp) / 8; ++k) h[k] = sp[k];
    double *dx, *dw; unsigned long long *dn;
    hipMalloc(&dx, n * 8); hipMalloc(&dw, 64); hipMalloc(&dn, 8);
    hipMemcpy(dx, h.data(), n * 8, hipMemcpyHostToDevice); hipMemset(dn, 0, 8); hipMemset(dw, 0, 64);
    hipLaunchKernelGGL(cmp, dim3((n + 255) / 256), dim3(256), 0, 0, dx, n, dn, dw);
    unsigned long long nd; double w[3];
    hipMemcpy(&nd, dn, 8, hipMemcpyDeviceToHost); hipMemcpy(w, dw, 24, hipMemcpyDeviceToHost);
    printf("bitwise differences: %llu of %lld (last: x=%.17g exp=%.17g gexp=%.17g)\n", nd, n, w[0], w[1], w[2]);
    double *out; hipMalloc(&out, 256 * 1024 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 2; ++which)
        for (int rep = 0; rep < 2; ++rep)
        {
            hipEventRecord(e0);
            if (which) hipLaunchKernelGGL(timeit<1>, dim3(1024), dim3(256), 0, 0, out, -1.0, 4096);
            else hipLaunchKernelGGL(timeit<0>, dim3(1024), dim3(256), 0, 0, out, -1.0, 4096);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%s: %.3f ms for %.2e evaluations -> %.1f Gexp/s\n", which ? "gexp" : "exp ", ms, 1024.0 * 256 * 4096, 1024.0 * 256 * 4096 / ms / 1e6);
        }
    return 0;
}
