// developer probe: lm_solve_wave against lm_solve, bit for bit, and their cycle counts
// hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=on -I../../gslnls_amd/csrc -o probe probe.hip && ./probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "lm_core.hpp"
#ifndef GSLNLS_HAVE_SOLVE_WAVE
#include "lm_solve_wave.hpp"
#endif
using namespace gslnls;

template <int P, int WHICH>
__global__ __launch_bounds__(64) void solve_kernel(const double *A, const double *diag, const double *mu, const double *rhs, double *sol,
                                                  unsigned long long *cyc, int reps)
{
    constexpr int NT = P * (P + 1) / 2;
    const int t = blockIdx.x + (threadIdx.x >> 6); // the same for all 64 lanes, but not provably: values stay in VGPRs as in the real kernels
    double a[NT], d[P], r[P], s[P];
    for (int i = 0; i < NT; ++i) a[i] = A[t * NT + i];
    for (int i = 0; i < P; ++i) { d[i] = diag[t * P + i]; r[i] = rhs[t * P + i]; s[i] = 0.0; }
    double m = mu[t];
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int rep = 0; rep < reps; ++rep)
    {
#if defined(__HIP_DEVICE_COMPILE__)
        if (WHICH == 0) lm_solve<P>(a, d, m, r, s);
        else lm_solve_wave<P>(a, d, m, r, s);
#endif
        if (reps > 1) { m += s[0] * 1e-300; }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0)
    {
        for (int i = 0; i < P; ++i) sol[t * P + i] = s[i];
        cyc[t] = (t1 - t0) / reps;
    }
}

template <int P> int run(int trials)
{
    constexpr int NT = P * (P + 1) / 2;
    std::vector<double> A(trials * NT), D(trials * P), R(trials * P), MU(trials);
    for (int t = 0; t < trials; ++t)
    {
        double J[40][P];
        for (int i = 0; i < 40; ++i) for (int k = 0; k < P; ++k) J[i][k] = (rand() / (double)RAND_MAX - 0.5) * (k % 3 == 0 ? 1e3 : 1.0);
        if (t % 5 == 0) for (int i = 0; i < 40; ++i) J[i][P - 1] = J[i][0] * 2.0;
        if (t % 7 == 0) for (int i = 0; i < 40; ++i) J[i][1] = 0.0;
        for (int i = 0; i < P; ++i) for (int j = 0; j <= i; ++j) { double v = 0; for (int r = 0; r < 40; ++r) v += J[r][i] * J[r][j]; A[t * NT + tri(i, j)] = v; }
        for (int k = 0; k < P; ++k) { D[t * P + k] = (t % 7 == 0 && k == 1) ? 1.0 : sqrt(A[t * NT + tri(k, k)]); R[t * P + k] = rand() / (double)RAND_MAX - 0.5; }
        if (t % 11 == 0) A[t * NT + 2] = NAN;
        MU[t] = (t % 3) ? 1e-3 : 0.0;
    }
    double *dA, *dD, *dR, *dMU, *dS0, *dS1; unsigned long long *dC0, *dC1;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dD, D.size() * 8); hipMalloc(&dR, R.size() * 8); hipMalloc(&dMU, MU.size() * 8);
    hipMalloc(&dS0, R.size() * 8); hipMalloc(&dS1, R.size() * 8); hipMalloc(&dC0, trials * 8); hipMalloc(&dC1, trials * 8);
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dD, D.data(), D.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dR, R.data(), R.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dMU, MU.data(), MU.size() * 8, hipMemcpyHostToDevice);
    solve_kernel<P, 0><<<trials, 64>>>(dA, dD, dMU, dR, dS0, dC0, 1);
    solve_kernel<P, 1><<<trials, 64>>>(dA, dD, dMU, dR, dS1, dC1, 1);
    std::vector<double> S0(trials * P), S1(trials * P);
    hipMemcpy(S0.data(), dS0, S0.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(S1.data(), dS1, S1.size() * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < trials; ++t) if (memcmp(&S0[t * P], &S1[t * P], P * 8)) { if (bad < 3) { printf("  trial %d:", t); for (int k = 0; k < P; ++k) printf(" %.17g/%.17g", S0[t*P+k], S1[t*P+k]); printf("\n"); } ++bad; }
    solve_kernel<P, 0><<<1, 64>>>(dA + NT, dD + P, dMU + 1, dR + P, dS0, dC0, 50);
    solve_kernel<P, 1><<<1, 64>>>(dA + NT, dD + P, dMU + 1, dR + P, dS1, dC1, 50);
    unsigned long long c0, c1;
    hipMemcpy(&c0, dC0, 8, hipMemcpyDeviceToHost); hipMemcpy(&c1, dC1, 8, hipMemcpyDeviceToHost);
    printf("P=%d: %d of %d solutions differ | cycles per solve (s_memtime, one wave alone): registers %llu, lanes %llu\n", P, bad, trials, c0, c1);
    return bad;
}
int main() { int bad = run<5>(3000) + run<6>(3000) + run<7>(3000) + run<8>(3000); printf(bad ? "FAILED\n" : "ok\n"); return bad != 0; }
