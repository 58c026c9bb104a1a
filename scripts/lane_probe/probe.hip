// developer probe (gfx950): semantics of v_permlane16_swap / v_permlane32_swap and the cost of a dependent chain on ONE
// wavefront -- v_max_f64, DPP moves, VALU -> SGPR -> VALU round trips, ds_bpermute -- the numbers behind the layout of
// wide_solve_reg (wide_core.hpp).  Build: hipcc --offload-arch=gfx950 -O3 -o probe probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void sem_kernel(unsigned *out)
{
    const unsigned l = threadIdx.x;
    unsigned a = l, b = 100 + l;
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    auto s = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[l] = r[0];
    out[64 + l] = r[1];
    out[128 + l] = s[0];
    out[192 + l] = s[1];
}

template <int CTRL>
__device__ __forceinline__ double dpp64(double v)
{
    const long long bits = __double_as_longlong(v);
    int lo = (int)(bits & 0xffffffffll), hi = (int)(bits >> 32);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double rl64(double v, int lane)
{
    const long long bits = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(bits & 0xffffffffll), lane);
    const int hi = __builtin_amdgcn_readlane((int)(bits >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double vmax(double a, double b)
{
    double r;
    asm volatile("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// kind: 0 dependent v_max_f64; 1 dependent v_fma_f64; 2 dpp stage (2 mov_dpp + max); 3 readlane x2 -> max (SGPR round trip);
// 4 ds_bpermute x2 -> max; 5 permlane32_swap x2 -> max; 6 dependent v_add_u32; 7 8 independent fma chains (issue rate);
// 8 v_cmp -> ballot -> s_ff1 -> readlane (argmax style round trip)
template <int KIND>
__global__ void chain_kernel(double *out, const double *in, int reps, long long *cycles)
{
    const int l = threadIdx.x;
    double x = in[l], y = in[64 + l];
    double z[8];
    for (int k = 0; k < 8; ++k)
        z[k] = in[128 + 64 * k + l];
    unsigned u = (unsigned)l;
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r)
    {
#pragma unroll
        for (int k = 0; k < 16; ++k)
        {
            if constexpr (KIND == 0)
                x = vmax(x, y);
            else if constexpr (KIND == 1)
                asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x) : "v"(y));
            else if constexpr (KIND == 2)
                x = vmax(x, dpp64<0xB1>(x));
            else if constexpr (KIND == 3)
                x = vmax(y, rl64(x, 16));
            else if constexpr (KIND == 4)
            {
                const long long bits = __double_as_longlong(x);
                const int lo = __builtin_amdgcn_ds_bpermute(((l ^ 32) << 2), (int)(bits & 0xffffffffll));
                const int hi = __builtin_amdgcn_ds_bpermute(((l ^ 32) << 2), (int)(bits >> 32));
                x = vmax(x, __longlong_as_double(((long long)hi << 32) | (unsigned int)lo));
            }
            else if constexpr (KIND == 5)
            {
                const long long bits = __double_as_longlong(x);
                const unsigned lo = (unsigned)(bits & 0xffffffffll), hi = (unsigned)(bits >> 32);
                auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
                auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
                const double p0 = __longlong_as_double(((long long)b[0] << 32) | a[0]);
                const double p1 = __longlong_as_double(((long long)b[1] << 32) | a[1]);
                x = vmax(p0, p1);
            }
            else if constexpr (KIND == 6)
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(u) : "v"(l));
            else if constexpr (KIND == 7)
            {
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(z[q]) : "v"(y));
            }
            else if constexpr (KIND == 8)
            {
                const unsigned long long hit = __builtin_amdgcn_ballot_w64(x == y);
                const int q = hit ? (int)__builtin_ctzll(hit) : 0;
                x = rl64(y, q) + x;
            }
        }
    }
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    double s = x + (double)u;
    for (int k = 0; k < 8; ++k)
        s += z[k];
    out[l] = s;
    if (l == 0)
        cycles[0] = t1 - t0;
}

template <int KIND>
static void run(const char *name, int per_iter)
{
    double *d_in, *d_out;
    long long *d_c;
    hipMalloc(&d_in, sizeof(double) * 64 * 10);
    hipMalloc(&d_out, sizeof(double) * 64);
    hipMalloc(&d_c, sizeof(long long));
    std::vector<double> h(640);
    for (int i = 0; i < 640; ++i)
        h[i] = 1.0 + 1e-3 * i;
    hipMemcpy(d_in, h.data(), sizeof(double) * 640, hipMemcpyHostToDevice);
    const int reps = 2000;
    for (int w = 0; w < 2; ++w)
        hipLaunchKernelGGL(chain_kernel<KIND>, dim3(1), dim3(64), 0, 0, d_out, d_in, reps, d_c);
    long long c = 0;
    hipMemcpy(&c, d_c, sizeof c, hipMemcpyDeviceToHost);
    printf("%-56s %7.1f shader clocks per link (%d per unrolled step)\n", name, (double)c / (reps * 16.0 * per_iter), per_iter);
    hipFree(d_in);
    hipFree(d_out);
    hipFree(d_c);
}

int main()
{
    unsigned *d;
    hipMalloc(&d, sizeof(unsigned) * 256);
    hipLaunchKernelGGL(sem_kernel, dim3(1), dim3(64), 0, 0, d);
    unsigned h[256];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char *names[4] = {"permlane16_swap(a=l, b=100+l) -> [0]", "permlane16_swap -> [1]", "permlane32_swap -> [0]", "permlane32_swap -> [1]"};
    for (int k = 0; k < 4; ++k)
    {
        printf("%s:", names[k]);
        for (int l = 0; l < 64; l += 4)
            printf(" %u", h[64 * k + l]);
        printf("\n");
    }
    run<0>("dependent v_max_f64", 1);
    run<1>("dependent v_fma_f64", 1);
    run<2>("2 x v_mov_b32_dpp + v_max_f64 (one butterfly stage)", 1);
    run<3>("2 x v_readlane -> v_max_f64 with the SGPR pair", 1);
    run<4>("2 x ds_bpermute -> v_max_f64", 1);
    run<5>("2 x v_permlane32_swap -> v_max_f64", 1);
    run<6>("dependent v_add_u32", 1);
    run<7>("8 independent v_fma_f64 chains (per instruction)", 8);
    run<8>("v_cmp -> ballot -> s_ff1 -> 2 x v_readlane -> v_add_f64", 1);
    return 0;
}
