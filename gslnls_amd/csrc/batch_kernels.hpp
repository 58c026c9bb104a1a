// batch_kernels.hpp -- lane-per-fit kernel for multi-start (BASELINE config C4: thousands of
// independent starts on a tiny data set, e.g. BoxBOD n = 6, p = 2).
//
// Each lane owns one sample slot of gsl_multistart_driver's point loop
// (src/nls_mstart.c:42-128 / :236-349): it generates its own Sobol point from the global
// draw index (sobol.hpp), maps it to the sampling range, runs the det filter and the short
// LM fit entirely in registers (batch_core.hpp) and writes one fixed-size record.  The data
// set is staged once per workgroup in LDS and read as broadcasts (every lane reads the same
// row at the same time: conflict-free), so the kernel touches HBM only for the records.
// Bound by fp64 VALU + exp latency, not by memory.
#pragma once
#include <hip/hip_runtime.h>
#include "batch_core.hpp"
#include "models.hpp"

namespace gslnls
{

constexpr int MS_T = 64;           // one wavefront per workgroup: lanes diverge per fit anyway
constexpr int MS_LDS_ROWS = 1536;  // rows staged in LDS (x NX + y + sw doubles each)

template <int P>
struct MsKernelArgs
{
    const double *x[4];
    const double *y;
    const double *sw;
    const long long *draw; // per point, -1 = explicit start; nullptr = consecutive draws first_draw + idx
    long long first_draw;
    const double *start;   // count x P
    double *records;       // count x K
    const SobolTable *sobol;
    int lo, hi;
    double l0[P], l1[P], kd[P];
    double lu[2 * P];
    int has_lu;
    MsParams mp;
};

// rows of a tiny data set (BoxBOD: 6) in registers: the LDS round trip in front of every row is a third of a
// pass when a row costs ~50 instructions
constexpr int MS_REG_ROWS = 8;

// lane exchange inside a quad (DPP quad_perm: a VALU move, no LDS)
template <int CTRL>
__device__ __forceinline__ double ms_dpp_mov(double v)
{
    const long long bits = __double_as_longlong(v);
    int lo = (int)(bits & 0xffffffffll), hi = (int)(bits >> 32);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// LPF lanes (1, 2 or 4; groups are aligned inside quads) share one fit: see ms_pass
template <int NX, int LPF_>
struct RowsReg
{
    static constexpr int STATIC_N = MS_REG_ROWS;
    static constexpr int LPF = LPF_;
    double x[MS_REG_ROWS][NX], y[MS_REG_ROWS], sw[MS_REG_ROWS];
    int sub; // this lane's place in its group
    template <int P>
    __device__ void combine(PassSums<P> &acc) const
    {
        double *v = reinterpret_cast<double *>(&acc);
#pragma unroll
        for (int k = 0; k < PassSums<P>::NV; ++k)
        {
            if (LPF >= 2)
                v[k] += ms_dpp_mov<0xB1>(v[k]); // quad_perm [1,0,3,2]
            if (LPF >= 4)
                v[k] += ms_dpp_mov<0x4E>(v[k]); // quad_perm [2,3,0,1]
        }
    }
};

template <int NX>
struct RowsLds
{
    static constexpr int STATIC_N = 0, LPF = 1;
    const double *base;
    int n;
    __device__ void operator()(int i, double *xr, double &y, double &sw) const
    {
#pragma unroll
        for (int c = 0; c < NX; ++c)
            xr[c] = base[c * n + i];
        y = base[NX * n + i];
        sw = base[(NX + 1) * n + i];
    }
};

template <int NX>
struct RowsGlobal
{
    static constexpr int STATIC_N = 0, LPF = 1;
    const double *x[4];
    const double *y;
    const double *sw;
    __device__ void operator()(int i, double *xr, double &yy, double &w) const
    {
#pragma unroll
        for (int c = 0; c < NX; ++c)
            xr[c] = x[c][i];
        yy = y[i];
        w = sw ? sw[i] : 1.0;
    }
};

// LPF > 1 (small batches of tiny data sets, chosen by the host): LPF lanes per fit, the workgroup covers MS_T / LPF points
template <class M, int JAC, int LPF = 1>
__global__ __launch_bounds__(MS_T) void ms_fit_kernel(MsKernelArgs<M::P> a)
{
    constexpr int P = M::P, NX = M::NX;
    extern __shared__ __attribute__((aligned(16))) double lds_rows[];
    const int n = a.mp.n;
    const bool staged = n <= MS_LDS_ROWS;
    if (staged)
    {
        for (int i = threadIdx.x; i < n; i += MS_T)
        {
#pragma unroll
            for (int c = 0; c < NX; ++c)
                lds_rows[c * n + i] = a.x[c][i];
            lds_rows[NX * n + i] = a.y[i];
            lds_rows[(NX + 1) * n + i] = a.sw ? a.sw[i] : 1.0;
        }
        __syncthreads();
    }
    // The P columns of the direction-number table go through LDS as well: read from global memory inside
    // sobol_coord's bit loop they are up to 30 dependent cold loads per coordinate, ~10 us in front of every fit.
    __shared__ unsigned int lds_sobol[SOBOL_BITS][P];
    const bool halton = a.sobol->halton != 0;
    if (!halton)
    {
        for (int e = threadIdx.x; e < SOBOL_BITS * P; e += MS_T)
            lds_sobol[e / P][e % P] = a.sobol->v[e / P][e % P];
        __syncthreads();
    }
    const int idx = a.lo + (blockIdx.x * MS_T + threadIdx.x) / LPF;
    if (idx >= a.hi)
        return; // (whole groups: idx is the same for the LPF lanes of a group)
    double start[P];
    const long long d = a.draw ? a.draw[idx] : a.first_draw + idx;
#pragma unroll
    for (int k = 0; k < P; ++k)
    {
        if (d >= 0)
        {
            double u;
            if (halton)
                u = sobol_coord(*a.sobol, (unsigned int)d, k);
            else
            {
                // same XOR as sobol_coord, over all 30 bits with the unset ones masked out: 30 independent
                // broadcast reads instead of a data-dependent loop
                const unsigned int kk = (unsigned int)d + 1u;
                const unsigned int g = kk ^ (kk >> 1);
                unsigned int num = 0;
#pragma unroll
                for (int b = 0; b < SOBOL_BITS; ++b)
                    num ^= lds_sobol[b][k] & (0u - ((g >> b) & 1u));
                u = (double)num * (1.0 / 1073741824.0); // 2^-30
            }
            start[k] = sobol_to_range(u, a.l0[k], a.l1[k], a.kd[k]);
        }
        else
            start[k] = a.start[(size_t)idx * P + k];
    }
    MsRecord<P> rec;
    if (staged && n <= MS_REG_ROWS)
    {
        RowsReg<NX, LPF> rows;
        rows.sub = threadIdx.x % LPF;
#pragma unroll
        for (int i = 0; i < MS_REG_ROWS; ++i)
        {
            const int ic = i < n ? i : 0;
#pragma unroll
            for (int c = 0; c < NX; ++c)
                rows.x[i][c] = lds_rows[c * n + ic];
            rows.y[i] = lds_rows[NX * n + ic];
            rows.sw[i] = lds_rows[(NX + 1) * n + ic];
        }
        ms_fit_point<M, JAC>(a.mp, rows, start, a.has_lu ? a.lu : nullptr, rec);
    }
    else if (staged)
    {
        RowsLds<NX> rows{lds_rows, n};
        ms_fit_point<M, JAC>(a.mp, rows, start, a.has_lu ? a.lu : nullptr, rec);
    }
    else
    {
        RowsGlobal<NX> rows;
#pragma unroll
        for (int c = 0; c < NX; ++c)
            rows.x[c] = a.x[c];
        rows.y = a.y;
        rows.sw = a.sw;
        ms_fit_point<M, JAC>(a.mp, rows, start, a.has_lu ? a.lu : nullptr, rec);
    }
    if (LPF > 1 && threadIdx.x % LPF != 0)
        return; // every lane of the group holds the same record
    double *out = a.records + (size_t)idx * MsRecord<P>::K;
    const double *src = reinterpret_cast<const double *>(&rec);
#pragma unroll
    for (int k = 0; k < MsRecord<P>::K; ++k)
        out[k] = src[k];
}

} // namespace gslnls
