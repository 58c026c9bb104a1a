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

// ---------------------------------------------------------------------------------------------------------
// Lane refill: in the kernel above a lane is idle from the moment its point is finished until the slowest fit of
// the wavefront ends -- on BoxBOD about 35 % of the lane-cycles do work (every lane walks its own sequence of
// trials, up to 16 rejected ones per iteration).  Here a wavefront OWNS a contiguous slice of the batch and a lane
// that finishes takes the next point of the slice: lanes that finish in the same trip are served in lane order
// (ballot + popcount below the lane), so the assignment is deterministic given the fits, and since every record is
// written to its point's own place and a fit depends on nothing but its point, the records are bit-identical to the
// one-point-per-lane kernel's.  The slice length (points per wavefront) is chosen by the host: batch / waves.
template <class M, int JAC>
__global__ __launch_bounds__(MS_T, 2) void ms_fit_refill_kernel(MsKernelArgs<M::P> a, int slice, int thresh)
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
    __shared__ unsigned int lds_sobol[SOBOL_BITS][P];
    const bool halton = a.sobol->halton != 0;
    if (!halton)
    {
        for (int e = threadIdx.x; e < SOBOL_BITS * P; e += MS_T)
            lds_sobol[e / P][e % P] = a.sobol->v[e / P][e % P];
        __syncthreads();
    }
    const int lane = threadIdx.x;
    const long long s0 = (long long)a.lo + (long long)blockIdx.x * slice;
    const long long s1 = (s0 + slice < a.hi) ? s0 + slice : a.hi;
    if (s0 >= s1)
        return;
    auto start_of = [&](long long idx, double *start) {
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
    };
    const double *lu = a.has_lu ? a.lu : nullptr;
    auto run = [&](const auto &rows) {
        long long idx = s0 + lane, nxt = s0 + MS_T;
        bool active = idx < s1;
        MsPointState<P> q;
        PassSums<P> acc;
        {
            double st[P];
            start_of(active ? idx : s0, st);
            ms_point_begin<P>(q, st, lu);
        }
        // every trip: one pass + one step of the state machine for the lanes that hold a point; bounded like the
        // one-point kernel (a point ends after at most maxiter * 17 + 2 trips)
        // (... times the points of the slice: in the worst case one lane ends up with all of them)
        const long long max_trips = ((long long)a.mp.prm.maxiter * 17 + 3) * (a.mp.prm.trs ? 2 : 1) * (long long)slice + 8;
        // A finished lane parks (`pending`) until `thresh` lanes have finished or nobody is left working: writing the
        // record and setting up the next point (Sobol bits, range map, det of the final J^T J, 3p + 8 stores) is a few
        // hundred instructions that would otherwise run in almost every trip for a handful of lanes.
        bool pending = false;
        for (long long trip = 0; trip < max_trips; ++trip)
        {
            if (active && !pending)
            {
                ms_pass<M, JAC>(q.s, a.mp, rows, acc);
                pending = ms_point_step<P>(q, a.mp, acc);
            }
            const unsigned long long mask = __ballot(pending);
            const unsigned long long working = __ballot(active && !pending);
            if (mask && (__popcll(mask) >= thresh || !working))
            {
                if (pending)
                {
                    MsRecord<P> rec;
                    ms_point_record<P>(q, acc, rec);
                    double *out = a.records + (size_t)idx * MsRecord<P>::K;
                    const double *src = reinterpret_cast<const double *>(&rec);
#pragma unroll
                    for (int k = 0; k < MsRecord<P>::K; ++k)
                        out[k] = src[k];
                    const int rank = __popcll(mask & ((1ull << lane) - 1ull));
                    idx = nxt + rank;
                    active = idx < s1;
                    pending = false;
                    if (active)
                    {
                        double st[P];
                        start_of(idx, st);
                        ms_point_begin<P>(q, st, lu);
                    }
                }
                nxt += __popcll(mask);
            }
            if (!__ballot(active))
                break;
        }
    };
    if (staged && n <= MS_REG_ROWS)
    {
        RowsReg<NX, 1> rows;
        rows.sub = 0;
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
        run(rows);
    }
    else if (staged)
    {
        RowsLds<NX> rows{lds_rows, n};
        run(rows);
    }
    else
    {
        RowsGlobal<NX> rows;
#pragma unroll
        for (int c = 0; c < NX; ++c)
            rows.x[c] = a.x[c];
        rows.y = a.y;
        rows.sw = a.sw;
        run(rows);
    }
}

} // namespace gslnls
