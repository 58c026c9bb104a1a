// wide_kernels.hpp -- the pass over the n residual rows for 10 <= p <= 64 parameters: J^T J on the matrix cores.
//
// north_star: "MFMA tall-skinny GEMM for J^T J only when p fills a 16-wide tile".  The reference materialises the
// n x p Jacobian (src/nls.c:266, :885-912) and GSL's Cholesky solver forms J^T J with dsyrk (lower); here J is never
// stored: a wavefront evaluates 64 rows (lane = row: residual f_i and the p gradient entries, analytic from the
// compiled formula or by the reference's forward / central differences, src/fdjac.c), parks them in an LDS tile and
// immediately contracts the tile with v_mfma_f64_16x16x4_f64:
//
//   * tile layout: column-major J^T: tile[k][row], leading dimension 68 doubles -- the row phase writes 64
//     consecutive doubles per gradient entry (conflict-free), the MFMA phase reads, for the 4-row chunk c, lane l
//     (kk = l / 16, i = l % 16) the entry J[4c + kk][16 b + i] of every 16-column block b: i * 68 + kk walks the
//     banks in steps of 4 doubles + {0..3}: two lanes per 8-byte bank, the minimum for 64 lanes;
//   * operand layout of the instruction (probed on gfx950, scripts/mfma_probe): lane l supplies A'[l % 16][l / 16]
//     and B[l / 16][l % 16] and holds D[4 r + l / 16][l % 16] in result register r.  The register loaded for block b
//     is at once the A' operand of block row b and the B operand of block column b, so NB = PW / 16 loads feed the
//     NB (NB + 1) / 2 lower-triangle blocks of one chunk;
//   * J^T f rides along on the vector pipe: the same registers times f[4c + kk], one FMA per block and chunk, the four
//     row groups of a lane column added at the end; ssr and the non-finite flag are per-lane sums of the row phase;
//   * PW = 16 ceil(p / 16): the padding columns of the tile are zeroed once and never written.
//
// Compiled in process for one formula (rtc_host.hpp): M = the generated row model, M::P = the actual p.
#pragma once
#if !defined(__HIPCC_RTC__)
#include <hip/hip_runtime.h>
#endif
#include "lm_core.hpp"
#include "devmath.hpp"
#include "rowops.hpp"
#include "wide_core.hpp"

namespace gslnls
{

constexpr int WIDE_LD = 68;     // leading dimension of the tile (doubles)
constexpr int WIDE_T = 256;     // threads per workgroup of the pass
constexpr int WIDE_MAX_G = 512; // workgroups == partial sets (two per CU where registers and LDS allow)

struct WPassArgs
{
    const double *x;  // n x NX column-major
    const double *y;
    const double *sw; // sqrt(weights) or nullptr
    long long n;
    const WState *state;
    double *partials; // [G][NV]: ssr, badj, packed lower J^T J, J^T f -- one contiguous set per workgroup
    double h_df, h_fvv;
    int fvv_analytic;
    int wf_only; // gsl_nls_large: the weights scale f only -- the reference's callback never weights J (src/nls_large.c:629-633)
};

typedef double wide_v4f64 __attribute__((ext_vector_type(4)));

// theta as the generated model reads it: th[k] with a compile-time k, from the workgroup's LDS copy
struct WideTheta
{
    const double *p;
    __device__ __forceinline__ double operator[](int k) const { return p[k]; }
};
// theta + d e_j (finite differences): j is a run-time value, k a literal
struct WideThetaPert
{
    const double *p;
    int j;
    double d;
    __device__ __forceinline__ double operator[](int k) const { return k == j ? p[k] + d : p[k]; }
};
// theta + h v (second directional derivative by differences, src/fdfvv.c:35-77)
struct WideThetaDir
{
    const double *p, *v;
    double h;
    __device__ __forceinline__ double operator[](int k) const { return p[k] + h * v[k]; }
};
// gradient entry k of this lane's row -> tile[k][lane], weighted; the non-finite flag as in row_fj
struct WideTileSink
{
    double *col; // &tile[0][lane]
    double sw;
    double bad;
    __device__ __forceinline__ void set(int k, double v)
    {
        bad = fma(v, 0.0, bad);
        col[k * WIDE_LD] = v * sw;
    }
};
struct WideGradSink
{
    double *dst; // &grad[i]
    long long n;
    double sw;
    __device__ __forceinline__ void set(int k, double v) { dst[(size_t)n * k] = v * sw; }
};

template <class M, class TH, class XR>
__device__ __forceinline__ double wide_resid(const TH &th, const XR &xr, double y, double sw)
{
    const double m = M::value(th, xr);
    const double f = isfinite(m) ? m - y : INFINITY;
    return f * sw;
}

#ifndef GSLNLS_WIDE_WAVES
#define GSLNLS_WIDE_WAVES(PW) ((PW) <= 32 ? 2 : 1) // workgroups per CU the register budget is cut for (LDS allows 2 up to PW = 32)
#endif
template <class M, int JAC, int PW>
__global__ __launch_bounds__(WIDE_T, GSLNLS_WIDE_WAVES(PW)) void wide_pass_kernel(WPassArgs a)
{
    constexpr int P = M::P, NX = M::NX, NB = PW / 16, NW = WIDE_T / 64, NQ = NB * (NB + 1) / 2;
    constexpr int NA = P * (P + 1) / 2, NV = 2 + NA + P;
    static_assert(P <= PW && PW <= 64 && PW % 16 == 0, "PW = 16 ceil(p / 16)");
    __shared__ double tile[NW][PW * WIDE_LD];
    __shared__ double ftile[NW][64];
    __shared__ double th_s[P], vel_s[P], delta_s[P];
    __shared__ double red_s[NW][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const WState *S = a.state;
    const int phase = S->phase;
    if (phase == PH_DONE)
        return;
    for (int k = tid; k < P; k += WIDE_T)
    {
        const double t = (phase == PH_FVV) ? S->x[k] : S->xt[k];
        th_s[k] = t;
        vel_s[k] = S->vel[k];
        double d = a.h_df * fabs(t); // src/fdjac.c:36-38
        if (d == 0.0)
            d = a.h_df;
        delta_s[k] = d;
    }
    for (int e = tid; e < NW * PW * WIDE_LD; e += WIDE_T)
        (&tile[0][0])[e] = 0.0;
    __syncthreads();
    const WideTheta th{th_s};
    double *const mytile = tile[wave];
    const int kk = lane >> 4, ii = lane & 15;

    wide_v4f64 acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
        acc[q] = (wide_v4f64){0.0, 0.0, 0.0, 0.0};
    double gacc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
        gacc[b] = 0.0;
    double ssr = 0.0, bad = 0.0;

    const long long ntile = (a.n + 63) / 64;
    for (long long t = (long long)blockIdx.x * NW + wave; t < ntile; t += (long long)gridDim.x * NW)
    {
        // ---------------- row phase: lane = row ----------------
        const long long i = t * 64 + lane;
        const bool live = i < a.n;
        const long long ic = live ? i : a.n - 1;
        double xr[NX];
#pragma unroll
        for (int c = 0; c < NX; ++c)
            xr[c] = a.x[(size_t)c * a.n + ic];
        const double yy = a.y[ic];
        const double sw = live ? (a.sw ? a.sw[ic] : 1.0) : 0.0;
        double f;
        if constexpr (JAC == JAC_ANALYTIC)
        {
            WideTileSink sink{mytile + lane, a.wf_only ? (live ? 1.0 : 0.0) : sw, bad};
            const double m = M::value_grad_sink(th, xr, sink);
            bad = sink.bad;
            f = (isfinite(m) ? m - yy : INFINITY) * sw;
        }
        else
        {
            f = wide_resid<M>(th, xr, yy, sw);
            for (int j = 0; j < P; ++j)
            {
                const double d = delta_s[j];
                double col;
                if constexpr (JAC == JAC_FORWARD)
                {
                    const double fn = wide_resid<M>(WideThetaPert{th_s, j, d}, xr, yy, sw);
                    col = (fn - f) * (1.0 / d);
                }
                else
                {
                    const double fp = wide_resid<M>(WideThetaPert{th_s, j, 0.5 * d}, xr, yy, sw);
                    const double fm = wide_resid<M>(WideThetaPert{th_s, j, -0.5 * d}, xr, yy, sw);
                    col = (fp - fm) * (1.0 / d);
                }
                mytile[j * WIDE_LD + lane] = live ? col : 0.0;
            }
        }
        if (phase == PH_FVV)
        {
            // second directional derivative along the velocity at x (src/fdf.c:200-233, FD form src/fdfvv.c:35-77);
            // the accumulated vector is J^T fvv, J^T J is not needed
            double fv;
            if (a.fvv_analytic)
            {
                const double r = M::fvv(th, WideTheta{vel_s}, xr);
                bad = fma(r, 0.0, bad);
                fv = r * sw;
            }
            else
            {
                double u = 0.0;
                for (int j = 0; j < P; ++j)
                    u += mytile[j * WIDE_LD + lane] * vel_s[j];
                const double fip = wide_resid<M>(WideThetaDir{th_s, vel_s, a.h_fvv}, xr, yy, sw);
                const double hinv = 1.0 / a.h_fvv;
                fv = (2.0 * hinv) * ((fip - f) * hinv - u);
            }
            f = fv;
        }
        f = live ? f : 0.0;
        ssr += f * f;
        ftile[wave][lane] = f;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        // ---------------- contraction phase: 16 chunks of 4 rows ----------------
#pragma unroll 4
        for (int c = 0; c < 16; ++c)
        {
            double v[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b)
                v[b] = mytile[(b * 16 + ii) * WIDE_LD + c * 4 + kk];
            const double fl = ftile[wave][c * 4 + kk];
#pragma unroll
            for (int b = 0; b < NB; ++b)
                gacc[b] = fma(v[b], fl, gacc[b]);
            if (phase != PH_FVV)
            {
                int q = 0;
#pragma unroll
                for (int ba = 0; ba < NB; ++ba)
#pragma unroll
                    for (int bb = 0; bb <= ba; ++bb, ++q)
                        acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(v[ba], v[bb], acc[q], 0, 0, 0);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }

    // ---------------- workgroup reduction -> one partial set (fixed order: wave 0, 1, 2, 3) ----------------
    __syncthreads(); // every wave is done with its tile: the tiles become the staging area
    double *const full = &tile[0][0]; // PW x PW doubles, row-major: (i, j) -> i * PW + j   (PW * PW <= NW * PW * 68)
    double *const gfull = &ftile[0][0]; // PW doubles (<= NW * 64)
    ssr = wave_sum_wide(ssr);
    bad = wave_sum_wide(bad);
    if (lane == 0)
    {
        red_s[wave][0] = ssr;
        red_s[wave][1] = bad;
    }
    for (int w = 0; w < NW; ++w)
    {
        if (wave == w)
        {
            int q = 0;
#pragma unroll
            for (int ba = 0; ba < NB; ++ba)
#pragma unroll
                for (int bb = 0; bb <= ba; ++bb, ++q)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                    {
                        const int e = (ba * 16 + 4 * r + kk) * PW + bb * 16 + ii;
                        full[e] = (w == 0) ? acc[q][r] : full[e] + acc[q][r];
                    }
            // J^T f: entry 16 b + i is the sum over the four row groups kk of a lane column, kk ascending
#pragma unroll
            for (int b = 0; b < NB; ++b)
            {
                double s = gacc[b];
                const double s1 = wide_shfl(s, ii + 16), s2 = wide_shfl(s, ii + 32), s3 = wide_shfl(s, ii + 48);
                s = ((wide_shfl(s, ii) + s1) + s2) + s3;
                if (kk == 0)
                    gfull[b * 16 + ii] = (w == 0) ? s : gfull[b * 16 + ii] + s;
            }
        }
        __syncthreads();
    }
    double *out = a.partials + (size_t)blockIdx.x * NV;
    if (tid == 0)
    {
        double s0 = red_s[0][0], s1 = red_s[0][1];
        for (int w = 1; w < NW; ++w)
        {
            s0 += red_s[w][0];
            s1 += red_s[w][1];
        }
        out[0] = s0;
        out[1] = s1;
    }
    for (int e = tid; e < NA; e += WIDE_T)
    {
        // packed index e -> (i, j), j <= i
        int i = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
        while (i * (i + 1) / 2 > e)
            --i;
        while ((i + 1) * (i + 2) / 2 <= e)
            ++i;
        const int j = e - i * (i + 1) / 2;
        out[2 + e] = full[i * PW + j];
    }
    for (int k = tid; k < P; k += WIDE_T)
        out[2 + NA + k] = gfull[k];
}

// After the fit: weighted residual and Jacobian at the final point in the layout C_nls returns them (resid n; grad
// n x p column-major, src/nls.c:695-737)
template <class M, int JAC>
__global__ __launch_bounds__(256) void wide_finalize_kernel(WPassArgs a, double *resid, double *grad)
{
    constexpr int P = M::P, NX = M::NX;
    __shared__ double th_s[P], delta_s[P];
    const WState *S = a.state;
    for (int k = threadIdx.x; k < P; k += 256)
    {
        const double t = S->x[k];
        th_s[k] = t;
        double d = a.h_df * fabs(t);
        if (d == 0.0)
            d = a.h_df;
        delta_s[k] = d;
    }
    __syncthreads();
    const WideTheta th{th_s};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (long long)gridDim.x * 256)
    {
        double xr[NX];
#pragma unroll
        for (int c = 0; c < NX; ++c)
            xr[c] = a.x[(size_t)c * a.n + i];
        const double yy = a.y[i], sw = a.sw ? a.sw[i] : 1.0;
        double f;
        if constexpr (JAC == JAC_ANALYTIC)
        {
            if (grad)
            {
                WideGradSink sink{grad + i, a.n, sw};
                const double m = M::value_grad_sink(th, xr, sink);
                f = (isfinite(m) ? m - yy : INFINITY) * sw;
            }
            else
                f = wide_resid<M>(th, xr, yy, sw);
        }
        else
        {
            f = wide_resid<M>(th, xr, yy, sw);
            if (grad)
                for (int j = 0; j < P; ++j)
                {
                    const double d = delta_s[j];
                    double col;
                    if constexpr (JAC == JAC_FORWARD)
                        col = (wide_resid<M>(WideThetaPert{th_s, j, d}, xr, yy, sw) - f) * (1.0 / d);
                    else
                        col = (wide_resid<M>(WideThetaPert{th_s, j, 0.5 * d}, xr, yy, sw) -
                               wide_resid<M>(WideThetaPert{th_s, j, -0.5 * d}, xr, yy, sw)) *
                              (1.0 / d);
                    grad[i + (size_t)a.n * j] = col;
                }
        }
        if (resid)
            resid[i] = f;
    }
}

} // namespace gslnls
