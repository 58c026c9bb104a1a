// large_kernels.hpp -- matrix-free kernels behind gsl_nls_large() (gsl_multilarge_nlinear, cgst).
//
// In the reference every product with the Jacobian goes through gsl_df_large
// (src/nls_large.c:474-653): it RE-EVALUATES the whole R Jacobian closure, copies it element by
// element (:497-498, :524-526) and then calls dgemv (:629) or dsyrk (:633).  Here the Jacobian
// is never formed; two kinds of pass over the rows produce everything the trust-region /
// Steihaug-Toint iteration needs:
//
//   EVAL pass at a point x:   f = sqrt(w)(model - y),  ssr = f.f,  g = J^T f,  diag(J^T J)
//                             (trial evaluation and, speculatively, what an accepted step needs)
//   JTJV pass at x with u:    w = J u,  ||w||^2,  J^T w     (one CG iteration: the reference's
//                             NoTrans + Trans pair fused, A is read once instead of twice)
//
// Two model classes:
//   * registered row models (small p): thread-per-row, Jacobian row recomputed in registers;
//   * the dense GLM family f_i = exp(a_i . theta) with A (n x p, row-major fp64) resident in HBM
//     (BASELINE config C3: n = 1e7, p = 64, 5.12 GB): tiled kernel below, J = diag(m) A.
#pragma once
#include <hip/hip_runtime.h>
#include "dense_kernels.hpp"

namespace gslnls
{

enum
{
    LG_EVAL = 0,
    LG_JTJV = 1
};

// ------------------------------------------------------------------------------------------------
// row models: sums land in PassSums<P> (EVAL: ssr, J^T J, J^T f;  JTJV: ssr slot = ||J u||^2,
// g slots = J^T J u).  Weights scale f only -- the reference's callback never weights J
// (src/nls_large.c:629-633, quirk preserved).
template <class M, int T>
__global__ __launch_bounds__(T) void large_row_kernel(DenseCtx<M::P> ctx, int mode, const double *xpt,
                                                      const double *uvec, double *partials)
{
    constexpr int P = M::P, NX = M::NX;
    using Sums = PassSums<P>;
    constexpr int NV = Sums::NV;
    __shared__ double lds_red[(T / 64) * NV];
    double th[P], u[P], delta[P];
#pragma unroll
    for (int k = 0; k < P; ++k)
    {
        th[k] = xpt[k];
        u[k] = uvec ? uvec[k] : 0.0;
        delta[k] = 0.0;
    }
    Sums acc;
    pass_zero<P>(acc);
    const long long n = ctx.n, stride = (long long)gridDim.x * T;
    for (long long i = (long long)blockIdx.x * T + threadIdx.x; i < n; i += stride)
    {
        double xr[NX];
#pragma unroll
        for (int c = 0; c < NX; ++c)
            xr[c] = ctx.x[c][i];
        double Jrow[P];
        const double f0 = row_fj<M, JAC_ANALYTIC>(th, delta, xr, ctx.y[i], 1.0, Jrow, &acc.badj);
        if (mode == LG_EVAL)
        {
            const double f = f0 * (ctx.sw ? ctx.sw[i] : 1.0);
            acc_fj<P>(acc, f, Jrow);
        }
        else
        {
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < P; ++k)
                t += Jrow[k] * u[k];
            acc.ssr += t * t;
#pragma unroll
            for (int k = 0; k < P; ++k)
                acc.g[k] += Jrow[k] * t;
        }
    }
    const double tot = block_sum_slots<NV, T>(reinterpret_cast<const double *>(&acc), lds_red);
    if (threadIdx.x < NV)
        partials[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = tot;
}

// fixed-order sum of the per-workgroup partials: out[v] = sum_b partials[v][b]
static __global__ __launch_bounds__(64) void large_reduce_kernel(const double *partials, int nv, int nblk, double *out)
{
    const int v = blockIdx.x;
    if (v >= nv)
        return;
    double a = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 64)
        a += partials[(size_t)v * nblk + b];
    a = wave_sum(a);
    if (threadIdx.x == 0)
        out[v] = a;
}

// ------------------------------------------------------------------------------------------------
// dense GLM family, P columns (compile time, even, <= 128).  One wavefront instruction loads
// RPI = 64 / (P/2) consecutive rows, 16 B per lane (lane -> row h = lane / LPR, column pair
// c = lane % LPR): fully coalesced 1 KiB requests.  A tile of NT instructions (NT*RPI rows) is kept
// in registers; its per-lane partial dot products go through LDS so that lane r ends up with the
// whole dot product of row r (one exp per row instead of one per lane), then the per-row scalars
// come back through LDS for the transposed accumulation.  All orders are fixed.
struct GlmArgs
{
    const double *A;   // n x P row-major
    const double *y;   // n
    const double *sw;  // n or nullptr
    double *m;         // n: exp(a_i . x) at the point of the last EVAL pass (written by EVAL, read by JTJV)
    double *f;         // n: weighted residual at that point (written by EVAL)
    long long n;
    const double *xpt; // P: evaluation point (EVAL)
    const double *u;   // P: direction (JTJV)
    double *partials;  // [2 + 2P][nblk]: ssr, bad, g[P], d[P]
    int mode;
};

template <int P, int T>
__global__ __launch_bounds__(T) void glm_pass_kernel(GlmArgs a)
{
    constexpr int LPR = P / 2;      // lanes per row
    constexpr int RPI = 64 / LPR;   // rows per wave instruction
    constexpr int NT = (64 / RPI < 16) ? 64 / RPI : 16; // instructions per tile (at most 64 rows per tile)
    constexpr int ROWS = NT * RPI;  // rows per wave tile
    constexpr int NW = T / 64;
    constexpr int LD = LPR + 1;     // padded row length in LDS
    static_assert(64 % LPR == 0 && ROWS <= 64, "tile shape");
    __shared__ double lds_part[NW][ROWS * LD];
    __shared__ double lds_s1[NW][ROWS], lds_s2[NW][ROWS];
    __shared__ double lds_red[NW][2 * P + 2];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane / LPR, c = lane % LPR;
    const double v0 = (a.mode == LG_EVAL ? a.xpt : a.u)[2 * c], v1 = (a.mode == LG_EVAL ? a.xpt : a.u)[2 * c + 1];
    double g0 = 0.0, g1 = 0.0, d0 = 0.0, d1 = 0.0, ssr = 0.0;
    const long long ntiles = (a.n + ROWS - 1) / ROWS;
    const long long wstride = (long long)gridDim.x * NW;
    // uniform trip count per workgroup (tiles past the end are all-invalid rows) so that the LDS
    // hand-offs can use the workgroup barrier
    for (long long base = (long long)blockIdx.x * NW; base < ntiles; base += wstride)
    {
        const long long row0 = (base + wave) * ROWS;
        double2 av[NT];
#pragma unroll
        for (int k = 0; k < NT; ++k)
        {
            const long long row = row0 + k * RPI + h;
            av[k] = (row < a.n) ? *reinterpret_cast<const double2 *>(a.A + row * P + 2 * c) : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int k = 0; k < NT; ++k)
            lds_part[wave][(k * RPI + h) * LD + c] = av[k].x * v0 + av[k].y * v1;
        __syncthreads();
        double s1 = 0.0, s2 = 0.0;
        if (lane < ROWS)
        {
            const long long row = row0 + lane;
            double dot = 0.0;
#pragma unroll
            for (int j = 0; j < LPR; ++j)
                dot += lds_part[wave][lane * LD + j];
            if (row < a.n)
            {
                if (a.mode == LG_EVAL)
                {
                    const double mm = exp(dot);
                    const double w = a.sw ? a.sw[row] : 1.0;
                    const double ff = (isfinite(mm) ? mm - a.y[row] : INFINITY) * w;
                    a.m[row] = mm;
                    a.f[row] = ff;
                    ssr += ff * ff;
                    s1 = mm * ff;  // J^T f: row scalar m_i f_i
                    s2 = mm * mm;  // diag(J^T J): m_i^2
                }
                else
                {
                    const double mm = a.m[row];
                    const double t = mm * dot; // (J u)_i
                    ssr += t * t;
                    s1 = mm * t;
                }
            }
            lds_s1[wave][lane] = s1;
            lds_s2[wave][lane] = s2;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NT; ++k)
        {
            const double r1 = lds_s1[wave][k * RPI + h];
            g0 += av[k].x * r1;
            g1 += av[k].y * r1;
            if (a.mode == LG_EVAL)
            {
                const double r2 = lds_s2[wave][k * RPI + h];
                d0 += av[k].x * av[k].x * r2;
                d1 += av[k].y * av[k].y * r2;
            }
        }
        __syncthreads();
    }
    // combine the RPI row groups of the wave (lanes with equal c), fixed order
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1)
    {
        g0 += __shfl_xor(g0, off, 64);
        g1 += __shfl_xor(g1, off, 64);
        d0 += __shfl_xor(d0, off, 64);
        d1 += __shfl_xor(d1, off, 64);
    }
    ssr = wave_sum(ssr);
    if (lane < LPR)
    {
        lds_red[wave][2 + 2 * c] = g0;
        lds_red[wave][2 + 2 * c + 1] = g1;
        lds_red[wave][2 + P + 2 * c] = d0;
        lds_red[wave][2 + P + 2 * c + 1] = d1;
    }
    if (lane == 0)
    {
        lds_red[wave][0] = ssr;
        lds_red[wave][1] = 0.0;
    }
    __syncthreads();
    for (int v = threadIdx.x; v < 2 * P + 2; v += T)
    {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w)
            t += lds_red[w][v];
        a.partials[(size_t)v * gridDim.x + blockIdx.x] = t;
    }
}

// ------------------------------------------------------------------------------------------------
// K11: J^T J = A^T diag(m^2) A on the matrix cores for p = 16, 32, 48, 64 (north_star: "MFMA tall-skinny GEMM for
// J^T J only when p fills a 16-wide tile"; the reference densifies and calls dsyrk / spdgemm,
// src/nls_large.c:633,644-647).  v_mfma_f64_16x16x4_f64: D(16x16) += A'(16x4) B(4x16), operand layout probed on
// gfx950 (scripts/mfma_probe): lane l supplies A'[l%16][l/16] and B[l/16][l%16] and holds D[4r + l/16][l%16] in
// result register r.
// One wave instruction covers 4 consecutive rows: lane l (k = l/16, i = l%16) needs one entry of row k from each of
// the NB = P/16 column blocks.  Blocks are taken in pairs: a 16-byte load of J[row k][32 q + 2i, 32 q + 2i + 1]
// serves the blocks 2q and 2q+1 (every load instruction is dense over the wave: sixteen lanes read 256 contiguous
// bytes of a row), an odd last block is an 8-byte load of J[row k][16 (NB-1) + i].  So logical column (b, i) lives at
// physical column c(b, i) = 32 (b >> 1) + 2i + (b & 1) for the paired blocks and 16 b + i for an unpaired last one.
// The SAME register is the A' operand of block row b and the B operand of block column b, so the NB loaded values feed
// all NB (NB + 1) / 2 lower-triangle blocks: acc[ba][bb][r] at lane l is J^T J[c(ba, 4r + l/16)][c(bb, l%16)].
typedef double v4f64_t __attribute__((ext_vector_type(4)));

// Streaming structure: a wave owns the 4-row chunks c, c + S, c + 2S, ...; the loads of the NEXT TWO chunks are already
// in flight (registers) while the MFMAs of the current one issue -- at p = 64 one chunk is 640 MFMA cycles against
// ~900+ cycles of HBM latency, and the matrix pipe of a SIMD runs one chain per wave, so the loads have to be ahead
// of it; the hot loop has no bounds check (the ragged tail is one extra, masked chunk per wave).  LDS is only the
// epilogue's scratch (the waves add into it one after the other), so that it does not limit residency.
template <int P>
__device__ __forceinline__ int glm_jtj_phys(int b, int i)
{
    constexpr int NB = P / 16;
    return ((NB & 1) && b == NB - 1) ? 16 * b + i : 32 * (b >> 1) + 2 * i + (b & 1);
}

template <int P, int T>
__global__ __launch_bounds__(T, 4) void glm_jtj_mfma_kernel(const double *__restrict__ A, const double *__restrict__ m,
                                                            long long n, double *partials /* [P*P][gridDim.x] */)
{
    static_assert(P % 16 == 0 && P >= 16 && P <= 64, "16-wide tiles");
    constexpr int NB = P / 16, NQ = NB * (NB + 1) / 2, NPAIR = NB / 2, NW = T / 64;
    constexpr bool ODD = (NB & 1) != 0;
    __shared__ double lds_acc[NQ * 4 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = lane >> 4, i = lane & 15;
    v4f64_t acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
        acc[q] = (v4f64_t){0.0, 0.0, 0.0, 0.0};
    const long long nfull = n / 4; // chunks whose four rows all exist
    const long long S = (long long)gridDim.x * NW;
    const long long c0 = (long long)blockIdx.x * NW + wave;
    typedef double v2f64_t __attribute__((ext_vector_type(2)));
    struct Chunk
    {
        double v[NB];
        double mm;
    };
    auto fetch = [&](long long c) {
        Chunk t;
        const long long row = c * 4 + k;
        const double *src = A + row * P;
#pragma unroll
        for (int q = 0; q < NPAIR; ++q)
        {
            const v2f64_t d = __builtin_nontemporal_load(reinterpret_cast<const v2f64_t *>(src + 32 * q + 2 * i));
            t.v[2 * q] = d.x;
            t.v[2 * q + 1] = d.y;
        }
        if constexpr (ODD)
            t.v[NB - 1] = __builtin_nontemporal_load(src + 16 * (NB - 1) + i);
        t.mm = m[row];
        return t;
    };
    auto consume = [&](const Chunk &t) {
        double val[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b)
            val[b] = t.v[b] * t.mm;
        int q = 0;
#pragma unroll
        for (int ba = 0; ba < NB; ++ba)
#pragma unroll
            for (int bb = 0; bb <= ba; ++bb, ++q)
                acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(val[ba], val[bb], acc[q], 0, 0, 0);
    };
    Chunk zero;
#pragma unroll
    for (int b = 0; b < NB; ++b)
        zero.v[b] = 0.0;
    zero.mm = 0.0;
    Chunk b0 = c0 < nfull ? fetch(c0) : zero;
    Chunk b1 = c0 + S < nfull ? fetch(c0 + S) : zero;
    long long c = c0;
    for (; c + 2 * S < nfull; c += S)
    {
        const Chunk b2 = fetch(c + 2 * S);
        consume(b0);
        b0 = b1;
        b1 = b2;
    }
    if (c < nfull)
        consume(b0);
    if (c + S < nfull)
        consume(b1);
    // the ragged tail (n not a multiple of 4): one masked chunk, taken by the wave whose turn it would be
    if ((n & 3) && (nfull % S) == c0 % S && c0 <= nfull)
    {
        const long long row = nfull * 4 + k;
        Chunk t = zero;
        if (row < n)
        {
            const double *src = A + row * P;
#pragma unroll
            for (int q = 0; q < NPAIR; ++q)
            {
                t.v[2 * q] = src[32 * q + 2 * i];
                t.v[2 * q + 1] = src[32 * q + 2 * i + 1];
            }
            if constexpr (ODD)
                t.v[NB - 1] = src[16 * (NB - 1) + i];
            t.mm = m[row];
        }
        consume(t);
    }
    // waves of the workgroup -> one partial set (fixed order: wave 0, 1, ...), scattered to J^T J element order
    for (int w = 0; w < NW; ++w)
    {
        if (wave == w)
        {
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                {
                    const int e = (q * 4 + r) * 64 + lane;
                    lds_acc[e] = (w == 0) ? acc[q][r] : lds_acc[e] + acc[q][r];
                }
        }
        __syncthreads();
    }
    for (int e = threadIdx.x; e < NQ * 4 * 64; e += T)
    {
        const double t = lds_acc[e];
        const int q = e / 256, r = (e / 64) & 3, l = e & 63;
        int ba = 0, bb = 0, cnt = 0;
        for (int a = 0; a < NB; ++a)
            for (int b = 0; b <= a; ++b, ++cnt)
                if (cnt == q)
                {
                    ba = a;
                    bb = b;
                }
        const int gi = glm_jtj_phys<P>(ba, 4 * r + (l >> 4)), gj = glm_jtj_phys<P>(bb, l & 15);
        partials[((size_t)gi * P + gj) * gridDim.x + blockIdx.x] = t;
        if (ba != bb)
            partials[((size_t)gj * P + gi) * gridDim.x + blockIdx.x] = t; // mirror block
    }
}

} // namespace gslnls
