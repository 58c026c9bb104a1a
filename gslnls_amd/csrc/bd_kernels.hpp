// bd_kernels.hpp -- device kernels of the "big dense" path: gsl_nls() with more than 64 parameters, and gsl_nls() on an
// R function (any p), where the Jacobian is a MATRIX in HBM instead of rows recomputed in registers.
//
// The reference allocates an n x p workspace for any p (gsl_multifit_nlinear_alloc, src/nls.c:266) and fills it from the
// closure's result (gsl_df, src/nls.c:864-916) or by finite differences (src/fdjac.c:24-128); GSL's Cholesky solver then
// forms J^T J with dsyrk (multifit_nlinear/cholesky.c) and J^T f with dgemv (trust.c).  Here J lives column-major in HBM
// (n x p, exactly the layout C_nls hands back as `grad`, src/nls.c:718) and
//   bd_syrk_kernel     J^T J on the matrix cores: 64 x 64 blocks of the lower triangle, v_mfma_f64_16x16x4_f64, the two column
//                      panels of a block staged through LDS 64 rows at a time (the tile layout of wide_kernels.hpp);
//                      row slices in parallel, partial blocks summed in slice order by bd_syrk_reduce_kernel (no atomics:
//                      bit-identical run to run), which also mirrors the upper triangle for the device factorisation;
//   bd_gemv_t_kernel   J^T f, one workgroup per column (contiguous in memory);
//   bd_resid_kernel    f = sqrt(w) (m - y) with the reference's non-finite rule (src/nls.c:843-849) + sum of squares;
//   bd_weight_kernel   rows of an analytic Jacobian scaled by sqrt(w) (src/fdf.c:135-177) + non-finite check;
//   bd_fdcol_kernel    one column of a forward / central difference Jacobian (src/fdjac.c:81-128, :147-168);
//   bd_quad_kernel     rows of v^T (J^T J) v for the predicted reduction (GSL lm_preduction);
//   bd_cooks_kernel    hat values and Cook's distances of the robust multi-start second pass (src/nls_utils.c:88-150).
#pragma once
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "lm_core.hpp"

namespace gslnls
{

constexpr int BD_T = 256;
constexpr int BD_LD = 68;   // leading dimension of a staged tile (doubles): conflict-free column writes, 2 lanes per bank on reads
constexpr int BD_MAXG = 1024;

typedef double bd_v4f64 __attribute__((ext_vector_type(4)));

// fixed-shape workgroup sum (BD_T threads): wavefront xor butterflies, then the four wave sums in wave order
__device__ __forceinline__ double bd_block_sum(double v, double *red_s)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
    {
        const long long bits = __double_as_longlong(v);
        const int lo = __shfl_xor((int)(bits & 0xffffffffll), m, 64), hi = __shfl_xor((int)(bits >> 32), m, 64);
        v += __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0)
        red_s[wave] = v;
    __syncthreads();
    double s = red_s[0];
    for (int w = 1; w < BD_T / 64; ++w)
        s += red_s[w];
    return s;
}

// f_i = sqrt(w_i) (m_i - y_i), +Inf where the model value is not finite (src/nls.c:843-849); part[block] = sum f_i^2
__global__ __launch_bounds__(BD_T) void bd_resid_kernel(const double *fval, const double *y, const double *sw, long long n,
                                                        double *f, double *part)
{
    __shared__ double red_s[BD_T / 64];
    double s = 0.0;
    for (long long i = (long long)blockIdx.x * BD_T + threadIdx.x; i < n; i += (long long)gridDim.x * BD_T)
    {
        const double m = fval[i];
        double r = isfinite(m) ? m - y[i] : INFINITY;
        if (sw)
            r *= sw[i];
        f[i] = r;
        s += r * r;
    }
    s = bd_block_sum(s, red_s);
    if (threadIdx.x == 0 && part)
        part[blockIdx.x] = s;
}

// rows of J (n x p column-major) scaled by sqrt(w); part[block] = NaN when an entry of J is not finite, else 0
__global__ __launch_bounds__(BD_T) void bd_weight_kernel(double *J, const double *sw, long long n, int p, double *part)
{
    __shared__ double red_s[BD_T / 64];
    double bad = 0.0;
    const long long tot = n * (long long)p;
    for (long long e = (long long)blockIdx.x * BD_T + threadIdx.x; e < tot; e += (long long)gridDim.x * BD_T)
    {
        const double v = J[e];
        bad = fma(v, 0.0, bad);
        if (sw)
            J[e] = v * sw[e % n];
    }
    bad = bd_block_sum(bad, red_s);
    if (threadIdx.x == 0)
        part[blockIdx.x] = bad;
}

// column of a difference Jacobian: (fa - fb) * dinv (forward: fa = f(x + d e_j), fb = f(x); central: f(x +- d/2 e_j))
__global__ __launch_bounds__(BD_T) void bd_fdcol_kernel(const double *fa, const double *fb, double dinv, double *col, long long n)
{
    for (long long i = (long long)blockIdx.x * BD_T + threadIdx.x; i < n; i += (long long)gridDim.x * BD_T)
        col[i] = (fa[i] - fb[i]) * dinv;
}

// g_j = sum_i J[i][j] f_i: one workgroup per column
__global__ __launch_bounds__(BD_T) void bd_gemv_t_kernel(const double *J, const double *f, long long n, int p, double *g)
{
    __shared__ double red_s[BD_T / 64];
    const double *col = J + (size_t)blockIdx.x * n;
    double s = 0.0;
    for (long long i = threadIdx.x; i < n; i += BD_T)
        s += col[i] * f[i];
    s = bd_block_sum(s, red_s);
    if (threadIdx.x == 0)
        g[blockIdx.x] = s;
}

// u_i = sum_k J[i][k] v_k (row i of J v), for the second directional derivative by differences (src/fdfvv.c:35-77)
__global__ __launch_bounds__(BD_T) void bd_gemv_n_kernel(const double *J, const double *v, long long n, int p, double *u)
{
    for (long long i = (long long)blockIdx.x * BD_T + threadIdx.x; i < n; i += (long long)gridDim.x * BD_T)
    {
        double s = 0.0;
        for (int k = 0; k < p; ++k)
            s += J[(size_t)k * n + i] * v[k];
        u[i] = s;
    }
}

// fvv_i = (2 / h) ((f_i(x + h v) - f_i(x)) / h - (J v)_i)   (src/fdfvv.c:60-72)
__global__ __launch_bounds__(BD_T) void bd_fdfvv_kernel(const double *fp, const double *f, const double *u, double h, double *out,
                                                        long long n)
{
    const double hinv = 1.0 / h;
    for (long long i = (long long)blockIdx.x * BD_T + threadIdx.x; i < n; i += (long long)gridDim.x * BD_T)
        out[i] = (2.0 * hinv) * ((fp[i] - f[i]) * hinv - u[i]);
}

// analytic fvv from a closure: weighted, non-finite -> flag (part[block] NaN)
__global__ __launch_bounds__(BD_T) void bd_weight_vec_kernel(double *v, const double *sw, long long n, double *part)
{
    __shared__ double red_s[BD_T / 64];
    double bad = 0.0;
    for (long long i = (long long)blockIdx.x * BD_T + threadIdx.x; i < n; i += (long long)gridDim.x * BD_T)
    {
        const double r = v[i];
        bad = fma(r, 0.0, bad);
        if (sw)
            v[i] = r * sw[i];
    }
    bad = bd_block_sum(bad, red_s);
    if (threadIdx.x == 0)
        part[blockIdx.x] = bad;
}

// out[i] = (sum_j C[i][j] v_j) v_i: the rows of v^T C v (C p x p row-major, symmetric); the host adds them in index order
__global__ __launch_bounds__(BD_T) void bd_quad_kernel(const double *C, const double *v, int p, double *out)
{
    __shared__ double red_s[BD_T / 64];
    const int i = blockIdx.x;
    const double *row = C + (size_t)i * p;
    double s = 0.0;
    for (int j = threadIdx.x; j < p; j += BD_T)
        s += row[j] * v[j];
    s = bd_block_sum(s, red_s);
    if (threadIdx.x == 0)
        out[i] = s * v[i];
}

// hat values h_i = J_i (J^T J)^-1 J_i^T and Cook's distances D_i = e_i^2 / (p s^2) h_i / (1 - h_i)^2 from the resident
// residual and Jacobian (hat_values / cooks_d, src/nls_utils.c:88-150), any p: lane = row, (J^T J)^-1 read as broadcasts
// (every lane the same address) in the order of the reference's row sum
__global__ __launch_bounds__(BD_T) void bd_cooks_kernel(const double *resid, const double *J, long long n, int p, const double *Cinv,
                                                        double s2, double *d, unsigned long long *keys, double *hat)
{
    for (long long i = (long long)blockIdx.x * BD_T + threadIdx.x; i < n; i += (long long)gridDim.x * BD_T)
    {
        double h = 0.0;
        for (int j = 0; j < p; ++j)
        {
            double t = 0.0;
            for (int k = 0; k < p; ++k)
                t += J[i + (size_t)n * k] * Cinv[(size_t)k * p + j];
            h += t * J[i + (size_t)n * j];
        }
        const double e = resid[i];
        const double di = (e * e) / (p * s2) * (h / ((1 - h) * (1 - h)));
        if (d)
            d[i] = di;
        if (keys)
            keys[i] = (unsigned long long)__double_as_longlong(fabs(di));
        if (hat)
            hat[i] = h;
    }
}

// ---- J^T J ------------------------------------------------------------------------------------------------------------
// Block (I, Jb), I >= Jb, of the lower triangle in units of 64 columns; `slice` of the row tiles.  The four wavefronts of a
// workgroup share the two staged panels (64 rows x 64 columns each, tile[column][row]); wavefront w owns block row w of the
// 4 x 4 grid of 16 x 16 products: per 4-row chunk one A' operand (panel I, columns 16 w ..) and four B operands (panel
// Jb), four MFMAs.  Operand / result layout of v_mfma_f64_16x16x4_f64 as in wide_kernels.hpp: lane l supplies
// A'[l % 16][l / 16] and B[l / 16][l % 16] and holds D[4 r + l / 16][l % 16] in result register r.
// cpart: [slice][pair][64 x 64] partial blocks.
__global__ __launch_bounds__(BD_T, 2) void bd_syrk_kernel(const double *J, long long n, int p, int nslice, double *cpart)
{
    __shared__ double tile[2][64 * BD_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kk = lane >> 4, ii = lane & 15;
    // pair index -> (I, Jb), Jb <= I
    const int pair = blockIdx.x;
    int I = (int)((sqrt(8.0 * pair + 1.0) - 1.0) * 0.5);
    while (I * (I + 1) / 2 > pair)
        --I;
    while ((I + 1) * (I + 2) / 2 <= pair)
        ++I;
    const int Jb = pair - I * (I + 1) / 2;
    const int slice = blockIdx.y;
    const bool diagblk = I == Jb;
    bd_v4f64 acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
        acc[b] = (bd_v4f64){0.0, 0.0, 0.0, 0.0};
    const long long ntile = (n + 63) / 64;
    // Round 5: every load of a tile is issued before the first is waited for, and the next tile's loads are in flight while
    // the matrix instructions of the current one run.  (`x = cond ? load : 0` in a loop compiles to a branch and a wait per
    // load -- 32 serial round trips per tile; the kernel timeline of the matrix path showed 23 us for the one or two tiles
    // a workgroup has at n = 3000.)  Rows and columns that do not exist read element 0 and are discarded.
    double la[16], lb[16];
    auto fetch = [&](long long t) {
        const long long r = t * 64 + lane;
#pragma unroll
        for (int q = 0; q < 16; ++q)
        {
            const int c = wave + 4 * q, cI = I * 64 + c, cJ = Jb * 64 + c;
            la[q] = J[(t < ntile && r < n && cI < p) ? (size_t)cI * n + r : 0];
            lb[q] = J[(t < ntile && !diagblk && r < n && cJ < p) ? (size_t)cJ * n + r : 0];
        }
    };
    fetch(slice);
    for (long long t = slice; t < ntile; t += nslice)
    {
        const long long r = t * 64 + lane;
        __syncthreads(); // the previous tile has been consumed
        // stage: wavefront w holds columns w, w + 4, ... of each panel, lane = row (512 contiguous bytes per load instruction)
#pragma unroll
        for (int q = 0; q < 16; ++q)
        {
            const int c = wave + 4 * q, cI = I * 64 + c, cJ = Jb * 64 + c;
            tile[0][c * BD_LD + lane] = (r < n && cI < p) ? la[q] : 0.0;
            if (!diagblk)
                tile[1][c * BD_LD + lane] = (r < n && cJ < p) ? lb[q] : 0.0;
        }
        __syncthreads();
        fetch(t + nslice);
        const double *tA = tile[0], *tB = diagblk ? tile[0] : tile[1];
        // (the LDS reads of chunk c + 2 are requested before the matrix instructions of chunk c issue: "read, wait, four
        // MFMAs" per chunk left the pipe idle for an LDS round trip sixteen times per tile)
        constexpr int AH = 2;
        double va[AH + 1], vb[AH + 1][4];
#pragma unroll
        for (int c = 0; c < AH; ++c)
        {
            va[c] = tA[(wave * 16 + ii) * BD_LD + c * 4 + kk];
#pragma unroll
            for (int b = 0; b < 4; ++b)
                vb[c][b] = tB[(b * 16 + ii) * BD_LD + c * 4 + kk];
        }
#pragma unroll
        for (int c = 0; c < 16; ++c)
        {
            constexpr int R = AH + 1;
            if (c + AH < 16)
            {
                va[(c + AH) % R] = tA[(wave * 16 + ii) * BD_LD + (c + AH) * 4 + kk];
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    vb[(c + AH) % R][b] = tB[(b * 16 + ii) * BD_LD + (c + AH) * 4 + kk];
            }
#pragma unroll
            for (int b = 0; b < 4; ++b)
                acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(va[c % R], vb[c % R][b], acc[b], 0, 0, 0);
        }
    }
    // block row `wave`: element (16 wave + 4 r + kk, 16 b + ii)
    double *out = cpart + ((size_t)slice * gridDim.x + pair) * 4096;
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            out[(wave * 16 + 4 * r + kk) * 64 + b * 16 + ii] = acc[b][r];
}

// C[i][j] = sum over the slices (in slice order) of the partial blocks; both triangles are written (C p x p row-major).
// Workgroup (pair, y of ny): 256 elements of the 64 x 64 block per trip, one per thread.
__device__ __forceinline__ void bd_syrk_reduce_body(const double *cpart, int p, int npair, int nslice, double *C, int pair, int y, int ny)
{
    int I = (int)((sqrt(8.0 * pair + 1.0) - 1.0) * 0.5);
    while (I * (I + 1) / 2 > pair)
        --I;
    while ((I + 1) * (I + 2) / 2 <= pair)
        ++I;
    const int Jb = pair - I * (I + 1) / 2;
    for (int e = y * BD_T + threadIdx.x; e < 4096; e += ny * BD_T)
    {
        const int r = e >> 6, c = e & 63;
        const int gi = I * 64 + r, gj = Jb * 64 + c;
        if (gi >= p || gj >= p)
            continue;
        double s = cpart[(size_t)pair * 4096 + e];
        for (int sl = 1; sl < nslice; ++sl)
            s += cpart[((size_t)sl * npair + pair) * 4096 + e];
        if (I != Jb || gj <= gi)
        {
            C[(size_t)gi * p + gj] = s;
            C[(size_t)gj * p + gi] = s;
        }
    }
}
__global__ __launch_bounds__(BD_T) void bd_syrk_reduce_kernel(const double *cpart, int p, int npair, int nslice, double *C)
{
    bd_syrk_reduce_body(cpart, p, npair, nslice, C, blockIdx.x, blockIdx.y, gridDim.y);
}

// ---- round 5: J^T J in 128-column blocks ------------------------------------------------------------------------------------------
// bd_syrk_kernel moves 64 + 64 columns per 64 x 64 block of products: at p = 501 it pulls 3.7 TB/s out of L2 / Infinity Cache
// for 25 TFLOP/s (n p^2) -- bandwidth, not the matrix pipe, is its limit.  Here a workgroup owns a 128 x 128 block (I2, J2),
// J2 <= I2: per 32-row stage it stages 128 + 128 columns (the same 64 KB as a 64-row stage of the kernel above) for four
// times the products of a 64 x 64 block's stage -- twice the flops per byte.  Wavefront w owns rows 32 w .. 32 w + 31 of the
// block: two A' operands and eight B operands per 4-row chunk, sixteen accumulators.  Two workgroups per CU (LDS 72 KB each):
// one stages while the other multiplies.  Used from p = 384 and n = 2048 on (bd_syrk_geom); partial blocks summed in slice order by
// bd_syrk128_reduce_kernel -- no atomics, bit-identical run to run (not to the 64-column kernel: other slices, other sums).
constexpr int BD_LD32 = 36; // leading dimension of a 32-row stage (doubles): same bank pattern as BD_LD (36 = 68 = 4 mod 32)
__global__ __launch_bounds__(BD_T, 2) void bd_syrk128_kernel(const double *J, long long n, int p, int nslice, double *cpart)
{
    __shared__ double tile[2][128 * BD_LD32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kk = lane >> 4, ii = lane & 15;
    const int pair = blockIdx.x;
    int I = (int)((sqrt(8.0 * pair + 1.0) - 1.0) * 0.5);
    while (I * (I + 1) / 2 > pair)
        --I;
    while ((I + 1) * (I + 2) / 2 <= pair)
        ++I;
    const int Jb = pair - I * (I + 1) / 2;
    const int slice = blockIdx.y;
    const bool diagblk = I == Jb;
    bd_v4f64 acc[2][8];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b)
            acc[a][b] = (bd_v4f64){0.0, 0.0, 0.0, 0.0};
    const long long nstage = (n + 31) / 32;
    const int row = tid & 31, cg = tid >> 5; // this thread stages row `row` of columns cg, cg + 8, ... of each side
    for (long long t = slice; t < nstage; t += nslice)
    {
        const long long r = t * 32 + row;
        // every load of the stage before the first wait (element 0 where a row or a column does not exist)
        double la[16], lb[16];
#pragma unroll
        for (int q = 0; q < 16; ++q)
        {
            const int c = cg + 8 * q, cI = I * 128 + c, cJ = Jb * 128 + c;
            la[q] = J[(r < n && cI < p) ? (size_t)cI * n + r : 0];
            lb[q] = J[(!diagblk && r < n && cJ < p) ? (size_t)cJ * n + r : 0];
        }
        __syncthreads(); // the previous stage has been consumed
#pragma unroll
        for (int q = 0; q < 16; ++q)
        {
            const int c = cg + 8 * q, cI = I * 128 + c, cJ = Jb * 128 + c;
            tile[0][c * BD_LD32 + row] = (r < n && cI < p) ? la[q] : 0.0;
            if (!diagblk)
                tile[1][c * BD_LD32 + row] = (r < n && cJ < p) ? lb[q] : 0.0;
        }
        __syncthreads();
        const double *tA = tile[0], *tB = diagblk ? tile[0] : tile[1];
#pragma unroll 2
        for (int c = 0; c < 8; ++c)
        {
            double va[2], vb[8];
#pragma unroll
            for (int a = 0; a < 2; ++a)
                va[a] = tA[((2 * wave + a) * 16 + ii) * BD_LD32 + c * 4 + kk];
#pragma unroll
            for (int b = 0; b < 8; ++b)
                vb[b] = tB[(b * 16 + ii) * BD_LD32 + c * 4 + kk];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 8; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(va[a], vb[b], acc[a][b], 0, 0, 0);
        }
    }
    // element (32 wave + 16 a + 4 r + kk, 16 b + ii) of the 128 x 128 block
    double *out = cpart + ((size_t)slice * gridDim.x + pair) * 16384;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                out[(wave * 32 + a * 16 + 4 * r + kk) * 128 + b * 16 + ii] = acc[a][b][r];
}

// C[i][j] = sum over the slices (in slice order) of the 128 x 128 partial blocks; both triangles are written
__device__ __forceinline__ void bd_syrk128_reduce_body(const double *cpart, int p, int npair, int nslice, double *C, int pair, int y, int ny)
{
    int I = (int)((sqrt(8.0 * pair + 1.0) - 1.0) * 0.5);
    while (I * (I + 1) / 2 > pair)
        --I;
    while ((I + 1) * (I + 2) / 2 <= pair)
        ++I;
    const int Jb = pair - I * (I + 1) / 2;
    for (int e = y * BD_T + threadIdx.x; e < 16384; e += ny * BD_T)
    {
        const int r = e >> 7, c = e & 127;
        const int gi = I * 128 + r, gj = Jb * 128 + c;
        if (gi >= p || gj >= p)
            continue;
        if (I == Jb && gj > gi)
            continue;
        double s = cpart[(size_t)pair * 16384 + e];
        for (int sl = 1; sl < nslice; ++sl)
            s += cpart[((size_t)sl * npair + pair) * 16384 + e];
        C[(size_t)gi * p + gj] = s;
        C[(size_t)gj * p + gi] = s;
    }
}
__global__ __launch_bounds__(BD_T) void bd_syrk128_reduce_kernel(const double *cpart, int p, int npair, int nslice, double *C)
{
    bd_syrk128_reduce_body(cpart, p, npair, nslice, C, blockIdx.x, blockIdx.y, gridDim.y);
}

// which of the two kernels forms J^T J of an n x p matrix, with how many workgroups, and the scratch it needs
struct BdSyrkGeom
{
    int wide = 0;      // 1: 128-column blocks (bd_syrk128_kernel), 0: 64-column blocks (bd_syrk_kernel)
    int npair = 0, nslice = 1;
    size_t scratch = 0; // doubles of partial blocks
};
inline BdSyrkGeom bd_syrk_geom(long long n, int p)
{
    BdSyrkGeom g;
    const char *e = getenv("GSLNLS_BD_SYRK64"); // developer switch: the 64-column kernel at every p
    // (measured, gpurun_out r05: n = 5000, p = 198: 33 us against 46; n = 501, p = 500: 17 against 22 -- small problems want
    // the many small blocks; n = 20000, p = 501: 197 against 178; p = 1000 / 2000: 1.97 / 3.80 ms against 1.27 / 2.15)
    g.wide = (p >= 384 && n >= 2048 && !(e && atoi(e) != 0)) ? 1 : 0;
    const int bw = g.wide ? 128 : 64, rows = g.wide ? 32 : 64;
    const int npanel = (p + bw - 1) / bw;
    g.npair = npanel * (npanel + 1) / 2;
    const long long nt = (n + rows - 1) / rows;
    // npair x slices workgroups in ONE round of the chip's 512 slots (two per CU)
    long long sl = 512 / g.npair;
    sl = sl > nt ? nt : sl;
    const long long cap = g.wide ? 48 : 32;
    sl = sl > cap ? cap : (sl < 1 ? 1 : sl);
    g.nslice = (int)sl;
    g.scratch = (size_t)g.nslice * g.npair * (g.wide ? 16384 : 4096);
    return g;
}
inline void bd_syrk_launch(const double *d_J, long long n, int p, double *d_C, double *d_cpart, const BdSyrkGeom &g, hipStream_t st)
{
    if (g.wide)
    {
        hipLaunchKernelGGL(bd_syrk128_kernel, dim3(g.npair, g.nslice), dim3(BD_T), 0, st, d_J, n, p, g.nslice, d_cpart);
        hipLaunchKernelGGL(bd_syrk128_reduce_kernel, dim3(g.npair, 64), dim3(BD_T), 0, st, d_cpart, p, g.npair, g.nslice, d_C);
    }
    else
    {
        hipLaunchKernelGGL(bd_syrk_kernel, dim3(g.npair, g.nslice), dim3(BD_T), 0, st, d_J, n, p, g.nslice, d_cpart);
        hipLaunchKernelGGL(bd_syrk_reduce_kernel, dim3(g.npair, 16), dim3(BD_T), 0, st, d_cpart, p, g.npair, g.nslice, d_C);
    }
}

// the reduction of the partial blocks and J^T f in ONE launch (round 5: they do not depend on each other -- the first
// `nred` workgroups are the reduction's, the other p are bd_gemv_t_kernel's, each with its own kernel's arithmetic)
__global__ __launch_bounds__(BD_T) void bd_reduce_gemv_kernel(const double *cpart, int p, int npair, int nslice, int wide, double *C, int ny, int nred,
                                                              const double *J, const double *f, long long n, double *g)
{
    const int b = blockIdx.x;
    if (b < nred)
    {
        if (wide)
            bd_syrk128_reduce_body(cpart, p, npair, nslice, C, b / ny, b % ny, ny);
        else
            bd_syrk_reduce_body(cpart, p, npair, nslice, C, b / ny, b % ny, ny);
        return;
    }
    __shared__ double red_s[BD_T / 64];
    const int k = b - nred;
    const double *col = J + (size_t)k * n;
    double s = 0.0;
    for (long long i = threadIdx.x; i < n; i += BD_T)
        s += col[i] * f[i];
    s = bd_block_sum(s, red_s);
    if (threadIdx.x == 0)
        g[k] = s;
}
// J^T J (into d_C) and g = J^T f in two launches: the products, then [reduction | J^T f]
inline void bd_syrk_gemv_launch(const double *d_J, long long n, int p, double *d_C, double *d_cpart, const BdSyrkGeom &g, const double *d_f,
                                double *d_g, hipStream_t st)
{
    if (g.wide)
        hipLaunchKernelGGL(bd_syrk128_kernel, dim3(g.npair, g.nslice), dim3(BD_T), 0, st, d_J, n, p, g.nslice, d_cpart);
    else
        hipLaunchKernelGGL(bd_syrk_kernel, dim3(g.npair, g.nslice), dim3(BD_T), 0, st, d_J, n, p, g.nslice, d_cpart);
    const int ny = g.wide ? 64 : 16, nred = g.npair * ny;
    hipLaunchKernelGGL(bd_reduce_gemv_kernel, dim3(nred + p), dim3(BD_T), 0, st, d_cpart, p, g.npair, g.nslice, g.wide, d_C, ny, nred, d_J, d_f, n,
                       d_g);
}

// ---- round 5: the covariance and the solver-routing diagnostic of a fit's end, on the device -------------------------------
// (until then host code, cubic in p: at p = 501 the two took 44 of a call's 76 ms, the LM loop itself 6)
//
// wavefront sum by xor butterflies (every lane ends with the total)
__device__ __forceinline__ double bd_wave_sum(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
    {
        const long long bits = __double_as_longlong(v);
        const int lo = __shfl_xor((int)(bits & 0xffffffffll), m, 64), hi = __shfl_xor((int)(bits >> 32), m, 64);
        v += __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
    }
    return v;
}
__device__ __forceinline__ double bd_lane_bcast(double v, int src)
{
    const long long bits = __double_as_longlong(v);
    const int lo = __shfl((int)(bits & 0xffffffffll), src, 64), hi = __shfl((int)(bits >> 32), src, 64);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// Column j of X = L^-1 (L: the factor the damped solve's kernels leave, p x p row-major lower triangle, dinv[c] = 1 / L_cc):
// workgroup j solves L x = e_j by blocked forward substitution -- 64 rows at a time, the part of a row in front of the
// block as a coalesced dot product with the x found so far (16 rows per wavefront), the block's own 64 x 64 triangle from
// LDS by the first wavefront (lane = row, x_c broadcast, 64 steps) -- and writes x below the zeros of rows < j into
// X[. + p j] (column-major p x p: the operand J of bd_syrk_kernel, so that (J^T J)^-1 = X^T X comes from the matrix cores).
// Dynamic LDS: 64 ceil((p - j) / 64) + 64 * 65 + 64 doubles.
constexpr int BD_TI_LD = 65;
__global__ __launch_bounds__(BD_T) void bd_trinv_kernel(const double *Lf, const double *dinv, int p, double *X)
{
    extern __shared__ double bd_ti_lds[];
    const int j = blockIdx.x, m = p - j, nblk = (m + 63) / 64;
    double *xs = bd_ti_lds, *D = xs + 64 * nblk, *ts = D + 64 * BD_TI_LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int b = 0; b < nblk; ++b)
    {
        const int r0 = j + 64 * b; // first row and column of this block
        for (int e = tid; e < 4096; e += BD_T)
        {
            const int r = e >> 6, c = e & 63;
            const bool ok = r0 + r < p && c <= r;
            const double v = Lf[ok ? (size_t)(r0 + r) * p + r0 + c : 0];
            D[r * BD_TI_LD + c] = ok ? v : 0.0;
        }
        // t_r = (e_j)_r - sum_{k < 64 b} L[r0 + r][j + k] x_k, four rows of a wavefront in flight
        for (int rr = wave * 16; rr < wave * 16 + 16; rr += 4)
        {
            double s[4] = {0.0, 0.0, 0.0, 0.0};
            for (int k = lane; k < 64 * b; k += 64)
            {
                const double xk = xs[k];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                {
                    const bool ok = r0 + rr + u < p;
                    const double l = Lf[ok ? (size_t)(r0 + rr + u) * p + j + k : 0];
                    s[u] += (ok ? l : 0.0) * xk;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
            {
                const double tot = bd_wave_sum(s[u]);
                if (lane == 0)
                    ts[rr + u] = ((b == 0 && rr + u == 0) ? 1.0 : 0.0) - tot;
            }
        }
        __syncthreads();
        if (wave == 0)
        {
            double t = ts[lane], xv = 0.0;
            const double dv = r0 + lane < p ? dinv[r0 + lane] : 0.0;
            for (int c = 0; c < 64; ++c)
            {
                const double xc = bd_lane_bcast(t, c) * bd_lane_bcast(dv, c);
                xv = lane == c ? xc : xv;
                t -= D[lane * BD_TI_LD + c] * xc; // (zero above the diagonal; lane c's own t is not read again)
            }
            xs[64 * b + lane] = xv;
        }
        __syncthreads();
    }
    for (int i = tid; i < p; i += BD_T)
        X[(size_t)p * j + i] = i < j ? 0.0 : xs[i - j];
}

// One step of a power iteration on S M S: w_out = S M S (w_in / |w_in|), *nrm_out = |w_in| (the norm the host loop of
// bd_scaled_cond takes after its product).  mode 0: M = A = J^T J, s_i = 1 / sqrt(A_ii) -- the column-scaled normal matrix
// C; mode 1: M = A^-1 (the covariance), s_i = sqrt(A_ii) -- C^-1 = S^-1 A^-1 S^-1, whose largest eigenvalue is 1 /
// lambda_min(C).  Every workgroup forms the norm and the scaled vector itself (same sums in the same order: same bits),
// then one row per wavefront.  Dynamic LDS: p doubles.
__global__ __launch_bounds__(BD_T) void bd_power_kernel(const double *M, const double *A, int p, int mode, const double *w_in, double *w_out,
                                                        double *nrm_out)
{
    extern __shared__ double bd_pw_lds[];
    __shared__ double red_s[BD_T / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double q = 0.0;
    for (int k = tid; k < p; k += BD_T)
        q += w_in[k] * w_in[k];
    const double nrm = sqrt(bd_block_sum(q, red_s));
    for (int k = tid; k < p; k += BD_T)
    {
        const double d = A[(size_t)k * p + k];
        bd_pw_lds[k] = (mode ? sqrt(d) : 1.0 / sqrt(d)) * (w_in[k] / nrm);
    }
    __syncthreads();
    if (blockIdx.x == 0 && tid == 0)
        *nrm_out = nrm;
    const int i = blockIdx.x * (BD_T / 64) + wave;
    if (i >= p)
        return;
    const double *row = M + (size_t)i * p;
    double s = 0.0;
    for (int k = lane; k < p; k += 64)
        s += row[k] * bd_pw_lds[k];
    s = bd_wave_sum(s);
    if (lane == 0)
    {
        const double d = A[(size_t)i * p + i];
        w_out[i] = (mode ? sqrt(d) : 1.0 / sqrt(d)) * s;
    }
}

// the two start vectors of bd_scaled_cond: (1, 1, ...) / sqrt(p) and (+1, -1, ...) / sqrt(p)
__global__ void bd_power_start_kernel(int p, double *v0, double *v1)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < p)
    {
        const double u = 1.0 / sqrt((double)p);
        v0[k] = u;
        v1[k] = (k & 1) ? -u : u;
    }
}

} // namespace gslnls
