// mchol_device.hip -- the damped normal equations of gsl_nls_large(algorithm = "lm") solved on the device.
//
// (J^T J + mu D^2) dx = -g by GSL's pivoted modified Cholesky (gsl_linalg_mcholesky, Gill-Murray-Wright: the same
// algorithm as lg_mchol_solve() in large_host.hpp and lm_solve<P> in lm_core.hpp, which see for the reference lines) for
// p from a hundred to a few thousand, where the host's column-at-a-time factorisation is most of the run (README Example
// 4, p = 500: 247 solves x 3.96 ms = 977 of 1060 ms).
//
// The factorisation is a chain of p dependent pivot steps; what a step costs on the device is latency (a block-wide
// reduction, one read of the pivot row), not arithmetic.  So:
//   * nothing is interchanged: thread r owns row r of the matrix for the whole factorisation, the reference's permutation
//     is a position per row (pos[r]; the tie rule of the pivot search -- the first position wins -- reads it) and
//     ord[j] = the row eliminated at step j;
//   * steps are taken in panels of NB by ONE workgroup (mchol_panel_kernel): the matrix in global memory is only read --
//     the column of the pivot is the row A[q][.] as it stood when the panel began (one coalesced read, issued the moment
//     q is known) minus the contributions of the panel's earlier steps, whose columns sit in LDS (left-looking); the
//     diagonal, the right-hand side and the positions live in LDS for the whole panel;
//   * after a panel the whole grid applies its NB rank-one updates to the matrix at once (mchol_trail_kernel), rows
//     and columns already eliminated carry zeros and are not touched;
//   * the forward substitution rides along with the steps; the multipliers go to global memory step by step (Lg[j][r]),
//     and the back substitution (mchol_backsub_kernel) walks them in reverse with one block-wide sum per step.
// Against the host routine the sums are the same, their association is not (a panel's contributions to a column are
// added up before they are subtracted, the sums of the back substitution are trees, v_i / alpha multiplies v_k where
// interchanged rows would have it the other way round): parity with the oracle is to round-off, tests/test_gpu_large.py.
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <mutex>
#include "../../include/gslnls_core.h"
#include "dense_host.hpp"
#include "large_host.hpp"
#include "wide_core.hpp"

namespace gslnls
{

constexpr int MC_T = 1024;            // threads of the panel workgroup (16 wavefronts)
constexpr int MC_W = MC_T / 64;
constexpr int MC_NB_MAX = 32;         // pivot steps per panel
constexpr int MC_PMAX = 4096;
constexpr int MC_RPT = 4;             // rows per thread at most (the block is sized to p: MC_RPT x blockDim.x >= p)
constexpr int MC_LDS_BYTES = 150 * 1024;

struct MCholArgs
{
    double *A;     // p x p, symmetric, row-major; updated panel by panel
    double *Lg;    // p x p: Lg[j][r] = multiplier of row r at step j (0 where the row took no part)
    double *Cg;    // NB x p: the columns of the panel that just ended, for the trailing update
    double *ainvg; // NB
    double *dcur, *b, *dinv; // p: current diagonal, right-hand side (forward substitution applied), 1 / alpha of the row's step
    int *pos, *ord;          // p: position of a row / row eliminated at a step
    double *scal;            // [0] sqrt(beta)
    int p, kb, nb;
};

// candidate of a wavefront / of the block for the pivot: largest value, then smallest position
struct MCholCand
{
    double val, b;
    int pos, row;
};

__device__ __forceinline__ bool mchol_better(double v, int ps, double bv, int bp)
{
    return v > bv || (v == bv && ps < bp);
}

// wavefront-wide winner among the first 16 R lanes' candidates (value, then smallest position); every lane gets its lane index
template <int R>
__device__ __forceinline__ int mchol_wave_winner(const MCholCand &c)
{
    const double wm = wide_wave_max_rows<R>(c.val);
    unsigned long long hit = __builtin_amdgcn_ballot_w64(c.val == wm);
    int win = hit ? (int)__builtin_ctzll(hit) : 0;
    if (wm >= 0.0 && (hit & (hit - 1)))
    {
        int best = 0x7fffffff;
        while (hit)
        {
            const int l = (int)__builtin_ctzll(hit);
            hit &= hit - 1;
            const int pl = __builtin_amdgcn_readlane(c.pos, l);
            if (pl < best)
            {
                best = pl;
                win = l;
            }
        }
    }
    return win;
}

// block-wide: the (value, position)-best of every thread's candidate; all threads get the winner.  Wavefront winners go to
// LDS (rec: MC_W records + 1), the first wavefront picks among them, everybody reads the result: two barriers, and no
// thread scans sixteen records
__device__ __forceinline__ MCholCand mchol_block_best(MCholCand c, MCholCand *rec, int tid, int nwaves)
{
    const int lane = tid & 63, wave = tid >> 6;
    const int win = mchol_wave_winner<4>(c);
    if (lane == win)
        rec[wave] = c;
    __syncthreads();
    if (wave == 0)
    {
        MCholCand o;
        o.val = -2.0;
        o.pos = 0x7fffffff;
        o.row = 0;
        o.b = 0.0;
        if (lane < nwaves)
            o = rec[lane];
        const int w2 = mchol_wave_winner<1>(o);
        if (lane == w2)
            rec[MC_W] = o;
    }
    __syncthreads();
    return rec[MC_W];
}

// gamma = max |a_ii| and xi = max |a_ij| (i != j) as bit patterns in scal[1], scal[2] (non-negative doubles order like
// unsigned integers: atomicMax; NaNs are skipped as fmax skips them), and the per-row state
// (src != nullptr: the matrix is J^T J as it sits on the device, A = src + mu diag(dmp)^2 is formed on the way)
__global__ __launch_bounds__(256) void mchol_init_kernel(MCholArgs a, const double *rhs, const double *src, const double *dmp,
                                                         double mu)
{
    const int p = a.p;
    const size_t pp = (size_t)p * p, stride = (size_t)gridDim.x * 256;
    double gm = 0.0, xm = 0.0;
#pragma unroll 4
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < pp; e += stride)
    {
        const size_t i = e / p, k = e - i * p;
        if (src)
        {
            double x = src[e];
            if (i == k)
                x = __dadd_rn(x, __dmul_rn(__dmul_rn(mu, dmp[i]), dmp[i])); // (rounded like the host's A[i][i] += mu d d)
            a.A[e] = x;
        }
        const double v = fabs(a.A[e]);
        if (i == k)
            gm = fmax(gm, v);
        else
            xm = fmax(xm, v);
    }
    gm = wide_wave_max_rows<4>(gm);
    xm = wide_wave_max_rows<4>(xm);
    if ((threadIdx.x & 63) == 0)
    {
        atomicMax(reinterpret_cast<unsigned long long *>(a.scal + 1), (unsigned long long)__double_as_longlong(gm));
        atomicMax(reinterpret_cast<unsigned long long *>(a.scal + 2), (unsigned long long)__double_as_longlong(xm));
    }
    for (size_t r = (size_t)blockIdx.x * 256 + threadIdx.x; r < (size_t)p; r += stride)
    {
        // (another thread may own a.A[r][r])
        a.dcur[r] = src ? __dadd_rn(src[r * p + r], __dmul_rn(__dmul_rn(mu, dmp[r]), dmp[r])) : a.A[r * p + r];
        a.b[r] = rhs[r];
        a.dinv[r] = 0.0;
        a.pos[r] = (int)r;
        a.ord[r] = 0;
    }
}

// steps kb .. kb + nb - 1.  Dynamic LDS: Cp[nb][p] | ainv[NB_MAX] | wq[NB_MAX] | info | records.  The diagonal, the
// right-hand side and the position of a row live in the registers of the thread that owns it (thread tid: rows tid,
// tid + T, ...) for the whole panel.
__global__ __launch_bounds__(MC_T) void mchol_panel_kernel(MCholArgs a)
{
    extern __shared__ double lds[];
    const int tid = threadIdx.x, p = a.p, kb = a.kb, nb = a.nb;
    double *Cp = lds;
    double *ainv_s = Cp + (size_t)nb * p;
    double *wq = ainv_s + MC_NB_MAX;  // wq[k] = c_qk / alpha_k of the pivot row, for the left-looking sum of this step
    double *info = wq + MC_NB_MAX;    // the pivot of this step as the first wavefront publishes it (8 doubles)
    MCholCand *rec = reinterpret_cast<MCholCand *>(info + 8);
    // (rec: MC_W wavefront winners)
    __shared__ int s_nanq[2];
    __shared__ double s_nanb[2];
    const int T = blockDim.x, nwaves = T >> 6;
    double dreg[MC_RPT], breg[MC_RPT];
    int preg[MC_RPT];
#pragma unroll
    for (int u = 0; u < MC_RPT; ++u)
    {
        const int r = tid + u * T;
        dreg[u] = r < p ? a.dcur[r] : 0.0;
        breg[u] = r < p ? a.b[r] : 0.0;
        preg[u] = r < p ? a.pos[r] : -1;
    }
    if (tid == 0)
        s_nanq[0] = s_nanq[1] = -1;
    double betas;
    {
        const double gamma = a.scal[1], xi = a.scal[2];
        const double beta = (p == 1) ? fmax(fmax(gamma, xi), DBL_EPSILON)
                                     : fmax(fmax(gamma, xi / sqrt((double)p * p - 1.0)), DBL_EPSILON);
        betas = sqrt(beta);
    }
    const double binv = 1.0 / betas;
    __syncthreads();
    for (int t = 0; t < nb; ++t)
    {
        const int j = kb + t;
        if (tid == 0)
            s_nanq[(t + 1) & 1] = -1; // (last read two barriers ago, next written after this step's barriers)
        // ---- pivot: the first position holding the largest |diagonal| among positions >= j (`if (d > maxd)` of the
        // sequential scan: the first wins ties, NaNs never win -- unless one sits at position j, where the scan starts)
        MCholCand c;
        c.val = -1.0;
        c.pos = 0x7fffffff;
        c.row = 0;
        c.b = 0.0;
#pragma unroll
        for (int u = 0; u < MC_RPT; ++u)
        {
            const int ps = preg[u];
            if (ps >= j)
            {
                const double d = dreg[u];
                const double ad = fmax(fabs(d), 0.0); // NaN -> 0
                if (ps == j && d != d)
                {
                    s_nanq[t & 1] = tid + u * T;
                    s_nanb[t & 1] = breg[u];
                }
                if (mchol_better(ad, ps, c.val, c.pos))
                {
                    c.val = ad;
                    c.pos = ps;
                    c.row = tid + u * T;
                    c.b = breg[u];
                }
            }
        }
        // wavefront winners -> LDS; the first wavefront picks the block's, applies the NaN rule, starts 1 / max(eps, |d_qq|)
        // and lays out the pivot row's share of the left-looking sums (wq[k] = c_qk / alpha_k) while the others wait at
        // the second barrier anyway
        {
            const int lane = tid & 63, wave = tid >> 6;
            const int win = mchol_wave_winner<4>(c);
            if (lane == win)
                rec[wave] = c;
            __syncthreads();
            if (wave == 0)
            {
                MCholCand o;
                o.val = -2.0;
                o.pos = 0x7fffffff;
                o.row = 0;
                o.b = 0.0;
                if (lane < nwaves)
                    o = rec[lane];
                const int w2 = mchol_wave_winner<1>(o);
                int q0 = __builtin_amdgcn_readlane(o.row, w2), posq0 = __builtin_amdgcn_readlane(o.pos, w2);
                double dq0 = wide_bcast(o.val, w2), bq0 = wide_bcast(o.b, w2);
                const int nq = s_nanq[t & 1];
                if (nq >= 0)
                {
                    q0 = nq;
                    posq0 = j;
                    dq0 = __longlong_as_double(0x7ff8000000000000ll);
                    bq0 = s_nanb[t & 1];
                }
                const double a00 = fmax(DBL_EPSILON, dq0);
                if (lane < t)
                    wq[lane] = Cp[(size_t)lane * p + q0] * ainv_s[lane];
                if (lane == 0)
                {
                    info[0] = (double)q0;
                    info[1] = (double)posq0;
                    info[2] = bq0;
                    info[3] = a00;
                    info[4] = 1.0 / a00;
                }
            }
            __syncthreads();
        }
        const int q = (int)info[0], posq = (int)info[1];
        const double bq = info[2], a0 = info[3], ainv0 = info[4];
        // ---- the pivot's column: A[q][.] as the panel found it, minus the panel's earlier steps
        double cr[MC_RPT];
        bool raise = false;
#pragma unroll
        for (int u = 0; u < MC_RPT; ++u)
        {
            const int r = tid + u * T;
            cr[u] = 0.0;
            if (r < p)
            {
                int ps = preg[u];
                // rows at position j and q trade positions (nothing moves)
                if (ps == j)
                    ps = posq;
                if (r == q)
                    ps = j;
                preg[u] = ps;
                if (ps > j)
                {
                    const double arq = a.A[(size_t)q * p + r];
                    double s = 0.0;
                    int k = 0;
                    for (; k + 3 < t; k += 4)
                    {
                        // (the reads of a group ahead of its arithmetic; the sum keeps its order)
                        const double c0 = Cp[(size_t)k * p + r], c1 = Cp[(size_t)(k + 1) * p + r], c2 = Cp[(size_t)(k + 2) * p + r],
                                     c3 = Cp[(size_t)(k + 3) * p + r];
                        const double w0 = wq[k], w1 = wq[k + 1], w2 = wq[k + 2], w3 = wq[k + 3];
                        s += c0 * w0;
                        s += c1 * w1;
                        s += c2 * w2;
                        s += c3 * w3;
                    }
                    for (; k < t; ++k)
                        s += Cp[(size_t)k * p + r] * wq[k];
                    cr[u] = arq - s;
                    const double wv = fabs(cr[u]) * binv;
                    raise = raise || (wv * wv > a0 * 0.9999999999999);
                }
            }
        }
        // ---- alpha = max(eps, |d_qq|, theta^2 / beta), theta = max |c_r|: theta only matters when it raises alpha, and
        // "some row raises it" is the same condition as "the maximum raises it" (rounding is monotone)
        double alpha = a0;
        if (__syncthreads_or(raise ? 1 : 0))
        {
            MCholCand m;
            m.val = 0.0;
            m.pos = tid;
            m.row = 0;
            m.b = 0.0;
#pragma unroll
            for (int u = 0; u < MC_RPT; ++u)
                m.val = fmax(m.val, fmax(fabs(cr[u]), 0.0));
            const MCholCand th = mchol_block_best(m, rec, tid, nwaves);
            const double uu = th.val / betas;
            alpha = fmax(a0, uu * uu);
        }
        const double ainv = alpha == a0 ? ainv0 : 1.0 / alpha;
#pragma unroll
        for (int u = 0; u < MC_RPT; ++u)
        {
            const int r = tid + u * T;
            if (r < p)
            {
                const double cv = cr[u]; // 0 in rows that take no part
                const double l = cv * ainv;
                Cp[(size_t)t * p + r] = cv;
                a.Cg[(size_t)t * p + r] = cv;
                a.Lg[(size_t)j * p + r] = l;
                if (preg[u] > j)
                {
                    dreg[u] -= l * cv;
                    breg[u] -= l * bq;
                }
                if (r == q)
                {
                    a.dinv[r] = ainv;
                    a.ord[j] = q;
                }
            }
        }
        if (tid == 0)
        {
            ainv_s[t] = ainv;
            a.ainvg[t] = ainv;
        }
        // (no barrier here: the first thing another thread reads of this step -- Cp[t][.], ainv_s[t] -- it reads after
        // the two barriers of the next pivot search)
    }
#pragma unroll
    for (int u = 0; u < MC_RPT; ++u)
    {
        const int r = tid + u * T;
        if (r < p)
        {
            a.dcur[r] = dreg[u];
            a.b[r] = breg[u];
            a.pos[r] = preg[u];
        }
    }
}

// A[i][k] -= sum_s (c_is / alpha_s) c_ks over the panel's steps, in their order; 64 x 64 tile per workgroup, 4 x 4 per thread
__global__ __launch_bounds__(256) void mchol_trail_kernel(MCholArgs a)
{
    __shared__ double li[MC_NB_MAX][64], ck[MC_NB_MAX][64];
    const int tid = threadIdx.x, p = a.p, nb = a.nb;
    const int i0 = blockIdx.y * 64, k0 = blockIdx.x * 64;
    for (int e = tid; e < nb * 64; e += 256)
    {
        const int s = e >> 6, c = e & 63;
        const double ai = a.ainvg[s];
        li[s][c] = i0 + c < p ? a.Cg[(size_t)s * p + i0 + c] * ai : 0.0;
        ck[s][c] = k0 + c < p ? a.Cg[(size_t)s * p + k0 + c] : 0.0;
    }
    __syncthreads();
    const int ty = tid >> 4, tx = tid & 15;
    double acc[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v)
        {
            const int i = i0 + ty * 4 + u, k = k0 + tx + 16 * v;
            acc[u][v] = (i < p && k < p) ? a.A[(size_t)i * p + k] : 0.0;
        }
    for (int s = 0; s < nb; ++s)
    {
        double l[4], c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
        {
            l[u] = li[s][ty * 4 + u];
            c[u] = ck[s][tx + 16 * u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v)
                acc[u][v] -= l[u] * c[v];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v)
        {
            const int i = i0 + ty * 4 + u, k = k0 + tx + 16 * v;
            if (i < p && k < p)
                a.A[(size_t)i * p + k] = acc[u][v];
        }
}

// z = D^-1 (L^-1 P b) is in b * dinv; L^T w = z from the last step back: w_{q_s} = z_{q_s} - sum over the rows eliminated
// after step s of Lg[s][r] w_r; sol[r] = w_r (rows are original indices: the permutation is already undone).
// Blocked: 32 steps at a time.  What the rows eliminated after the block contribute to each of its 32 equations is 32
// independent dot products (a wavefront per equation, lanes over the rows, one DPP sum each), the 32 x 32 triangle that
// couples the block's own pivots is gathered into LDS meanwhile and solved by the first wavefront -- two barriers per
// 32 steps where the step-by-step form had one per step (0.56 us each: 0.28 of the 1.65 ms of a p = 500 solve).
constexpr int MC_BS = 32;
__global__ __launch_bounds__(MC_T) void mchol_backsub_kernel(MCholArgs a, double *sol)
{
    extern __shared__ double lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = a.p;
    const int T = blockDim.x, nwaves = T >> 6;
    double *w = lds;                        // [p] the solution by original row (z until a row's step is reached)
    double *tri_s = w + p;                  // [MC_BS][MC_BS + 1]
    double *part = tri_s + MC_BS * (MC_BS + 1); // [MC_BS]
    int *pos = reinterpret_cast<int *>(part + MC_BS); // [p]
    int *qb = pos + p;                      // [MC_BS]
    for (int r = tid; r < p; r += T)
    {
        w[r] = a.b[r] * a.dinv[r];
        pos[r] = a.pos[r];
    }
    __syncthreads();
    for (int s1 = p - 1; s1 >= 0; s1 -= MC_BS)
    {
        const int s0 = s1 - (MC_BS - 1) > 0 ? s1 - (MC_BS - 1) : 0, nb = s1 - s0 + 1;
        if (tid < nb)
            qb[tid] = a.ord[s0 + tid];
        __syncthreads();
        // the triangle among the block's pivots: tri[k][j] = multiplier of row q_{s0+j} at step s0 + k (nonzero for j > k)
        for (int e = tid; e < nb * nb; e += T)
        {
            const int k = e / nb, j = e - k * nb;
            tri_s[k * (MC_BS + 1) + j] = a.Lg[(size_t)(s0 + k) * p + qb[j]];
        }
        // what the rows eliminated after the block (positions > s1) contribute to equation s0 + k
        for (int k = wave; k < nb; k += nwaves)
        {
            const double *Ls = a.Lg + (size_t)(s0 + k) * p;
            double acc = 0.0;
            for (int r = lane; r < p; r += 64)
                if (pos[r] > s1)
                    acc += Ls[r] * w[r];
            acc += wide_dpp<0xB1>(acc);
            acc += wide_dpp<0x4E>(acc);
            acc += wide_dpp<0x141>(acc);
            acc += wide_dpp<0x140>(acc);
            acc = (wide_bcast(acc, 0) + wide_bcast(acc, 16)) + (wide_bcast(acc, 32) + wide_bcast(acc, 48));
            if (lane == 0)
                part[k] = acc;
        }
        __syncthreads();
        if (wave == 0)
        {
            const int k = lane < nb ? lane : 0;
            double val = lane < nb ? w[qb[k]] - part[k] : 0.0;
            for (int j = nb - 1; j >= 1; --j)
            {
                const double wj = wide_bcast(val, j); // final for step s0 + j
                if (lane < j)
                    val -= tri_s[lane * (MC_BS + 1) + j] * wj;
            }
            if (lane < nb)
                w[qb[k]] = val;
        }
        __syncthreads();
    }
    for (int r = tid; r < p; r += T)
        sol[r] = w[r];
}

struct MCholBuffers
{
    std::mutex mu;
    int cap = 0, device = -1;
    double *A = nullptr, *Lg = nullptr, *Cg = nullptr, *vec = nullptr; // vec: ainvg | dcur | b | dinv | scal | rhs | sol
    int *ivec = nullptr;                                              // pos | ord
    bool attr_set = false;
};
static MCholBuffers &mchol_buffers()
{
    static MCholBuffers *b = new MCholBuffers; // (never destroyed: no HIP calls during static teardown)
    return *b;
}

// 0, or a GSLNLS_E_* code when the device cannot take it (the caller keeps the host routine)
// A_host != nullptr: the matrix itself from the host.  Otherwise jtj_dev (p x p on the device, left untouched) with
// diag_host and mu: A = J^T J + mu D^2 is formed on the device -- the 8 p^2 bytes do not travel.
static int mchol_device_solve_impl(int p, const double *A_host, const double *jtj_dev, const double *diag_host, double mu,
                                   const double *rhs_host, double *sol_host)
{
    if (p < 1 || p > MC_PMAX)
        return GSLNLS_E_UNSUPPORTED;
    MCholBuffers &B = mchol_buffers();
    std::lock_guard<std::mutex> lock(B.mu);
    int dev = 0;
    GSLNLS_HIP_OK(hipGetDevice(&dev));
    if (B.cap < p || B.device != dev)
    {
        // (the buffers belong to the device they were allocated on: a process that moved on with gslnls_set_device gets
        // new ones; hipFree takes pointers of any device)
        (void)hipFree(B.A);
        (void)hipFree(B.Lg);
        (void)hipFree(B.Cg);
        (void)hipFree(B.vec);
        (void)hipFree(B.ivec);
        B.A = B.Lg = B.Cg = B.vec = nullptr;
        B.ivec = nullptr;
        B.cap = 0;
        const size_t pp = (size_t)p * p;
        if (hipMalloc(&B.A, sizeof(double) * pp) != hipSuccess || hipMalloc(&B.Lg, sizeof(double) * pp) != hipSuccess ||
            hipMalloc(&B.Cg, sizeof(double) * (size_t)MC_NB_MAX * p) != hipSuccess ||
            hipMalloc(&B.vec, sizeof(double) * ((size_t)6 * p + MC_NB_MAX + 8)) != hipSuccess ||
            hipMalloc(&B.ivec, sizeof(int) * (size_t)2 * p) != hipSuccess)
        {
            (void)hipGetLastError();
            return GSLNLS_E_NODEVICE;
        }
        B.cap = p;
        B.device = dev;
        B.attr_set = false;
    }
    if (!B.attr_set)
    {
        if (hipFuncSetAttribute((const void *)mchol_panel_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MC_LDS_BYTES) !=
            hipSuccess)
        {
            (void)hipGetLastError();
            return GSLNLS_E_NODEVICE;
        }
        B.attr_set = true;
    }
    MCholArgs a;
    a.A = B.A;
    a.Lg = B.Lg;
    a.Cg = B.Cg;
    a.ainvg = B.vec;
    a.dcur = a.ainvg + MC_NB_MAX;
    a.b = a.dcur + p;
    a.dinv = a.b + p;
    a.scal = a.dinv + p;
    double *d_rhs = a.scal + 8, *d_sol = d_rhs + p, *d_dmp = d_sol + p;
    a.pos = B.ivec;
    a.ord = B.ivec + p;
    a.p = p;
    a.kb = 0;
    a.nb = 0;
    // panel width: NB columns of the panel in LDS
    const size_t fixed = sizeof(double) * ((size_t)2 * MC_NB_MAX + 8) + sizeof(MCholCand) * 2 * MC_W + 64;
    if (fixed + sizeof(double) * (size_t)2 * p > (size_t)MC_LDS_BYTES)
        return GSLNLS_E_UNSUPPORTED;
    int NB = (int)(((size_t)MC_LDS_BYTES - fixed) / (sizeof(double) * (size_t)p));
    NB = NB > MC_NB_MAX ? MC_NB_MAX : NB;
    if (A_host)
        GSLNLS_HIP_OK(hipMemcpy(B.A, A_host, sizeof(double) * (size_t)p * p, hipMemcpyHostToDevice));
    else
        GSLNLS_HIP_OK(hipMemcpy(d_dmp, diag_host, sizeof(double) * p, hipMemcpyHostToDevice));
    GSLNLS_HIP_OK(hipMemcpy(d_rhs, rhs_host, sizeof(double) * p, hipMemcpyHostToDevice));
    GSLNLS_HIP_OK(hipMemsetAsync(a.scal, 0, sizeof(double) * 8, 0));
    {
        long long g = ((long long)p * p + 256 * 8 - 1) / (256 * 8);
        g = g < 1 ? 1 : (g > 1024 ? 1024 : g);
        hipLaunchKernelGGL(mchol_init_kernel, dim3((unsigned)g), dim3(256), 0, 0, a, d_rhs, A_host ? nullptr : jtj_dev, d_dmp, mu);
    }
    // the panel workgroup is sized to p (a wavefront without rows still pays for every barrier and reduction)
    int T = 64 * ((p + 63) / 64);
    T = T > MC_T ? MC_T : T;
    const int tiles = (p + 63) / 64;
    for (int kb = 0; kb < p; kb += NB)
    {
        a.kb = kb;
        a.nb = p - kb < NB ? p - kb : NB;
        const size_t lds = fixed + sizeof(double) * (size_t)a.nb * p;
        hipLaunchKernelGGL(mchol_panel_kernel, dim3(1), dim3(T), lds, 0, a);
        if (kb + a.nb < p)
            hipLaunchKernelGGL(mchol_trail_kernel, dim3(tiles, tiles), dim3(256), 0, 0, a);
    }
    {
        const size_t bl = sizeof(double) * ((size_t)p + MC_BS * (MC_BS + 1) + MC_BS) + sizeof(int) * ((size_t)p + MC_BS) + 64;
        hipLaunchKernelGGL(mchol_backsub_kernel, dim3(1), dim3(T), bl, 0, a, d_sol);
    }
    GSLNLS_HIP_OK(hipMemcpy(sol_host, d_sol, sizeof(double) * p, hipMemcpyDeviceToHost));
    GSLNLS_HIP_OK(hipGetLastError());
    return GSLNLS_SUCCESS;
}

int mchol_device_solve(int p, const double *A_host, const double *rhs_host, double *sol_host)
{
    return mchol_device_solve_impl(p, A_host, nullptr, nullptr, 0.0, rhs_host, sol_host);
}
int mchol_device_solve_resident(int p, const double *jtj_dev, const double *diag_host, double mu, const double *rhs_host,
                                double *sol_host)
{
    if (!jtj_dev || !diag_host)
        return GSLNLS_EINVAL;
    return mchol_device_solve_impl(p, nullptr, jtj_dev, diag_host, mu, rhs_host, sol_host);
}

} // namespace gslnls

// test hook: (A + mu diag(d)^2) sol = rhs on the device (A: p x p symmetric, row-major)
extern "C" int gslnls_debug_mchol_solve(int p, const double *A, const double *diag, double mu, const double *rhs, double *sol)
{
    if (p < 1 || !A || !rhs || !sol)
        return GSLNLS_EINVAL;
    double *M = (double *)malloc(sizeof(double) * (size_t)p * p);
    if (!M)
        return GSLNLS_FAILURE;
    for (size_t e = 0; e < (size_t)p * p; ++e)
        M[e] = A[e];
    if (diag)
        for (int i = 0; i < p; ++i)
            M[(size_t)i * p + i] += mu * diag[i] * diag[i];
    const int rc = gslnls::mchol_device_solve(p, M, rhs, sol);
    free(M);
    return rc;
}

// test hook (no device needed): the same solve by the host routine of the large path (large_host.hpp: lg_mchol_solve,
// multiversioned for the host's vector width with the baseline's rounding) -- what runs below the device threshold
extern "C" int gslnls_debug_host_mchol_solve(int p, const double *A, const double *diag, double mu, const double *rhs,
                                             double *sol)
{
    if (p < 1 || !A || !rhs || !sol)
        return GSLNLS_EINVAL;
    std::vector<double> M(A, A + (size_t)p * p), r(rhs, rhs + p), x;
    if (diag)
        for (int i = 0; i < p; ++i)
            M[(size_t)i * p + i] += mu * diag[i] * diag[i];
    gslnls::lg_mchol_solve(p, M, r, x);
    for (int i = 0; i < p; ++i)
        sol[i] = x[i];
    return GSLNLS_SUCCESS;
}
