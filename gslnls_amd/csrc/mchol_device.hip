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
#include <atomic>
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
constexpr int MC_EXTRA_MAX = 2 * MC_PMAX + 2048; // doubles a tail may send to the host with the solution (MCholTail)

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

// ---- natural-order blocked Cholesky: the factorisation that serves the matrices an LM step actually meets ----------------
// J^T J + mu D^2 with mu > 0 is positive definite, and on a numerically positive definite matrix gsl_linalg_mcholesky
// never modifies anything: its pivoting only decides the ORDER in which the same factor is rounded.  Without the pivot
// search the factorisation is level 3: 64 columns per step --
//   cholb_panel_kernel   every workgroup factors the 64 x 64 diagonal block of the step in LDS (redundantly: same bits,
//                        no hand-off), workgroup 0 stores it, workgroup b > 0 solves its 64 rows of the panel against it
//                        (X L11^T = W21);
//   cholb_trail_kernel   W22 -= L21 L21^T on the matrix cores (v_mfma_f64_16x16x4_f64), one 64 x 64 tile per workgroup;
//   the right-hand side rides through both as one more row of the matrix (L y = b happens inside the factorisation);
//   cholb_back_kernel    L^T x = y, one launch per block from the last to the first.
// A pivot that is not safely positive (<= 1e-12 of the diagonal entry it started as, or NaN) raises a flag: the matrix
// is not numerically positive definite and the caller runs the pivoted, modified factorisation above on it.
constexpr int CB = 64;
constexpr int CB_LD = CB + 1; // LDS rows: conflict-free column walks
constexpr int CBT_LD = 68;    // MFMA staging tiles (as BD_LD)
typedef double cb_v4f64 __attribute__((ext_vector_type(4)));

// X L11^T = W21 for 64 rows of the panel: wavefront w owns rows 16 w .. 16 w + 15, lane = (row = lane % 16, column class
// = lane / 16) with the 16 columns class, class + 4, ... of its row in REGISTERS for all 64 steps.  Step J: the lane
// that owns column J scales it (x_J = r / L_JJ) and the other three classes of the row fetch it from that lane
// (ds_bpermute inside the wavefront -- with a class per wavefront it was a workgroup barrier per step, 64 of them, and
// 18 us of a 42 us panel), then every lane takes x_J L[c][J] out of its columns c > J.  J is a literal (a recursion, not
// a loop): every register index is static.  Same operations on the same values as before: not a bit changes.
template <int Q, int J>
__device__ __forceinline__ void cholb_trsm_update(double (&r)[16], double xj, const double *Lt, int cls)
{
    if constexpr (Q < 16)
    {
        constexpr int QJ = J >> 2;
        if constexpr (Q > QJ)
            r[Q] -= xj * Lt[(cls + 4 * Q) * CB_LD + J];
        else if constexpr (Q == QJ)
        {
            if (cls > (J & 3))
                r[Q] -= xj * Lt[(cls + 4 * Q) * CB_LD + J];
        }
        cholb_trsm_update<Q + 1, J>(r, xj, Lt, cls);
    }
}
// (step J starts when the factorisation, one wavefront over, has published column J: `ready` counts the columns of L11
// that are final in LDS)
template <int J>
__device__ __forceinline__ void cholb_trsm_steps(double (&r)[16], const double *Lt, const double *invd, int row, int cls,
                                                 const int *ready, int seen)
{
    if constexpr (J < CB)
    {
        while (seen <= J)
        {
            seen = __hip_atomic_load(ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (seen <= J)
                __builtin_amdgcn_s_sleep(1);
        }
        if (cls == (J & 3))
            r[J >> 2] *= invd[J];
        const double xj = wide_shfl(r[J >> 2], row + 16 * (J & 3));
        cholb_trsm_update<0, J>(r, xj, Lt, cls);
        cholb_trsm_steps<J + 1>(r, Lt, invd, row, cls, ready, seen);
    }
}

// One step of 64 columns, five wavefronts per workgroup.  The first wavefront of EVERY workgroup factors the 64 x 64
// diagonal block in its registers (lane = row, the rank-one update of a step as DPP row broadcasts: the natural-order form
// of wide_chol_reg, wide_core.hpp; redundantly in every workgroup: same bits, no hand-off between workgroups), writes each
// finished column of L11 to LDS and counts it in `ready`; the other four solve the workgroup's 64 rows of the panel
// against it (16 rows each, cholb_trsm_steps) ONE COLUMN BEHIND the factorisation instead of after it -- the rows are
// requested before anything else and the solve ends a step after the factor does.  A pivot that is not safely positive
// does not end the factorisation (a wavefront waiting for columns would wait for ever): it is remembered, the arithmetic
// runs on, the flag tells the host to discard all of it.  Workgroup 0 has no rows: its four wavefronts store L11 when it is
// complete, its first carries the right-hand side (one more row of the matrix).
constexpr int CBP_T = 320;
__global__ __launch_bounds__(CBP_T) void cholb_panel_kernel(const double *W, double *Lf, int p, int k0, const double *dorig, int *flag,
                                                            double *yv, double *dinvg)
{
    __shared__ double Lt[CB * CB_LD], invd[CB];
    __shared__ int ready_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nb = p - k0 < CB ? p - k0 : CB;
    if (tid == 0)
        ready_s = 0;
    // this workgroup's rows of the panel (workgroup b > 0: rows r0 .. r0 + 63): thread (row = 16 (wave - 1) + lane % 16,
    // class = lane / 16) of wavefronts 1 .. 4 holds columns class, class + 4, ... of row r0 + row
    const int r0 = k0 + CB * (int)blockIdx.x;
    const int prow = 16 * (wave - 1) + (lane & 15), cls = lane >> 4;
    // (every global load of the kernel is issued here, without a condition in front of it -- a load under a condition is a
    // branch and a wait per load; what does not exist reads element 0 and is discarded)
    double rr[16];
#pragma unroll
    for (int q = 0; q < 16; ++q)
    {
        const int c = cls + 4 * q;
        const bool ok = wave > 0 && blockIdx.x > 0 && r0 + prow < p && c < nb;
        rr[q] = W[ok ? (size_t)(r0 + prow) * p + k0 + c : 0];
        rr[q] = ok ? rr[q] : 0.0;
    }
    // the diagonal block, lower triangle, identity beyond nb (a partial last block)
    {
        constexpr int NL = (CB * CB + CBP_T - 1) / CBP_T;
        double dv[NL];
#pragma unroll
        for (int it = 0; it < NL; ++it)
        {
            const int e = tid + CBP_T * it, i = e >> 6, j = e & 63;
            dv[it] = W[(e < CB * CB && i < nb && j <= i) ? (size_t)(k0 + i) * p + k0 + j : 0];
        }
#pragma unroll
        for (int it = 0; it < NL; ++it)
        {
            const int e = tid + CBP_T * it, i = e >> 6, j = e & 63;
            if (e < CB * CB)
                Lt[i * CB_LD + j] = (i < nb && j <= i) ? dv[it] : (i == j ? 1.0 : 0.0);
        }
    }
    __syncthreads(); // (the only one: from here on the wavefronts meet through `ready`)
    if (wave == 0)
    {
        __builtin_amdgcn_s_setprio(3); // (five wavefronts on four SIMDs: this one is the critical path of the launch)
        constexpr int R = CB / 16;
        double m[CB];
#pragma unroll
        for (int k = 0; k < CB; ++k)
            m[k] = k < lane ? Lt[lane * CB_LD + k] : (k > lane ? Lt[k * CB_LD + lane] : 0.0);
        double dg = Lt[lane * CB_LD + lane];
        const double thr = lane < nb ? fmax(1e-12 * dorig[k0 + lane], DBL_EPSILON) : 0.0;
        wide_lds_sync(); // every lane has its row before the factor overwrites the block
        bool bad = false;
        // 1 / sqrt(d): the hardware's estimate and two Newton steps (y <- y + y/2 (1 - d y^2)): rounding level after the
        // second, 7 dependent operations where 1.0 / sqrt(d) was about 35; 1 / d as its square: no division in a step.
        // This chain, 64 times, is the critical path of the launch, and a wavefront issues in order: the chain of step
        // j + 1 (it needs d_{j+1} only, final as soon as column j + 1 of the update is) is written BETWEEN the pieces of
        // step j's rank-one update, one operation per eighth, and scheduling barriers keep it there -- behind the update
        // it waited for nothing but itself, 230 cycles of every step.
        auto step = [&](auto self, auto jj, double dj, double rs, double ainv) __attribute__((always_inline)) -> void {
            constexpr int j = decltype(jj)::value;
            if constexpr (j < CB)
            {
                // column j - 1 is published here: its LDS writes have long landed, the release waits for nothing
                // (every lane stores the same word, here and into invd: a store under `if (lane == ..)` is a branch, and
                // the compiler moves the chain below across it, back to where it is waited for)
                if constexpr (j > 0)
                    __hip_atomic_store(&ready_s, j, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                const double v = lane > j ? m[j] : 0.0;
                const double t = ainv * v;
                dg -= t * v;
                Lt[lane * CB_LD + j] = lane == j ? dj * rs : v * rs; // column j of L11 (0 above the diagonal)
                invd[j] = rs;
                if constexpr (j + 1 < CB)
                {
                    double vb[R];
                    wide_row_copies<R>(v, vb);
                    const double tn = -t;
                    wide_chol_update_range<j + 1, j + 2, CB, R>(m, vb, tn); // (column j + 1 first: the next step's v)
                    const double dn = wide_bcast(dg, j + 1), tj = wide_bcast(thr, j + 1);
                    // (below eps the reference's alpha = max(eps, |d|, ..) replaces the pivot: not this routine's case; NaN too)
                    bad = bad || !(dn > tj);
                    constexpr int F0 = j + 2, NF = CB - F0;
#define GSLNLS_CB_PIECE(c)                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                                \
    wide_chol_update_range<F0 + NF * (c) / 8, F0 + NF * ((c) + 1) / 8, CB, R>(m, vb, tn);                               \
    __builtin_amdgcn_sched_barrier(0);
                    double y = __builtin_amdgcn_rsq(dn);
                    GSLNLS_CB_PIECE(0)
                    double e = -dn * y, h = 0.5 * y;
                    GSLNLS_CB_PIECE(1)
                    e = fma(e, y, 1.0);
                    GSLNLS_CB_PIECE(2)
                    y = fma(h, e, y);
                    GSLNLS_CB_PIECE(3)
                    e = -dn * y, h = 0.5 * y;
                    GSLNLS_CB_PIECE(4)
                    e = fma(e, y, 1.0);
                    GSLNLS_CB_PIECE(5)
                    y = fma(h, e, y);
                    GSLNLS_CB_PIECE(6)
                    const double an = y * y;
                    GSLNLS_CB_PIECE(7)
#undef GSLNLS_CB_PIECE
                    self(self, WideInt<j + 1>{}, dn, y, an);
                }
            }
        };
        {
            const double d0 = wide_bcast(dg, 0), t0 = wide_bcast(thr, 0);
            bad = !(d0 > t0);
            double y = __builtin_amdgcn_rsq(d0);
            y = fma(0.5 * y, fma(-d0 * y, y, 1.0), y);
            y = fma(0.5 * y, fma(-d0 * y, y, 1.0), y);
            step(step, WideInt<0>{}, d0, y, y * y);
        }
        if (lane == 0)
            __hip_atomic_store(&ready_s, CB, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (blockIdx.x != 0)
            return;
        if (bad && lane == 0)
            *flag = 1;
        // the right-hand side rides along as one more row of the matrix: its slice of this step, L11 y = b (forward
        // substitution block by block; the trailing update of the vector is cholb_trail_kernel's last workgroups)
        double v = lane < nb ? yv[k0 + lane] : 0.0;
        for (int j = 0; j < nb; ++j)
        {
            const double yj = wide_bcast(v, j) * invd[j];
            if (lane == j)
                v = yj;
            else if (lane > j && lane < nb)
                v -= Lt[lane * CB_LD + j] * yj;
        }
        if (lane < nb)
            yv[k0 + lane] = v;
        return;
    }
    if (blockIdx.x == 0)
    {
        // workgroup 0 has no rows of the panel: its wavefronts 1 .. 4 store L11 once it is complete
        while (__hip_atomic_load(&ready_s, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < CB)
            __builtin_amdgcn_s_sleep(4);
        for (int e = tid - 64; e < CB * CB; e += CBP_T - 64)
        {
            const int i = e >> 6, j = e & 63;
            if (i < nb && j <= i)
                Lf[(size_t)(k0 + i) * p + k0 + j] = Lt[i * CB_LD + j];
        }
        if (tid - 64 < nb)
            dinvg[k0 + tid - 64] = invd[tid - 64]; // 1 / L_jj for the back substitution
        return;
    }
    // rows r0 .. r0 + 63 of the panel: X L11^T = W21, a column behind the factorisation
    cholb_trsm_steps<0>(rr, Lt, invd, lane & 15, cls, &ready_s, 0);
    if (r0 + prow < p)
    {
#pragma unroll
        for (int q = 0; q < 16; ++q)
        {
            const int c = cls + 4 * q;
            if (c < nb)
                Lf[(size_t)(r0 + prow) * p + k0 + c] = rr[q];
        }
    }
}

// ---- round 5: the diagonal block on FOUR wavefronts ------------------------------------------------------------------------
// The kernel above spends its 25 us on one wavefront: a step issues the rank-one update of all 63 - j columns behind the
// pivot plus ~45 instructions of pivot arithmetic, 64 times, in order.  Only the update of the columns up to the end of
// the current 16-column sub-block is needed before the next pivot.  Here wavefront h of the first four owns columns
// 16 h .. 16 h + 15 of the block (lane = row, 16 registers): it takes the pivots of its own sub-block (the step of the
// kernel above, on 16 columns) and publishes, per step, the multiplier column t = v / d and the column v itself in LDS
// (plus 1 / L_jj and L_jj); the wavefronts behind it apply the step to THEIR columns from there (3 LDS reads + 16 DPP
// multiply-adds), a step or two behind the owner -- when the owner's sub-block is done the next wavefront has caught up
// and takes over.  What the owner does NOT do any more, because it is not needed for the next pivot: form the column of
// L (v / L_jj: whoever reads L multiplies -- same product, same rounding), make all four row-group copies of v for the
// DPP broadcasts (one suffices for 16 columns), test every pivot against its threshold as it goes (a lane's diagonal
// entry is final once its pivot is taken: the wavefront tests its 16 at the end).  The critical path is 64 x (16-column
// step of ~45 instructions) instead of 64 x (64-column step of ~110); every multiply-add is the one the kernel above
// issues, on the same operands in the same order (fma(-t_i, v_k, S_ik), steps ascending): the factor is bit-identical.
// Wavefronts 4 .. 7 solve the workgroup's panel rows a column behind, as in the kernel above.
constexpr int CBQ_T = 512;

template <int K, int KEND>
__device__ __forceinline__ void cholb_upd16(double (&m)[16], double vbh, double tn)
{
    if constexpr (K < KEND)
    {
        wide_fmac_rowbcast<K>(m[K], vbh, tn);
        cholb_upd16<K + 1, KEND>(m, vbh, tn);
    }
}

// v of row 16 H + l % 16 in lane l (the one row-group copy the DPP broadcasts of sub-block H need)
template <int H>
__device__ __forceinline__ double cholb_row_copy(double v)
{
    const long long bits = __double_as_longlong(v);
    const unsigned int lo = (unsigned int)(bits & 0xffffffffll), hi = (unsigned int)(bits >> 32);
    const auto l16 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false); // [0] = rows [0, 0, 2, 2], [1] = rows [1, 1, 3, 3]
    const auto h16 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const unsigned int ls = l16[H & 1], hs = h16[H & 1];
    const auto l32 = __builtin_amdgcn_permlane32_swap(ls, ls, false, false); // [0] = the lower row of the pair everywhere, [1] the upper
    const auto h32 = __builtin_amdgcn_permlane32_swap(hs, hs, false, false);
    double r = __longlong_as_double(((long long)h32[H >> 1] << 32) | l32[H >> 1]);
    asm volatile("s_nop 1" : "+v"(r)); // (a DPP operand written by the instruction before: 2 wait states)
    return r;
}

// what the diagonal-block wavefronts publish per column J: the multipliers t, the column v, 1 / L_JJ and L_JJ
struct CholbPub
{
    double T[CB * CB], V[CB * CB], invd[CB], ldiag[CB];
};

// wavefront H of the diagonal block: columns 16 H .. 16 H + 15.  Returns true when a pivot of its sub-block was not safely positive.
template <int H>
__device__ __forceinline__ bool cholb_diag_wave(const double *Lt, CholbPub &P, int *ready, const double *dorig, int k0, int nb, int lane)
{
    constexpr int C0 = 16 * H;
    double m[16];
#pragma unroll
    for (int c = 0; c < 16; ++c)
        m[c] = (C0 + c) < lane ? Lt[lane * CB_LD + C0 + c] : 0.0; // (entries on and above the diagonal never reach a result)
    double dg = Lt[lane * CB_LD + lane];
    const double thr = lane < nb ? fmax(1e-12 * dorig[k0 + lane], DBL_EPSILON) : 0.0;
    // the steps of the sub-blocks in front: applied from what their owners published
    int seen = 0;
    for (int J = 0; J < C0; ++J)
    {
        while (seen <= J)
        {
            seen = __hip_atomic_load(ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (seen <= J)
                __builtin_amdgcn_s_sleep(1);
        }
        const double t = P.T[J * CB + lane], v = P.V[J * CB + lane];
        double vbh = P.V[J * CB + C0 + (lane & 15)];
        dg -= t * v;
        const double tn = -t;
        asm volatile("s_nop 1" : "+v"(vbh)); // (a DPP operand written by the instruction before: 2 wait states)
        cholb_upd16<0, 16>(m, vbh, tn);
    }
    auto step = [&](auto self, auto jj, double dj, double rs, double ainv) __attribute__((always_inline)) -> void {
        constexpr int j = decltype(jj)::value; // local column; J = C0 + j
        if constexpr (j < 16)
        {
            constexpr int J = C0 + j;
            if constexpr (J > 0)
                __hip_atomic_store(ready, J, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); // columns < J are final
            const double v = lane > J ? m[j] : 0.0;
            const double t = ainv * v;
            dg -= t * v;
            P.T[J * CB + lane] = t;
            P.V[J * CB + lane] = v;
            P.invd[J] = rs; // (every lane stores the same word: a store under `if (lane == ..)` is a branch)
            P.ldiag[J] = dj * rs;
            if constexpr (j + 1 < 16)
            {
                const double vbh = cholb_row_copy<H>(v);
                const double tn = -t;
                cholb_upd16<j + 1, j + 2>(m, vbh, tn); // (column j + 1 first: the next step's v)
                const double dn = wide_bcast(dg, J + 1);
                constexpr int F0 = j + 2, NF = 16 - F0;
#define GSLNLS_CBQ_PIECE(c)                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                                \
    cholb_upd16<F0 + NF * (c) / 8, F0 + NF * ((c) + 1) / 8>(m, vbh, tn);                                                \
    __builtin_amdgcn_sched_barrier(0);
                double y = __builtin_amdgcn_rsq(dn);
                GSLNLS_CBQ_PIECE(0)
                double e = -dn * y, h = 0.5 * y;
                GSLNLS_CBQ_PIECE(1)
                e = fma(e, y, 1.0);
                GSLNLS_CBQ_PIECE(2)
                y = fma(h, e, y);
                GSLNLS_CBQ_PIECE(3)
                e = -dn * y, h = 0.5 * y;
                GSLNLS_CBQ_PIECE(4)
                e = fma(e, y, 1.0);
                GSLNLS_CBQ_PIECE(5)
                y = fma(h, e, y);
                GSLNLS_CBQ_PIECE(6)
                const double an = y * y;
                GSLNLS_CBQ_PIECE(7)
#undef GSLNLS_CBQ_PIECE
                self(self, WideInt<j + 1>{}, dn, y, an);
            }
        }
    };
    {
        const double d0 = wide_bcast(dg, C0);
        double y = __builtin_amdgcn_rsq(d0);
        y = fma(0.5 * y, fma(-d0 * y, y, 1.0), y);
        y = fma(0.5 * y, fma(-d0 * y, y, 1.0), y);
        step(step, WideInt<0>{}, d0, y, y * y);
    }
    // the whole sub-block is final: the wavefront behind takes over, the solving wavefronts go on
    __hip_atomic_store(ready, C0 + 16, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    // a lane's diagonal entry has not changed since its pivot was taken (its v is 0 from then on): the 16 pivots of the
    // sub-block against their thresholds, at once (below eps the reference's alpha = max(eps, |d|, ..) replaces the pivot:
    // not this routine's case; NaN too)
    const bool mine = lane >= C0 && lane < C0 + 16;
    return __ballot(mine && !(dg > thr)) != 0ull;
}

// X L11^T = W21 for 64 rows of the panel as cholb_trsm_steps does it, with L11 read as the published v / L_JJ (the product
// the one-wavefront kernel stores: same operands, same rounding)
template <int Q, int J>
__device__ __forceinline__ void cholb_trsm_update_v(double (&r)[16], double xj, const double *Vj, double rs, int cls)
{
    if constexpr (Q < 16)
    {
        constexpr int QJ = J >> 2;
        if constexpr (Q > QJ)
            r[Q] -= xj * __dmul_rn(Vj[cls + 4 * Q], rs);
        else if constexpr (Q == QJ)
        {
            if (cls > (J & 3))
                r[Q] -= xj * __dmul_rn(Vj[cls + 4 * Q], rs);
        }
        cholb_trsm_update_v<Q + 1, J>(r, xj, Vj, rs, cls);
    }
}
template <int J>
__device__ __forceinline__ void cholb_trsm_steps_v(double (&r)[16], const CholbPub &P, int row, int cls, const int *ready, int seen)
{
    if constexpr (J < CB)
    {
        while (seen <= J)
        {
            seen = __hip_atomic_load(ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (seen <= J)
                __builtin_amdgcn_s_sleep(1);
        }
        const double rs = P.invd[J];
        if (cls == (J & 3))
            r[J >> 2] *= rs;
        const double xj = wide_shfl(r[J >> 2], row + 16 * (J & 3));
        cholb_trsm_update_v<0, J>(r, xj, P.V + J * CB, rs, cls);
        cholb_trsm_steps_v<J + 1>(r, P, row, cls, ready, seen);
    }
}

// The part of a panel step behind its inputs (shared by cholb_panel4_kernel and cholb_step_kernel): wavefronts 0 .. 3 factor
// the diagonal block from Lt, wavefronts 4 .. 7 solve the workgroup's rows rr of the panel a column behind; workgroup 0
// stores L11 and solves the right-hand side's slice (from yhead in LDS when the caller has updated it there, else yv).
__device__ __forceinline__ void cholb_panel4_body(const double *Lt, CholbPub &P, int *ready_sp, double (&rr)[16], double *Lf, int p, int k0,
                                                  int nb, int r0, int prow, int cls, const double *dorig, int *flag, double *yv,
                                                  double *dinvg, const double *yhead)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int &ready_s = *ready_sp;
    if (wave < 4)
    {
        __builtin_amdgcn_s_setprio(3); // (the chain of pivots is the critical path of the launch)
        bool bad;
        switch (wave)
        {
        case 0: bad = cholb_diag_wave<0>(Lt, P, &ready_s, dorig, k0, nb, lane); break;
        case 1: bad = cholb_diag_wave<1>(Lt, P, &ready_s, dorig, k0, nb, lane); break;
        case 2: bad = cholb_diag_wave<2>(Lt, P, &ready_s, dorig, k0, nb, lane); break;
        default: bad = cholb_diag_wave<3>(Lt, P, &ready_s, dorig, k0, nb, lane); break;
        }
        if (blockIdx.x != 0)
            return;
        if (bad && lane == 0)
            *flag = 1;
        return;
    }
    if (blockIdx.x == 0)
    {
        // workgroup 0 has no rows of the panel.  Its fifth wavefront takes the right-hand side through this step -- it rides
        // along as one more row of the matrix: L11 y = b for its slice -- A COLUMN BEHIND the factorisation, like the panel
        // rows of the other workgroups.  (Until the end of round 5 the first wavefront did this AFTER its part of the
        // factorisation and after the last column: 64 dependent steps of two LDS reads, a broadcast and a multiply-add,
        // 280 clocks each -- in-kernel stamps: 18 k of the launch's 47 k clocks, the tail every panel step waited for.)
        if (wave == 4)
        {
            double v = lane < nb ? (yhead ? yhead[lane] : yv[k0 + lane]) : 0.0;
            int seen = 0;
            for (int j = 0; j < nb; ++j)
            {
                while (seen <= j)
                {
                    seen = __hip_atomic_load(&ready_s, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (seen <= j)
                        __builtin_amdgcn_s_sleep(1);
                }
                const double rs = P.invd[j];
                const double yj = wide_bcast(v, j) * rs;
                if (lane == j)
                    v = yj;
                else if (lane > j && lane < nb)
                    v -= __dmul_rn(P.V[j * CB + lane], rs) * yj;
            }
            if (lane < nb)
                yv[k0 + lane] = v;
        }
        // ... and wavefronts 4 .. 7 store L11 once it is complete
        while (__hip_atomic_load(&ready_s, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < CB)
            __builtin_amdgcn_s_sleep(4);
        for (int e = tid - 256; e < CB * CB; e += CBQ_T - 256)
        {
            const int i = e >> 6, j = e & 63;
            if (i < nb && j <= i)
                Lf[(size_t)(k0 + i) * p + k0 + j] = i == j ? P.ldiag[j] : __dmul_rn(P.V[j * CB + i], P.invd[j]);
        }
        if (tid - 256 < nb)
            dinvg[k0 + tid - 256] = P.invd[tid - 256]; // 1 / L_jj for the back substitution
        return;
    }
    // rows r0 .. r0 + 63 of the panel: X L11^T = W21, a column behind the factorisation
    cholb_trsm_steps_v<0>(rr, P, lane & 15, cls, &ready_s, 0);
    if (r0 + prow < p)
    {
#pragma unroll
        for (int q = 0; q < 16; ++q)
        {
            const int c = cls + 4 * q;
            if (c < nb)
                Lf[(size_t)(r0 + prow) * p + k0 + c] = rr[q];
        }
    }
}

__global__ __launch_bounds__(CBQ_T) void cholb_panel4_kernel(const double *W, double *Lf, int p, int k0, const double *dorig, int *flag,
                                                             double *yv, double *dinvg)
{
    __shared__ double Lt[CB * CB_LD];
    __shared__ CholbPub P;
    __shared__ int ready_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nb = p - k0 < CB ? p - k0 : CB;
    if (tid == 0)
        ready_s = 0;
    // this workgroup's rows of the panel (workgroup b > 0: rows r0 .. r0 + 63): thread (row = 16 (wave - 4) + lane % 16,
    // class = lane / 16) of wavefronts 4 .. 7 holds columns class, class + 4, ... of row r0 + row
    const int r0 = k0 + CB * (int)blockIdx.x;
    const int prow = 16 * (wave - 4) + (lane & 15), cls = lane >> 4;
    double rr[16];
#pragma unroll
    for (int q = 0; q < 16; ++q)
    {
        const int c = cls + 4 * q;
        const bool ok = wave >= 4 && blockIdx.x > 0 && r0 + prow < p && c < nb;
        rr[q] = W[ok ? (size_t)(r0 + prow) * p + k0 + c : 0];
        rr[q] = ok ? rr[q] : 0.0;
    }
    // the diagonal block, lower triangle, identity beyond nb (a partial last block)
    {
        constexpr int NL = (CB * CB + CBQ_T - 1) / CBQ_T;
        double dv[NL];
#pragma unroll
        for (int it = 0; it < NL; ++it)
        {
            const int e = tid + CBQ_T * it, i = e >> 6, j = e & 63;
            dv[it] = W[(e < CB * CB && i < nb && j <= i) ? (size_t)(k0 + i) * p + k0 + j : 0];
        }
#pragma unroll
        for (int it = 0; it < NL; ++it)
        {
            const int e = tid + CBQ_T * it, i = e >> 6, j = e & 63;
            if (e < CB * CB)
                Lt[i * CB_LD + j] = (i < nb && j <= i) ? dv[it] : (i == j ? 1.0 : 0.0);
        }
    }
    __syncthreads(); // (the only one: from here on the wavefronts meet through `ready`; Lt is only read after it)
    cholb_panel4_body(Lt, P, &ready_s, rr, Lf, p, k0, nb, r0, prow, cls, dorig, flag, yv, dinvg, nullptr);
}

// W[I][J] -= L[I][kblk] L[J][kblk]^T for the tiles I >= J behind the panel (tile index from blockIdx.x, lower triangle)
// part: 0 all tiles (+ the right-hand side), 1 the tiles of the NEXT panel's column block only, J = 0 (+ the right-hand
// side), 2 the tiles behind it, J >= 1 -- round 5: part 1 is all the next panel kernel waits for, part 2 runs on a
// second stream beside that panel (mchol_device_solve_impl).  A tile is computed the same way whichever launch it is in.
__global__ __launch_bounds__(256, 2) void cholb_trail_kernel(double *W, const double *Lf, int p, int k0, int ntile, double *yv, int part)
{
    __shared__ double tI[CB * CBT_LD], tJ[CB * CBT_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kk = lane >> 4, ii = lane & 15;
    const int t = blockIdx.x;
    if (t >= ntile)
    {
        // y[i] -= L[i][k-block] . y[k-block] for the rows behind the panel: 256 rows per workgroup, one per thread
        __shared__ double yk[CB];
        if (tid < CB)
            yk[tid] = yv[k0 + tid];
        __syncthreads();
        const int i = k0 + CB + (t - ntile) * 256 + tid;
        if (i < p)
        {
            // (all 64 loads of the row in flight before the first is added: the rows were written by the panel kernel on
            // other XCDs, a load is 2 us, and eight at a time made this workgroup the last of the launch by 10 us)
            const double *row = Lf + (size_t)i * p + k0;
            double rv[CB];
#pragma unroll
            for (int c = 0; c < CB; ++c)
                rv[c] = row[c];
            double s0 = 0.0;
#pragma unroll
            for (int c = 0; c < CB; ++c)
                s0 += rv[c] * yk[c];
            yv[i] -= s0;
        }
        return;
    }
    int I, J;
    if (part == 1)
    {
        I = t;
        J = 0;
    }
    else
    {
        I = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
        while (I * (I + 1) / 2 > t)
            --I;
        while ((I + 1) * (I + 2) / 2 <= t)
            ++I;
        J = t - I * (I + 1) / 2;
        if (part == 2)
        {
            // the lower triangle of the blocks behind the next panel's column: (I, J) -> (I + 1, J + 1)
            I += 1;
            J += 1;
        }
    }
    const int base = k0 + CB, ri = base + CB * I, rj = base + CB * J;
    // (k0 + 64 <= p here: the last panel has no trailing matrix)
    // Every global load of the workgroup is issued before the first is waited for -- the 16 + 16 values of the two panel
    // blocks and the 16 entries of W this thread will update: a load under a condition is a branch and a wait, and as
    // loops of "load, wait, store" (32 + 16 round trips of 1-2 us through caches the panel kernel filled on other XCDs)
    // this was 18 us of launch for 1 us of matrix-core work.  Rows and entries that do not exist read element 0 instead
    // and are discarded.
    double wv[16], li[16], lj[16];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r)
        {
            const int gi = ri + wave * 16 + 4 * r + kk, gj = rj + b * 16 + ii;
            const bool ok = gi < p && gj < p && gj <= gi;
            wv[b * 4 + r] = W[ok ? (size_t)gi * p + gj : 0];
        }
#pragma unroll
    for (int it = 0; it < 16; ++it)
    {
        const int e = tid + 256 * it, r = e >> 6, c = e & 63;
        li[it] = Lf[ri + r < p ? (size_t)(ri + r) * p + k0 + c : 0];
        lj[it] = Lf[rj + r < p ? (size_t)(rj + r) * p + k0 + c : 0];
    }
#pragma unroll
    for (int it = 0; it < 16; ++it)
    {
        const int e = tid + 256 * it, r = e >> 6, c = e & 63;
        tI[r * CBT_LD + c] = ri + r < p ? li[it] : 0.0;
        tJ[r * CBT_LD + c] = rj + r < p ? lj[it] : 0.0;
    }
    __syncthreads();
    cb_v4f64 acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
        acc[b] = (cb_v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int c = 0; c < 16; ++c)
    {
        // lane l supplies A'[l % 16][l / 16] = L_I[16 wave + ii][4 c + kk] and B[l / 16][l % 16] = L_J[16 b + ii][4 c + kk]
        const double va = tI[(wave * 16 + ii) * CBT_LD + c * 4 + kk];
        double vb[4];
#pragma unroll
        for (int b = 0; b < 4; ++b)
            vb[b] = tJ[(b * 16 + ii) * CBT_LD + c * 4 + kk];
#pragma unroll
        for (int b = 0; b < 4; ++b)
            acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(va, vb[b], acc[b], 0, 0, 0);
    }
    // result register r of block b: element (16 wave + 4 r + kk, 16 b + ii) of the tile
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r)
        {
            const int gi = ri + wave * 16 + 4 * r + kk, gj = rj + b * 16 + ii;
            if (gi < p && gj < p && gj <= gi)
                W[(size_t)gi * p + gj] = wv[b * 4 + r] - acc[b][r];
        }
}

// ---- round 5: one launch per step -----------------------------------------------------------------------------------------------
// cholb_step_kernel(k0) = the panel of column block k0 AND what is left of the trailing update of the panel before it
// (kp = k0 - 64), side by side in one launch:
//   workgroups 0 .. nrb - 1   the panel (cholb_panel4_kernel's work).  Their inputs -- the diagonal tile and each
//                             workgroup's own tile of column block k0 -- still lack the update of panel kp: they apply it
//                             themselves, on the fly, from L's column block kp (the tile arithmetic of cholb_trail_kernel:
//                             same staging, same MFMA chain, w - acc), into LDS / registers; nothing of column block k0 is
//                             written back to W (nobody reads it again).  Workgroup 0 also takes the update of the
//                             right-hand side's slice (wavefront 4, while the others update tiles).
//   the next nrest            tiles (I, J), 1 <= J <= I, behind column block k0: W -= L[I][kp] L[J][kp]^T in place, one
//                             tile per workgroup and trip (eight wavefronts: two column blocks each -- the same chains);
//   the last nrhs             y[i] -= L[i][kp] . y[kp] for the rows behind the panel, 512 per workgroup.
// The panel's chain of 64 pivots (19.6 us) no longer waits for the whole trailing update (6 - 17 us) and a launch: a step
// is one launch of max(panel + its own update, rest) instead of two launches end to end.  Same bits as the two-launch
// form (GSLNLS_LARGE_STEP_V1=1 keeps it).
__device__ __forceinline__ double cholb_row_dot(const double *row, const double *yk)
{
    double rv[CB];
#pragma unroll
    for (int c = 0; c < CB; ++c)
        rv[c] = row[c];
    double s0 = 0.0;
#pragma unroll
    for (int c = 0; c < CB; ++c)
        s0 += rv[c] * yk[c];
    return s0;
}

__global__ __launch_bounds__(CBQ_T) void cholb_step_kernel(double *W, double *Lf, int p, int k0, const double *dorig, int *flag, double *yv,
                                                           double *dinvg, int nrb, int nrest, int ntile_rest)
{
    constexpr int TILE = CB * CBT_LD;
    constexpr int PUBD = (int)(sizeof(CholbPub) / sizeof(double));
    static_assert(PUBD <= 2 * TILE, "the published columns alias the two staged tiles, not the third");
    __shared__ double Lt[CB * CB_LD];
    __shared__ __attribute__((aligned(16))) double stage[3 * TILE];
    __shared__ double yk_s[CB], yhead_s[CB];
    __shared__ int ready_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kk = lane >> 4, ii = lane & 15;
    const int kp = k0 - CB;
    double *tD = stage, *tR = stage + TILE, *tOut = stage + 2 * TILE;
    if ((int)blockIdx.x >= nrb + nrest)
    {
        // ---- the right-hand side behind the panel ----
        if (tid < CB)
            yk_s[tid] = yv[kp + tid];
        __syncthreads();
        const int i = k0 + CB + ((int)blockIdx.x - nrb - nrest) * CBQ_T + tid;
        if (i < p)
            yv[i] -= cholb_row_dot(Lf + (size_t)i * p + kp, yk_s);
        return;
    }
    if ((int)blockIdx.x >= nrb)
    {
        // ---- tiles behind column block k0 ----
        const int w4 = wave & 3, bh = (wave >> 2) * 2; // rows 16 w4 .. of the tile, column blocks bh, bh + 1
        for (int t = (int)blockIdx.x - nrb; t < ntile_rest; t += nrest)
        {
            int I = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
            while (I * (I + 1) / 2 > t)
                --I;
            while ((I + 1) * (I + 2) / 2 <= t)
                ++I;
            const int J = t - I * (I + 1) / 2 + 1;
            I += 1;
            const int ri = k0 + CB * I, rj = k0 + CB * J;
            double wv[8], li[8], lj[8];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                {
                    const int gi = ri + w4 * 16 + 4 * r + kk, gj = rj + (bh + b) * 16 + ii;
                    const bool ok = gi < p && gj < p && gj <= gi;
                    wv[b * 4 + r] = W[ok ? (size_t)gi * p + gj : 0];
                }
#pragma unroll
            for (int it = 0; it < 8; ++it)
            {
                const int e = tid + CBQ_T * it, r = e >> 6, c = e & 63;
                li[it] = Lf[ri + r < p ? (size_t)(ri + r) * p + kp + c : 0];
                lj[it] = Lf[rj + r < p ? (size_t)(rj + r) * p + kp + c : 0];
            }
#pragma unroll
            for (int it = 0; it < 8; ++it)
            {
                const int e = tid + CBQ_T * it, r = e >> 6, c = e & 63;
                tD[r * CBT_LD + c] = ri + r < p ? li[it] : 0.0;
                tR[r * CBT_LD + c] = rj + r < p ? lj[it] : 0.0;
            }
            __syncthreads();
            cb_v4f64 acc[2];
#pragma unroll
            for (int b = 0; b < 2; ++b)
                acc[b] = (cb_v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
            for (int c = 0; c < 16; ++c)
            {
                const double va = tD[(w4 * 16 + ii) * CBT_LD + c * 4 + kk];
                double vb[2];
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    vb[b] = tR[((bh + b) * 16 + ii) * CBT_LD + c * 4 + kk];
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(va, vb[b], acc[b], 0, 0, 0);
            }
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                {
                    const int gi = ri + w4 * 16 + 4 * r + kk, gj = rj + (bh + b) * 16 + ii;
                    if (gi < p && gj < p && gj <= gi)
                        W[(size_t)gi * p + gj] = wv[b * 4 + r] - acc[b][r];
                }
            __syncthreads(); // (the staged tiles are rewritten by the next trip)
        }
        return;
    }
    // ---- the panel of column block k0, its inputs updated on the fly ----
    const int nb = p - k0 < CB ? p - k0 : CB;
    CholbPub &P = *reinterpret_cast<CholbPub *>(stage);
    if (tid == 0)
        ready_s = 0;
    const int r0 = k0 + CB * (int)blockIdx.x;
    const int prow = 16 * (wave - 4) + (lane & 15), cls = lane >> 4;
    const bool own = blockIdx.x > 0;
    {
        // wavefronts 0 .. 3: the diagonal tile (k0, k0); wavefronts 4 .. 7: this workgroup's tile (r0, k0).  Every global load
        // before the first wait, as in cholb_trail_kernel.
        const int w4 = wave & 3;
        const int rt = wave < 4 ? k0 : r0; // first row of the tile this wavefront updates
        const bool work = wave < 4 || own;
        double wv[16], ld[8], lr[8];
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r)
            {
                const int gi = rt + w4 * 16 + 4 * r + kk, gj = k0 + b * 16 + ii;
                const bool ok = work && gi < p && gj < p && gj <= gi;
                wv[b * 4 + r] = W[ok ? (size_t)gi * p + gj : 0];
            }
#pragma unroll
        for (int it = 0; it < 8; ++it)
        {
            const int e = tid + CBQ_T * it, r = e >> 6, c = e & 63;
            ld[it] = Lf[k0 + r < p ? (size_t)(k0 + r) * p + kp + c : 0];
            lr[it] = Lf[(own && r0 + r < p) ? (size_t)(r0 + r) * p + kp + c : 0];
        }
        if (tid < CB)
            yk_s[tid] = yv[kp + tid];
#pragma unroll
        for (int it = 0; it < 8; ++it)
        {
            const int e = tid + CBQ_T * it, r = e >> 6, c = e & 63;
            tD[r * CBT_LD + c] = k0 + r < p ? ld[it] : 0.0;
            tR[r * CBT_LD + c] = (own && r0 + r < p) ? lr[it] : 0.0;
        }
        __syncthreads();
        if (work)
        {
            const double *tI = wave < 4 ? tD : tR;
            cb_v4f64 acc[4];
#pragma unroll
            for (int b = 0; b < 4; ++b)
                acc[b] = (cb_v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
            for (int c = 0; c < 16; ++c)
            {
                const double va = tI[(w4 * 16 + ii) * CBT_LD + c * 4 + kk];
                double vb[4];
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    vb[b] = tD[(b * 16 + ii) * CBT_LD + c * 4 + kk];
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(va, vb[b], acc[b], 0, 0, 0);
            }
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                {
                    const int i = w4 * 16 + 4 * r + kk, j = b * 16 + ii; // element of the tile
                    const double u = wv[b * 4 + r] - acc[b][r];
                    if (wave < 4)
                        Lt[i * CB_LD + j] = (i < nb && j <= i) ? u : (i == j ? 1.0 : 0.0); // (identity beyond nb, zero above the diagonal)
                    else
                        tOut[i * CBT_LD + j] = (r0 + i < p && j < nb) ? u : 0.0;
                }
        }
        else if (wave == 4)
        {
            // workgroup 0 has no tile of its own: its fifth wavefront takes the right-hand side's slice through the update
            const int i = k0 + lane;
            double yi = 0.0;
            if (i < p)
                yi = yv[i] - cholb_row_dot(Lf + (size_t)i * p + kp, yk_s);
            yhead_s[lane] = yi;
        }
    }
    __syncthreads(); // (Lt, tOut and yhead_s complete; tD / tR are dead: the published columns P take their place)
    double rr[16];
#pragma unroll
    for (int q = 0; q < 16; ++q)
    {
        const int c = cls + 4 * q;
        const bool ok = wave >= 4 && own && r0 + prow < p && c < nb;
        rr[q] = ok ? tOut[prow * CBT_LD + c] : 0.0;
    }
    cholb_panel4_body(Lt, P, &ready_s, rr, Lf, p, k0, nb, r0, prow, cls, dorig, flag, yv, dinvg, yhead_s);
}

// rhs | diag of a solve in the KERNEL ARGUMENTS (p <= CBV_MAX): see cholb_init_arg_kernel
constexpr int CBV_MAX = 224;
struct CholbVecArg
{
    double v[2 * CBV_MAX]; // rhs[0 .. p) | diag[0 .. p)
};

// ---- round 5: 64 < p <= 128 in ONE launch ---------------------------------------------------------------------------------------
// Two panels: init, panel, step were three launches of 8 + 20 + 21 us for work whose dependent part is two chains of 64 pivots
// (11 us each).  Here one workgroup of nine wavefronts does all of it without going back to memory in between: W = J^T J +
// mu D^2 straight from J^T J and the kernel arguments (rhs | diag), panel 0 (wavefronts 0 - 3 factor, 4 - 7 solve rows 64 ..
// p - 1 a column behind, wavefront 8 takes the right-hand side through), L10 staged in LDS, the (1, 1) tile and the right-hand
// side's second slice updated from it (the tile arithmetic of cholb_trail_kernel / cholb_step_kernel), panel 1.  The same
// device functions on the same operands in the same order as the general path: same bits (GSLNLS_LARGE_SMALL_OFF=1 keeps the
// three launches).  L, 1 / L_jj and y are left where the back substitution and a caller's tail expect them.
constexpr int CBS_T = 576;
template <int J>
__device__ __forceinline__ void cholb_back_steps(double &v, double di, const double (&c)[CB], int lane); // (defined with cholb_backall_kernel)
__global__ __launch_bounds__(CBS_T) void cholb_small_kernel(const double *src, int p, CholbVecArg va, double mu, double *Lf, double *yv,
                                                            double *dinvg, int *flag, double *sol)
{
    __shared__ double Lt[CB * CB_LD];
    __shared__ __attribute__((aligned(16))) double stage[CB * CBT_LD];
    __shared__ CholbPub P;
    __shared__ double dor[2 * CB], ys[2 * CB];
    __shared__ int ready_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kk = lane >> 4, ii = lane & 15;
    const double *rhs = va.v, *dmp = va.v + p;
    const int nb2 = p - CB; // 1 .. 64 columns in the second block
    if (tid == 0)
    {
        ready_s = 0;
        *flag = 0;
    }
    for (int r = tid; r < 2 * CB; r += CBS_T)
    {
        const double a = src[r < p ? (size_t)r * p + r : 0], d = dmp[r < p ? r : 0];
        dor[r] = r < p ? __dadd_rn(a, __dmul_rn(__dmul_rn(mu, d), d)) : 0.0;
        ys[r] = r < p ? rhs[r] : 0.0;
    }
    {
        // the diagonal tile (0, 0), lower triangle; every load before the first wait
        constexpr int NL = (CB * CB + CBS_T - 1) / CBS_T;
        double dv[NL];
#pragma unroll
        for (int it = 0; it < NL; ++it)
        {
            const int e = tid + CBS_T * it, i = e >> 6, j = e & 63;
            dv[it] = src[(e < CB * CB && j <= i) ? (size_t)i * p + j : 0];
        }
#pragma unroll
        for (int it = 0; it < NL; ++it)
        {
            const int e = tid + CBS_T * it, i = e >> 6, j = e & 63;
            if (e < CB * CB)
            {
                double x = dv[it];
                if (i == j)
                    x = __dadd_rn(x, __dmul_rn(__dmul_rn(mu, dmp[i]), dmp[i]));
                Lt[i * CB_LD + j] = j <= i ? x : 0.0;
            }
        }
    }
    // rows 64 .. p - 1 of column block 0: thread (row = 16 (wave - 4) + lane % 16, class = lane / 16) of wavefronts 4 .. 7
    const int prow = 16 * (wave - 4) + (lane & 15), cls = lane >> 4;
    double rr[16];
#pragma unroll
    for (int q = 0; q < 16; ++q)
    {
        const int c = cls + 4 * q;
        const bool ok = wave >= 4 && wave < 8 && CB + prow < p;
        rr[q] = src[ok ? (size_t)(CB + prow) * p + c : 0];
        rr[q] = ok ? rr[q] : 0.0;
    }
    __syncthreads();
    // ---- panel 0 ----
    auto rhs_slice = [&](int k0, int nb) {
        // L y = b for the slice of this panel, a column behind the factorisation (wavefront 8)
        double v = lane < nb ? ys[k0 + lane] : 0.0;
        int seen = 0;
        for (int j = 0; j < nb; ++j)
        {
            while (seen <= j)
            {
                seen = __hip_atomic_load(&ready_s, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (seen <= j)
                    __builtin_amdgcn_s_sleep(1);
            }
            const double rs = P.invd[j];
            const double yj = wide_bcast(v, j) * rs;
            if (lane == j)
                v = yj;
            else if (lane > j && lane < nb)
                v -= __dmul_rn(P.V[j * CB + lane], rs) * yj;
        }
        if (lane < nb)
            ys[k0 + lane] = v;
    };
    auto factor = [&](int k0, int nb) -> bool {
        __builtin_amdgcn_s_setprio(3);
        bool bad;
        switch (wave)
        {
        case 0: bad = cholb_diag_wave<0>(Lt, P, &ready_s, dor, k0, nb, lane); break;
        case 1: bad = cholb_diag_wave<1>(Lt, P, &ready_s, dor, k0, nb, lane); break;
        case 2: bad = cholb_diag_wave<2>(Lt, P, &ready_s, dor, k0, nb, lane); break;
        default: bad = cholb_diag_wave<3>(Lt, P, &ready_s, dor, k0, nb, lane); break;
        }
        __builtin_amdgcn_s_setprio(0);
        return bad;
    };
    if (wave < 4)
    {
        if (factor(0, CB) && lane == 0)
            *flag = 1;
    }
    else if (wave < 8)
        cholb_trsm_steps_v<0>(rr, P, lane & 15, cls, &ready_s, 0);
    else
        rhs_slice(0, CB);
    __syncthreads();
    // ---- L11, L10 and 1 / L_jj out; L10 staged; what the (1, 1) tile starts from ----
    for (int e = tid; e < CB * CB; e += CBS_T)
    {
        const int i = e >> 6, j = e & 63;
        if (j <= i)
            Lf[(size_t)i * p + j] = i == j ? P.ldiag[j] : __dmul_rn(P.V[j * CB + i], P.invd[j]);
    }
    if (tid < CB)
        dinvg[tid] = P.invd[tid];
    if (wave >= 4 && wave < 8)
    {
#pragma unroll
        for (int q = 0; q < 16; ++q)
        {
            const int c = cls + 4 * q;
            stage[prow * CBT_LD + c] = rr[q]; // (zero for the rows behind p)
            if (CB + prow < p)
                Lf[(size_t)(CB + prow) * p + c] = rr[q];
        }
    }
    double wv[16];
    if (wave < 4)
    {
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r)
            {
                const int i = wave * 16 + 4 * r + kk, j = b * 16 + ii;
                const bool ok = i < nb2 && j <= i;
                double x = src[ok ? (size_t)(CB + i) * p + CB + j : 0];
                if (ok && i == j)
                    x = __dadd_rn(x, __dmul_rn(__dmul_rn(mu, dmp[CB + i]), dmp[CB + i]));
                wv[b * 4 + r] = x;
            }
    }
    __syncthreads();
    if (tid == 0)
        ready_s = 0;
    if (wave < 4)
    {
        cb_v4f64 acc[4];
#pragma unroll
        for (int b = 0; b < 4; ++b)
            acc[b] = (cb_v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
        for (int c = 0; c < 16; ++c)
        {
            const double vaa = stage[(wave * 16 + ii) * CBT_LD + c * 4 + kk];
            double vb[4];
#pragma unroll
            for (int b = 0; b < 4; ++b)
                vb[b] = stage[(b * 16 + ii) * CBT_LD + c * 4 + kk];
#pragma unroll
            for (int b = 0; b < 4; ++b)
                acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(vaa, vb[b], acc[b], 0, 0, 0);
        }
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r)
            {
                const int i = wave * 16 + 4 * r + kk, j = b * 16 + ii;
                const double u = wv[b * 4 + r] - acc[b][r];
                Lt[i * CB_LD + j] = (i < nb2 && j <= i) ? u : (i == j ? 1.0 : 0.0);
            }
    }
    else if (wave == 8)
    {
        // the right-hand side's second slice through the update: y[i] -= L[i][0 .. 63] . y[0 .. 63], the sums of cholb_row_dot
        double rv[CB];
#pragma unroll
        for (int c = 0; c < CB; ++c)
            rv[c] = stage[lane * CBT_LD + c];
        double s0 = 0.0;
#pragma unroll
        for (int c = 0; c < CB; ++c)
            s0 += rv[c] * ys[c];
        if (lane < nb2)
            ys[CB + lane] = ys[CB + lane] - s0;
    }
    __syncthreads();
    // ---- panel 1 ----
    if (wave < 4)
    {
        if (factor(CB, nb2) && lane == 0)
            *flag = 1;
    }
    else if (wave == 8)
        rhs_slice(CB, nb2);
    __syncthreads();
    for (int e = tid; e < CB * CB; e += CBS_T)
    {
        const int i = e >> 6, j = e & 63;
        if (i < nb2 && j <= i)
            Lf[(size_t)(CB + i) * p + CB + j] = i == j ? P.ldiag[j] : __dmul_rn(P.V[j * CB + i], P.invd[j]);
    }
    if (tid < nb2)
        dinvg[CB + tid] = P.invd[tid];
    for (int r = tid; r < p; r += CBS_T)
        yv[r] = ys[r];
    // ---- L^T x = y, two blocks, in this launch: the sums of cholb_backall_kernel (its pipelined walk: block 1's triangle,
    // then block 0's 64 components updated and its triangle solved by the wavefront whose turn it is) ----
    // Wavefront 6 takes block 1, wavefront 5 block 0; each holds its block's triangle c[r] = L[k0 + r][k0 + lane], r > lane, in
    // ONE register array (two arrays in this kernel spill its factoring wavefronts): block 1's from the published columns
    // (the product that was stored), block 0's from memory, requested while wavefront 6 solves.
    double c[CB], di = 0.0;
#pragma unroll
    for (int r = 0; r < CB; ++r)
        c[r] = 0.0;
    if (wave == 6)
    {
#pragma unroll
        for (int r = 0; r < CB; ++r)
            c[r] = (r < nb2 && lane < r) ? __dmul_rn(P.V[lane * CB + r], P.invd[lane]) : 0.0;
        di = lane < nb2 ? P.invd[lane] : 0.0;
    }
    else if (wave == 5)
    {
#pragma unroll
        for (int r = 0; r < CB; ++r)
            c[r] = Lf[lane < r ? (size_t)r * p + lane : 0];
#pragma unroll
        for (int r = 0; r < CB; ++r)
            c[r] = lane < r ? c[r] : 0.0;
        di = dinvg[lane];
    }
    if (wave == 6)
    {
        double v = lane < nb2 ? ys[CB + lane] : 0.0;
        cholb_back_steps<CB - 1>(v, di, c, lane);
        if (lane < nb2)
        {
            ys[CB + lane] = v;
            sol[CB + lane] = v;
        }
    }
    __syncthreads();
    if (wave == 5)
    {
        double s0 = 0.0;
#pragma unroll 16
        for (int r = 0; r < CB; ++r)
        {
            const double l = r < nb2 ? stage[r * CBT_LD + lane] : 0.0; // L[64 + r][lane]
            s0 += l * ys[CB + r];
        }
        double v = ys[lane] - s0;
        cholb_back_steps<CB - 1>(v, di, c, lane);
        sol[lane] = v;
    }
}

// One block of the back substitution L^T x = y (blocks from the last to the first): every workgroup solves the block's
// 64 x 64 transposed triangle for x_k (redundantly: same bits, no hand-off); workgroup 0 stores x_k, workgroup g > 0
// takes x_k out of 256 of the components in front of the block: y[j] -= sum_r L[k0 + r][j] x_r (coalesced over j).
__global__ __launch_bounds__(256) void cholb_back_kernel(const double *Lf, int p, int k0, double *yv, double *sol, const double *dinvg)
{
    __shared__ double Ld[CB * CB_LD], xk[CB];
    const int tid = threadIdx.x, nb = p - k0 < CB ? p - k0 : CB;
    for (int e = tid; e < nb * nb; e += 256)
    {
        const int i = e / nb, j = e - i * nb;
        Ld[i * CB_LD + j] = j <= i ? Lf[(size_t)(k0 + i) * p + k0 + j] : 0.0;
    }
    __syncthreads();
    if (tid < 64)
    {
        double v = tid < nb ? yv[k0 + tid] : 0.0;
        const double di = tid < nb ? dinvg[k0 + tid] : 0.0;
        for (int j = nb - 1; j >= 0; --j)
        {
            const double xj = wide_bcast(v, j) * wide_bcast(di, j);
            if (tid == j)
                v = xj;
            else if (tid < j)
                v -= Ld[j * CB_LD + tid] * xj;
        }
        if (tid < nb)
            xk[tid] = v;
    }
    __syncthreads();
    if (blockIdx.x == 0)
    {
        if (tid < nb)
            sol[k0 + tid] = xk[tid];
        return;
    }
    const int j = (blockIdx.x - 1) * 256 + tid;
    if (j < k0)
    {
        double s0 = 0.0;
        for (int r = 0; r < nb; ++r)
            s0 += Lf[(size_t)(k0 + r) * p + j] * xk[r];
        yv[j] -= s0;
    }
}

// The whole back substitution L^T x = y in ONE launch, by one workgroup of eight wavefronts (the per-block form above: a
// launch per block of 64, 19 us each -- 154 us of the 750 of a p = 500 solve).  y lives in LDS.  Block s from the end is
// solved by wavefront s % 8: lane i holds column i of the block's triangle in REGISTERS (asked for one turn ahead, so the
// load is never waited for) and the 64 steps are a recursion with literal register indices -- readlane, multiply, fma per
// step.  Then all 512 threads take x_k out of the components in front of the block (coalesced over j; the first 32 of the
// 64 rows are requested before the solve starts).  Same sums in the same order as cholb_back_kernel: not a bit changes.
constexpr int CBA_T = 512, CBA_W = CBA_T / 64;
template <int J>
__device__ __forceinline__ void cholb_back_steps(double &v, double di, const double (&c)[CB], int lane)
{
    if constexpr (J >= 0)
    {
        const double xj = wide_bcast(v, J) * wide_bcast(di, J);
        if (lane == J)
            v = xj;
        else if (lane < J)
            v -= c[J] * xj;
        cholb_back_steps<J - 1>(v, di, c, lane);
    }
}
// Round 5: blocks kb_lo .. kb_hi - 1 only (from the last to the first), and only the components of that segment are
// updated here (written back to yv for the segments in front); the components in front of the segment receive the
// segment's contributions from cholb_backupd_kernel, spread over many workgroups -- as ONE workgroup the update of all
// p^2 / 2 entries of L^T was 0.47 ms of a 1.55 ms solve at p = 2000.  Same sums in the same order whichever way it is cut.
__global__ __launch_bounds__(CBA_T) void cholb_backall_kernel(const double *Lf, int p, double *yv, double *sol,
                                                              const double *dinvg, int kb_lo, int kb_hi, int pipelined)
{
    extern __shared__ double ys_seg[]; // 64 (kb_hi - kb_lo) doubles: y of the segment (zeros behind p), block by block overwritten by x
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int jbase = kb_lo * CB;
    double *ys = ys_seg - jbase; // indexed by the component's own number
    const int nblk = kb_hi;      // (the blocks behind kb_hi are done: their x is in sol, their contributions are in yv)
    for (int j = jbase + tid; j < nblk * CB; j += CBA_T)
        ys[j] = j < p ? yv[j] : 0.0;
    double c[CB], di = 0.0;
    auto load_block = [&](int kb) {
        const int k0 = kb * CB, nb = p - k0 < CB ? p - k0 : CB;
#pragma unroll
        for (int r = 0; r < CB; ++r)
            c[r] = (r < nb && lane < r) ? Lf[(size_t)(k0 + r) * p + k0 + lane] : 0.0; // L[k0 + r][k0 + lane], r > lane
        di = lane < nb ? dinvg[k0 + lane] : 0.0;
    };
#pragma unroll
    for (int r = 0; r < CB; ++r)
        c[r] = 0.0;
    const int nseg = kb_hi - kb_lo;
    if (wave < nseg)
        load_block(nblk - 1 - wave);
    __syncthreads();
    constexpr int H = 16; // rows of L per round of loads in the update
    // y[j] -= sum_r L[k0 + r][j] x_r, r ascending, for one component j of the segment in front of block (k0, nb): the sums of
    // every form of this kernel
    auto take_out = [&](int j, int k0, int nb) -> double {
        double s0 = 0.0;
#pragma unroll 1
        for (int r0 = 0; r0 < CB; r0 += H)
        {
            double lv[H];
#pragma unroll
            for (int r = 0; r < H; ++r)
                lv[r] = r0 + r < nb ? Lf[(size_t)(k0 + r0 + r) * p + j] : 0.0;
#pragma unroll
            for (int r = 0; r < H; ++r)
                s0 += lv[r] * ys[k0 + r0 + r];
        }
        return s0;
    };
    if (pipelined)
    {
        // Round 5, end: the blocks one behind the other WITHOUT the workgroup waiting for each block's update.  While seven
        // wavefronts take x of block kb out of the components further in front, the eighth -- the one whose turn is next --
        // takes it out of the 64 components of block kb - 1 only and solves that block at once: a block costs its 64
        // components' update + its triangle (3 us) instead of triangle, barrier, the whole update, barrier (5.9 us).  Every
        // component receives the same sums in the same order.
        if (wave == 0)
        {
            const int kb = nblk - 1, k0 = kb * CB, nb = p - k0 < CB ? p - k0 : CB;
            double v = lane < nb ? ys[k0 + lane] : 0.0;
            cholb_back_steps<CB - 1>(v, di, c, lane);
            if (lane < nb)
            {
                ys[k0 + lane] = v;
                sol[k0 + lane] = v;
            }
            if (CBA_W < nseg)
                load_block(nblk - 1 - CBA_W);
        }
        __syncthreads();
        for (int s = 0; s + 1 < nseg; ++s)
        {
            const int kb = nblk - 1 - s, k0 = kb * CB, nb = p - k0 < CB ? p - k0 : CB; // x of this block is in ys
            const int kn0 = k0 - CB;                                                   // the next block: always a full one
            const int wn = (s + 1) % CBA_W;
            if (wave == wn)
            {
                double v = ys[kn0 + lane] - take_out(kn0 + lane, k0, nb);
                cholb_back_steps<CB - 1>(v, di, c, lane);
                ys[kn0 + lane] = v;
                sol[kn0 + lane] = v;
                if (s + 1 + CBA_W < nseg)
                    load_block(nblk - 1 - (s + 1 + CBA_W)); // this wavefront's next block: eight turns to arrive
            }
            else
            {
                const int rank = wave < wn ? wave : wave - 1; // 0 .. 6 among the seven
                for (int j = jbase + rank * 64 + lane; j < kn0; j += (CBA_W - 1) * 64)
                    ys[j] -= take_out(j, k0, nb);
            }
            __syncthreads();
        }
        return;
    }
    for (int s = 0; s < nseg; ++s)
    {
        const int kb = nblk - 1 - s, k0 = kb * CB, nb = p - k0 < CB ? p - k0 : CB;
        // the update's first 16 rows of L for this thread's first component: requested before the solve, used behind it
        const int j0 = jbase + tid;
        double lv[H];
#pragma unroll
        for (int r = 0; r < H; ++r)
            lv[r] = (j0 < k0 && r < nb) ? Lf[(size_t)(k0 + r) * p + j0] : 0.0;
        if (wave == (s % CBA_W))
        {
            double v = lane < nb ? ys[k0 + lane] : 0.0;
            cholb_back_steps<CB - 1>(v, di, c, lane);
            if (lane < nb)
            {
                ys[k0 + lane] = v;
                sol[k0 + lane] = v;
            }
            if (s + CBA_W < nseg)
                load_block(nblk - 1 - (s + CBA_W)); // this wavefront's next block: eight turns to arrive
        }
        __syncthreads();
        // y[j] -= sum_r L[k0 + r][j] x_r for the components of the segment in front of the block, r ascending
        for (int j = j0; j < k0; j += CBA_T)
        {
            double s0 = 0.0;
#pragma unroll 1
            for (int r0 = 0; r0 < CB; r0 += H)
            {
                if (r0 > 0 || j != j0)
                {
#pragma unroll
                    for (int r = 0; r < H; ++r)
                        lv[r] = r0 + r < nb ? Lf[(size_t)(k0 + r0 + r) * p + j] : 0.0;
                }
#pragma unroll
                for (int r = 0; r < H; ++r)
                    s0 += lv[r] * ys[k0 + r0 + r];
            }
            ys[j] -= s0;
        }
        __syncthreads();
    }
}

// y[j] -= sum_r L[k0 + r][j] x_r for the blocks kb_hi - 1 .. kb_lo (in that order: the order the one-workgroup kernel
// subtracts them in) and the components j in front of the segment: one thread per component, coalesced over j
__global__ __launch_bounds__(64) void cholb_backupd_kernel(const double *Lf, int p, double *yv, const double *sol, int kb_lo, int kb_hi)
{
    __shared__ double xs[CB];
    const int j = blockIdx.x * 64 + threadIdx.x, jend = kb_lo * CB;
    double yj = j < jend ? yv[j] : 0.0;
    for (int kb = kb_hi - 1; kb >= kb_lo; --kb)
    {
        const int k0 = kb * CB, nb = p - k0 < CB ? p - k0 : CB;
        __syncthreads();
        xs[threadIdx.x] = (int)threadIdx.x < nb ? sol[k0 + threadIdx.x] : 0.0;
        __syncthreads();
        constexpr int H = 16;
        double s0 = 0.0;
#pragma unroll 1
        for (int r0 = 0; r0 < CB; r0 += H)
        {
            double lv[H];
#pragma unroll
            for (int r = 0; r < H; ++r)
                lv[r] = (j < jend && r0 + r < nb) ? Lf[(size_t)(k0 + r0 + r) * p + j] : 0.0;
#pragma unroll
            for (int r = 0; r < H; ++r)
                s0 += lv[r] * xs[r0 + r];
        }
        yj -= s0;
    }
    if (j < jend)
        yv[j] = yj;
}

// The same update with the blocks of the segment side by side: part[t - kb_lo][j] = sum_r L[64 t + r][j] x_r (r ascending,
// as above) by workgroup (j chunk, t) -- the blocks' contributions do not depend on each other, only the ORDER in which
// they are subtracted from y_j is fixed -- and cholb_backsum_kernel subtracts them, last block first.  Same bits as
// cholb_backupd_kernel, (kb_hi - kb_lo) times the workgroups.
__global__ __launch_bounds__(64) void cholb_backpart_kernel(const double *Lf, int p, const double *sol, int kb_lo, double *part)
{
    __shared__ double xs[CB];
    const int j = blockIdx.x * 64 + threadIdx.x, jend = kb_lo * CB, kb = kb_lo + (int)blockIdx.y;
    const int k0 = kb * CB, nb = p - k0 < CB ? p - k0 : CB;
    xs[threadIdx.x] = (int)threadIdx.x < nb ? sol[k0 + threadIdx.x] : 0.0;
    __syncthreads();
    constexpr int H = 16;
    double s0 = 0.0;
#pragma unroll 1
    for (int r0 = 0; r0 < CB; r0 += H)
    {
        double lv[H];
#pragma unroll
        for (int r = 0; r < H; ++r)
            lv[r] = (j < jend && r0 + r < nb) ? Lf[(size_t)(k0 + r0 + r) * p + j] : 0.0;
#pragma unroll
        for (int r = 0; r < H; ++r)
            s0 += lv[r] * xs[r0 + r];
    }
    if (j < jend)
        part[(size_t)blockIdx.y * p + j] = s0;
}
__global__ __launch_bounds__(256) void cholb_backsum_kernel(int p, double *yv, int kb_lo, int kb_hi, const double *part)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= kb_lo * CB)
        return;
    double yj = yv[j];
    for (int kb = kb_hi - 1; kb >= kb_lo; --kb)
        yj -= part[(size_t)(kb - kb_lo) * p + j];
    yv[j] = yj;
}

// the flag of the natural-order factorisation, as a double behind the solution: one copy brings both down
__global__ void cholb_flag_kernel(const int *flag, double *dst)
{
    *dst = (double)*flag;
}

// Round 5: the end of a solve as the host sees it.  The solution and the flag go to pinned host memory the device writes
// through its mapping, then -- behind a system-scope fence -- the sequence number of the solve: the host polls that word
// instead of waiting in hipStreamSynchronize (measured on the driver's box: 5.2 ms of wall time for 1.7 ms of device time
// at p = 2000, most of it the wake-up of the waiting thread).
__global__ __launch_bounds__(256) void cholb_publish_kernel(const double *sol, const int *flag, int p, double *h_out, unsigned long long seq,
                                                            const double *extra, int extra_n)
{
    for (int j = threadIdx.x; j < p; j += 256)
        h_out[j] = sol[j];
    for (int j = threadIdx.x; j < extra_n; j += 256) // (what a tail computed behind the solve: MCholTail)
        h_out[p + 2 + j] = extra[j];
    if (threadIdx.x == 0)
        h_out[p] = (double)*flag;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0)
    {
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(h_out + p + 1), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Round 5: what the natural-order factorisation needs before its first panel, in ONE launch: the lower triangle of
// A = J^T J + mu D^2 in 64 x 64 tiles (the kernels above never read above the diagonal; mchol_init_kernel copies the full
// matrix with a 64-bit division per element and reads it twice -- 106 us of a 1.32 ms solve at p = 2000 -- and finds
// gamma and xi, which only the pivoted routine wants), the diagonal as the thresholds' reference, the right-hand side as
// the extra row, the flag cleared.  src == nullptr: the matrix was uploaded into W already (only the vectors).
__global__ __launch_bounds__(256) void cholb_init_kernel(const double *src, double *W, int p, const double *dmp, double mu, const double *rhs,
                                                         double *work, double *dorig, int *flag, int ntile)
{
    const int tid = threadIdx.x, t = blockIdx.x;
    if (t >= ntile)
    {
        for (int r = tid; r < p; r += 256)
        {
            dorig[r] = src ? __dadd_rn(src[(size_t)r * p + r], __dmul_rn(__dmul_rn(mu, dmp[r]), dmp[r])) : W[(size_t)r * p + r];
            work[r] = rhs[r];
        }
        if (tid == 0)
            *flag = 0;
        return;
    }
    int I = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while (I * (I + 1) / 2 > t)
        --I;
    while ((I + 1) * (I + 2) / 2 <= t)
        ++I;
    const int J = t - I * (I + 1) / 2;
    double v[16];
#pragma unroll
    for (int it = 0; it < 16; ++it)
    {
        const int e = tid + 256 * it, i = I * CB + (e >> 6), k = J * CB + (e & 63);
        v[it] = src[(i < p && k < p) ? (size_t)i * p + k : 0];
    }
#pragma unroll
    for (int it = 0; it < 16; ++it)
    {
        const int e = tid + 256 * it, i = I * CB + (e >> 6), k = J * CB + (e & 63);
        if (i < p && k <= i)
        {
            double x = v[it];
            if (i == k)
                x = __dadd_rn(x, __dmul_rn(__dmul_rn(mu, dmp[i]), dmp[i])); // (rounded like the host's A[i][i] += mu d d)
            W[(size_t)i * p + k] = x;
        }
    }
}

// cholb_init_kernel with rhs | diag in the KERNEL ARGUMENTS (round 5, p <= CBV_MAX): the 2 p doubles travel with the launch
// packet instead of being read by the kernel through the staging area's mapping -- a PCIe round trip in front of every
// damped solve of the matrix path (init 6 -> 12 us when the copy-engine upload was dropped; back to 6 with this)
__global__ __launch_bounds__(256) void cholb_init_arg_kernel(const double *src, double *W, int p, CholbVecArg va, double mu, double *work,
                                                             double *dorig, int *flag, int ntile)
{
    const int tid = threadIdx.x, t = blockIdx.x;
    const double *rhs = va.v, *dmp = va.v + p;
    if (t >= ntile)
    {
        for (int r = tid; r < p; r += 256)
        {
            dorig[r] = __dadd_rn(src[(size_t)r * p + r], __dmul_rn(__dmul_rn(mu, dmp[r]), dmp[r]));
            work[r] = rhs[r];
        }
        if (tid == 0)
            *flag = 0;
        return;
    }
    int I = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while (I * (I + 1) / 2 > t)
        --I;
    while ((I + 1) * (I + 2) / 2 <= t)
        ++I;
    const int J = t - I * (I + 1) / 2;
    double v[16];
#pragma unroll
    for (int it = 0; it < 16; ++it)
    {
        const int e = tid + 256 * it, i = I * CB + (e >> 6), k = J * CB + (e & 63);
        v[it] = src[(i < p && k < p) ? (size_t)i * p + k : 0];
    }
#pragma unroll
    for (int it = 0; it < 16; ++it)
    {
        const int e = tid + 256 * it, i = I * CB + (e >> 6), k = J * CB + (e & 63);
        if (i < p && k <= i)
        {
            double x = v[it];
            if (i == k)
                x = __dadd_rn(x, __dmul_rn(__dmul_rn(mu, dmp[i]), dmp[i])); // (rounded like the host's A[i][i] += mu d d)
            W[(size_t)i * p + k] = x;
        }
    }
}

struct MCholBuffers
{
    std::mutex mu;
    int cap = 0, device = -1;
    double *A = nullptr, *Lg = nullptr, *Cg = nullptr, *vec = nullptr; // vec: ainvg | dcur | b | dinv | scal | rhs | sol
    int *ivec = nullptr;                                              // pos | ord | flag of the natural-order factorisation
    double *stage = nullptr; // pinned, 3 cap + 8 doubles: [rhs | diag] on the way up, [sol | flag] on the way down
    double *stage_dev = nullptr; // its device address (mapped): the natural-order init kernel reads rhs | diag in place
    hipStream_t sq = nullptr;                // the stream of every copy and kernel of a solve
    hipStream_t sq2 = nullptr;               // round 5: the trailing updates behind the next panel's column run beside that panel
    hipEvent_t evp[2] = {nullptr, nullptr}, evr[2] = {nullptr, nullptr}; // panel k done / rest of trailing update k done (ping-pong)
    hipEvent_t evdone = nullptr;             // behind the last kernel of a solve: polled beside the sequence word
    double *down = nullptr, *down_dev = nullptr; // pinned + mapped: solution (cap) | flag | sequence number | a tail's extra (MC_EXTRA_MAX)
    unsigned long long seq = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr; // around the kernels of a solve: gslnls_debug_mchol_last_device_ms
    float last_device_ms = -1.f;
    bool attr_set = false;
};
static std::atomic<bool> g_mchol_timing{false}; // gslnls_debug_mchol_timing(1): the solves carry the event pair of gslnls_debug_mchol_last_device_ms
static MCholBuffers &mchol_buffers()
{
    static MCholBuffers *b = new MCholBuffers; // (never destroyed: no HIP calls during static teardown)
    return *b;
}

// 0, or a GSLNLS_E_* code when the device cannot take it (the caller keeps the host routine)
// A_host != nullptr: the matrix itself from the host.  Otherwise jtj_dev (p x p on the device, left untouched) with
// diag_host and mu: A = J^T J + mu D^2 is formed on the device -- the 8 p^2 bytes do not travel.
static int mchol_device_solve_impl(int p, const double *A_host, const double *jtj_dev, const double *diag_host, double mu,
                                   const double *rhs_host, double *sol_host, const MCholTail *tail = nullptr, int *tail_valid = nullptr)
{
    if (p < 1 || p > MC_PMAX || (tail && (tail->extra_n < 0 || tail->extra_n > MC_EXTRA_MAX)))
        return GSLNLS_E_UNSUPPORTED;
    if (tail_valid)
        *tail_valid = 0;
    const double t_entry = now_s();
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
        if (B.stage)
            (void)hipHostFree(B.stage);
        B.stage = nullptr;
        B.stage_dev = nullptr;
        if (B.down)
            (void)hipHostFree(B.down);
        B.down = B.down_dev = nullptr;
        (void)hipFree(B.ivec);
        B.A = B.Lg = B.Cg = B.vec = nullptr;
        B.ivec = nullptr;
        B.cap = 0;
        const size_t pp = (size_t)p * p;
        if (hipMalloc(&B.A, sizeof(double) * pp) != hipSuccess || hipMalloc(&B.Lg, sizeof(double) * pp) != hipSuccess ||
            hipMalloc(&B.Cg, sizeof(double) * (size_t)MC_NB_MAX * p) != hipSuccess ||
            hipMalloc(&B.vec, sizeof(double) * ((size_t)7 * p + MC_NB_MAX + 24)) != hipSuccess ||
            hipHostMalloc(&B.stage, sizeof(double) * ((size_t)3 * p + 8), hipHostMallocMapped) != hipSuccess ||
            hipHostGetDevicePointer((void **)&B.stage_dev, B.stage, 0) != hipSuccess ||
            hipHostMalloc(&B.down, sizeof(double) * ((size_t)p + 2 + MC_EXTRA_MAX), hipHostMallocMapped) != hipSuccess ||
            hipHostGetDevicePointer((void **)&B.down_dev, B.down, 0) != hipSuccess ||
            hipMalloc(&B.ivec, sizeof(int) * ((size_t)2 * p + 4)) != hipSuccess)
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
    // a stream of the library's own (non-blocking): on the null stream every launch is ordered against every blocking
    // stream of the process -- with another runtime user in it (PyTorch) the 70 launches of a p = 2000 solve took 5 ms
    // of host time for 1.7 ms of device work
    if (!B.sq && hipStreamCreateWithFlags(&B.sq, hipStreamNonBlocking) != hipSuccess)
    {
        (void)hipGetLastError();
        return GSLNLS_E_NODEVICE;
    }
    hipStream_t sq = B.sq;
    if (!B.sq2)
    {
        if (hipStreamCreateWithFlags(&B.sq2, hipStreamNonBlocking) != hipSuccess)
        {
            (void)hipGetLastError();
            return GSLNLS_E_NODEVICE;
        }
        for (int k = 0; k < 2; ++k)
            if (hipEventCreateWithFlags(&B.evp[k], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&B.evr[k], hipEventDisableTiming) != hipSuccess)
            {
                (void)hipGetLastError();
                return GSLNLS_E_NODEVICE;
            }
        if (hipEventCreateWithFlags(&B.evdone, hipEventDisableTiming) != hipSuccess)
        {
            (void)hipGetLastError();
            return GSLNLS_E_NODEVICE;
        }
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
    // rhs | diag (one copy up) | sol | flag as a double (one copy down) | work
    double *d_rhs = a.scal + 8, *d_dmp = d_rhs + p, *d_sol = d_dmp + p;
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
        GSLNLS_HIP_OK(hipMemcpy(B.A, A_host, sizeof(double) * (size_t)p * p, hipMemcpyHostToDevice)); // (pageable: the blocking form)
    // the two vectors go up in ONE asynchronous copy from the pinned staging area, the solution and the flag come down in
    // one: a solve synchronises with the device once, at its end (four blocking copies were 50 us of a 470 us solve)
    memcpy(B.stage, rhs_host, sizeof(double) * p);
    if (!A_host)
        memcpy(B.stage + p, diag_host, sizeof(double) * p);
    // With J^T J resident the natural-order init kernel reads the 2 p doubles where they are, through the staging area's
    // mapping (round 5: the copy was a copy-engine kernel of 3 us and a 6 us wait of the init kernel for it, in every trial
    // step of the matrix path -- profiles/r05_matrix_step_timeline.txt).  The pivoted routine and the debug entry that
    // brings the whole matrix from the host keep the copy.
    bool uploaded = false;
    auto upload = [&]() -> int {
        if (!uploaded)
            GSLNLS_HIP_OK(hipMemcpyAsync(d_rhs, B.stage, sizeof(double) * (size_t)(A_host ? p : 2 * p), hipMemcpyHostToDevice, sq));
        uploaded = true;
        return GSLNLS_SUCCESS;
    };
    const bool in_place = !A_host && B.stage_dev != nullptr && !getenv("GSLNLS_LARGE_UPLOAD_COPY");
    if (!in_place)
        if (const int e = upload())
            return e;
    // (the event pair of gslnls_debug_mchol_last_device_ms only once somebody has asked for it: each record is a marker
    // packet the kernels behind it wait for -- 6 us between the back substitution and a caller's tail, measured in the
    // kernel timeline of the matrix path, profiles/r05_matrix_step_timeline.txt)
    const bool timed = g_mchol_timing.load();
    if (timed && !B.ev0 && (hipEventCreate(&B.ev0) != hipSuccess || hipEventCreate(&B.ev1) != hipSuccess))
        B.ev0 = B.ev1 = nullptr;
    B.last_device_ms = -1.f;
    if (timed && B.ev0)
        (void)hipEventRecord(B.ev0, sq);
    auto pivoted_init = [&]() -> int {
        if (const int e = upload())
            return e;
        GSLNLS_HIP_OK(hipMemsetAsync(a.scal, 0, sizeof(double) * 8, sq));
        long long g = ((long long)p * p + 256 * 8 - 1) / (256 * 8);
        g = g < 1 ? 1 : (g > 1024 ? 1024 : g);
        hipLaunchKernelGGL(mchol_init_kernel, dim3((unsigned)g), dim3(256), 0, sq, a, d_rhs, A_host ? nullptr : jtj_dev, d_dmp, mu);
        return GSLNLS_SUCCESS;
    };
    // natural order first (level 3, no pivot search); the pivoted, modified factorisation below when it reports a pivot
    // that is not safely positive, or always under GSLNLS_LARGE_CHOL_PIVOTED=1
    {
        const char *pe = getenv("GSLNLS_LARGE_CHOL_PIVOTED");
        if (pe && atoi(pe) != 0)
        {
            if (const int e = pivoted_init())
                return e;
        }
        else
        {
            int *d_flag = B.ivec + 2 * p;
            double *d_work = d_sol + p + 8;
            // 64 < p <= 128: init, the two panels and the update between them in ONE launch (cholb_small_kernel); off with any
            // of the developer switches that select an older form of a step
            const bool small = in_place && p > CB && p <= 2 * CB && p <= CBV_MAX && !getenv("GSLNLS_LARGE_SMALL_OFF") &&
                               !getenv("GSLNLS_LARGE_STEP_V1") && !getenv("GSLNLS_LARGE_PANEL_V1") && !getenv("GSLNLS_LARGE_LOOKAHEAD") &&
                               !getenv("GSLNLS_LARGE_BACK_V1") && !getenv("GSLNLS_LARGE_BACK_BLOCKS") && !getenv("GSLNLS_LARGE_BACK_STEPWISE") &&
                               !getenv("GSLNLS_LARGE_BACKUPD_V1");
            // the lower triangle of A, the diagonal, the right-hand side as one more row of the matrix (L y = b happens
            // inside the factorisation), the flag: one launch
            if (!small)
            {
                const int nb64 = (p + CB - 1) / CB, ntile = A_host ? 0 : nb64 * (nb64 + 1) / 2;
                if (in_place && p <= CBV_MAX)
                {
                    CholbVecArg va;
                    memcpy(va.v, rhs_host, sizeof(double) * p);
                    memcpy(va.v + p, diag_host, sizeof(double) * p);
                    hipLaunchKernelGGL(cholb_init_arg_kernel, dim3(ntile + 1), dim3(256), 0, sq, jtj_dev, B.A, p, va, mu, d_work, a.dcur, d_flag, ntile);
                }
                else
                hipLaunchKernelGGL(cholb_init_kernel, dim3(ntile + 1), dim3(256), 0, sq, A_host ? nullptr : jtj_dev, B.A, p,
                                   in_place ? B.stage_dev + p : d_dmp, mu, in_place ? B.stage_dev : d_rhs,
                                   d_work, a.dcur, d_flag, ntile);
            }
            const bool panel_v1 = getenv("GSLNLS_LARGE_PANEL_V1") != nullptr; // (developer switch: the one-wavefront diagonal block, same bits)
            // GSLNLS_LARGE_LOOKAHEAD=1: the trailing update behind the next panel's column on a second stream, beside that
            // panel.  Built, bit-identical, and MEASURED SLOWER on this runtime (p = 500 / 1000 / 2000: 0.357 / 0.774 / 1.66 ms
            // against 0.291 / 0.607 / 1.33 in stream order, gpurun_out mchol_r05_b.txt): every cross-stream dependency is
            // a barrier packet whose latency exceeds the 7-10 us of trailing update it hides.  Off by default.
            const bool lookahead = getenv("GSLNLS_LARGE_LOOKAHEAD") != nullptr;
            const bool back_v1 = getenv("GSLNLS_LARGE_BACK_V1") != nullptr; // (developer switch: the one-workgroup back substitution, same bits)
            const bool back_pipe = getenv("GSLNLS_LARGE_BACK_STEPWISE") == nullptr; // (developer switch off: a block's update between two barriers, same bits)
            int nrest = 0; // launches on the second stream so far
            // round 5: from the second panel on, ONE launch per step -- the panel with its inputs updated on the fly beside
            // the rest of the previous panel's trailing update (cholb_step_kernel).  GSLNLS_LARGE_STEP_V1=1: panel and
            // trailing update as two launches (same bits); the developer switches above imply it.
            const bool step_v1 = getenv("GSLNLS_LARGE_STEP_V1") != nullptr || panel_v1 || lookahead;
            if (small)
            {
                CholbVecArg va;
                memcpy(va.v, rhs_host, sizeof(double) * p);
                memcpy(va.v + p, diag_host, sizeof(double) * p);
                hipLaunchKernelGGL(cholb_small_kernel, dim3(1), dim3(CBS_T), 0, sq, jtj_dev, p, va, mu, B.Lg, d_work, a.dinv, d_flag, d_sol);
            }
            else if (!step_v1)
            {
                int ncu = 256;
                (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
                for (int k0 = 0; k0 < p; k0 += CB)
                {
                    const int nrb = (p - k0 + CB - 1) / CB;
                    if (k0 == 0)
                        hipLaunchKernelGGL(cholb_panel4_kernel, dim3(nrb), dim3(CBQ_T), 0, sq, B.A, B.Lg, p, k0, a.dcur, d_flag, d_work, a.dinv);
                    else
                    {
                        const int ntile_rest = (nrb - 1) * nrb / 2;
                        int nrestwg = ntile_rest < 2 * ncu ? ntile_rest : 2 * ncu;
                        const int rows_behind = p - k0 - CB;
                        const int nrhs = rows_behind > 0 ? (rows_behind + CBQ_T - 1) / CBQ_T : 0;
                        hipLaunchKernelGGL(cholb_step_kernel, dim3(nrb + nrestwg + nrhs), dim3(CBQ_T), 0, sq, B.A, B.Lg, p, k0, a.dcur, d_flag,
                                           d_work, a.dinv, nrb, nrestwg, ntile_rest);
                    }
                }
            }
            else
            for (int k0 = 0, step = 0; k0 < p; k0 += CB, ++step)
            {
                const int nrb = (p - k0 + CB - 1) / CB; // row blocks from the diagonal block down
                // (workgroup 0: the diagonal block and the right-hand side; workgroup b: row block b of the panel)
                if (panel_v1)
                    hipLaunchKernelGGL(cholb_panel_kernel, dim3(nrb), dim3(CBP_T), 0, sq, B.A, B.Lg, p, k0, a.dcur, d_flag, d_work, a.dinv);
                else
                    hipLaunchKernelGGL(cholb_panel4_kernel, dim3(nrb), dim3(CBQ_T), 0, sq, B.A, B.Lg, p, k0, a.dcur, d_flag, d_work, a.dinv);
                if (nrb > 1)
                {
                    const int nt = (nrb - 1) * nrb / 2, nrhs = (p - k0 - CB + 255) / 256;
                    if (!lookahead || nrb == 2)
                    {
                        if (nrest > 0) // (the last of the second stream's updates reaches into this tile)
                            GSLNLS_HIP_OK(hipStreamWaitEvent(sq, B.evr[(nrest - 1) & 1], 0));
                        hipLaunchKernelGGL(cholb_trail_kernel, dim3(nt + nrhs), dim3(256), 0, sq, B.A, B.Lg, p, k0, nt, d_work, 0);
                        nrest = 0;
                    }
                    else
                    {
                        // The next panel only needs its own column block (and the right-hand side) updated: those nrb - 1
                        // tiles stay on this stream; the tiles behind them go to the second stream, which they share with
                        // nothing but the same tiles of the steps before and after -- beside the next panel kernel, whose
                        // 20 us are the latency of one wavefront while the other 250 CUs are idle.
                        GSLNLS_HIP_OK(hipEventRecord(B.evp[step & 1], sq)); // panel `step` is complete
                        GSLNLS_HIP_OK(hipStreamWaitEvent(B.sq2, B.evp[step & 1], 0));
                        const int nt2 = (nrb - 2) * (nrb - 1) / 2;
                        hipLaunchKernelGGL(cholb_trail_kernel, dim3(nt2), dim3(256), 0, B.sq2, B.A, B.Lg, p, k0, nt2, d_work, 2);
                        GSLNLS_HIP_OK(hipEventRecord(B.evr[nrest & 1], B.sq2));
                        // this step's tiles of the next column block come behind the previous step's update of them
                        if (nrest > 0)
                            GSLNLS_HIP_OK(hipStreamWaitEvent(sq, B.evr[(nrest - 1) & 1], 0));
                        hipLaunchKernelGGL(cholb_trail_kernel, dim3(nrb - 1 + nrhs), dim3(256), 0, sq, B.A, B.Lg, p, k0, nrb - 1, d_work, 1);
                        nrest += 1;
                    }
                }
            }
            if (small)
                ; // (the two blocks' back substitution ran at the end of cholb_small_kernel)
            else if (getenv("GSLNLS_LARGE_BACK_BLOCKS")) // (developer switch: the launch-per-block form, same bits)
                for (int k0 = ((p - 1) / CB) * CB; k0 >= 0; k0 -= CB)
                    hipLaunchKernelGGL(cholb_back_kernel, dim3(1 + (k0 + 255) / 256), dim3(256), 0, sq, B.Lg, p, k0, d_work, d_sol, a.dinv);
            else
            {
                // segments of the blocks, from the last to the first: one workgroup walks a segment, then every workgroup
                // the device has takes the segment's x out of the components in front of it
                const int nblk = (p + CB - 1) / CB;
                // (measured, gpurun_out mchol_r05_d.txt: up to 8 blocks one segment is fastest -- p = 500: 0.260 ms against 0.265
                // cut in two --, from 16 blocks on the cut pays: p = 1000 0.536 against 0.577, p = 2000 1.16 against 1.39)
                // (measured again with the pipelined walk, gpurun_out r05: 22 blocks 0.740 ms cut in six, 0.715 in eleven; 32 blocks
                // 1.080 / 1.077 / 1.088 at 8 / 11 / 16; 64 blocks 2.750 / 2.743 / 2.752 / 2.908 at 8 / 11 / 16 / 32)
                int seg = back_v1 ? nblk : (nblk <= 8 ? nblk : (nblk <= 16 ? (nblk + 1) / 2 : 11));
                if (const char *e = getenv("GSLNLS_LARGE_BACK_SEG")) // (developer switch: blocks per segment)
                    seg = atoi(e) > 0 ? (atoi(e) < nblk ? atoi(e) : nblk) : seg;
                for (int hi = nblk; hi > 0; hi -= seg)
                {
                    const int lo = hi - seg > 0 ? hi - seg : 0;
                    hipLaunchKernelGGL(cholb_backall_kernel, dim3(1), dim3(CBA_T), sizeof(double) * (size_t)CB * (hi - lo), sq, B.Lg, p, d_work,
                                       d_sol, a.dinv, lo, hi, back_pipe ? 1 : 0);
                    if (lo > 0 && getenv("GSLNLS_LARGE_BACKUPD_V1")) // (developer switch: the blocks of a segment one after the other, same bits)
                        hipLaunchKernelGGL(cholb_backupd_kernel, dim3((lo * CB + 63) / 64), dim3(64), 0, sq, B.Lg, p, d_work, d_sol, lo, hi);
                    else if (lo > 0)
                    {
                        // (B.Cg: MC_NB_MAX x p doubles of the pivoted routine, idle here; a segment has at most 16 blocks)
                        hipLaunchKernelGGL(cholb_backpart_kernel, dim3((lo * CB + 63) / 64, hi - lo), dim3(64), 0, sq, B.Lg, p, d_sol, lo, B.Cg);
                        hipLaunchKernelGGL(cholb_backsum_kernel, dim3((lo * CB + 255) / 256), dim3(256), 0, sq, p, d_work, lo, hi, B.Cg);
                    }
                }
            }
            if (timed && B.ev0)
                (void)hipEventRecord(B.ev1, sq);
            B.seq += 1;
            B.down[p + 1] = 0.0; // (the word the device is about to write; any value but the new sequence number)
            if (tail && tail->enqueue)
                tail->enqueue(tail->ctx, (void *)sq, d_sol); // (the caller's kernels, behind the back substitution on this stream)
            if (tail && tail->enqueue_factor)
                tail->enqueue_factor(tail->ctx, (void *)sq, B.Lg, a.dinv, p);
            hipLaunchKernelGGL(cholb_publish_kernel, dim3(1), dim3(256), 0, sq, d_sol, d_flag, p, B.down_dev, B.seq,
                               tail ? tail->extra_dev : nullptr, tail ? tail->extra_n : 0);
            const double t_enq = now_s();
            {
                // the device writes the sequence number behind the solution: poll it; the stream is asked now and then only to
                // notice a launch failure or a device fault, which would never write the word (an event recorded behind the
                // last kernel and queried in every turn until round 5: a marker packet and a runtime call per poll)
                volatile unsigned long long *word = reinterpret_cast<volatile unsigned long long *>(B.down + p + 1);
                for (unsigned spin = 1;; ++spin)
                {
                    if (*word == B.seq)
                        break;
                    if (spin & 255u)
                    {
                        __builtin_ia32_pause();
                        continue;
                    }
                    const hipError_t q = hipStreamQuery(sq);
                    if (q == hipSuccess)
                    {
                        if (*word != B.seq)
                            GSLNLS_HIP_OK(hipStreamSynchronize(sq)); // (the write is on its way: the stream's end covers it)
                        break;
                    }
                    if (q != hipErrorNotReady)
                    {
                        (void)hipGetLastError();
                        return GSLNLS_E_NODEVICE;
                    }
                }
                __sync_synchronize();
            }
            double *h_down = B.down;
            if (getenv("GSLNLS_LARGE_PROF"))
                fprintf(stderr, "[mchol] p = %d: enqueue %.3f ms, wait %.3f ms\n", p, 1e3 * (t_enq - t_entry), 1e3 * (now_s() - t_enq));
            B.last_device_ms = timed ? -2.f : -1.f; // (read on demand: gslnls_debug_mchol_last_device_ms -- no event wait inside a solve)
            if (h_down[p] == 0.0)
            {
                memcpy(sol_host, h_down, sizeof(double) * p);
                if (tail && tail->extra_n > 0 && tail->extra_host)
                    memcpy(tail->extra_host, h_down + p + 2, sizeof(double) * (size_t)tail->extra_n);
                if (tail_valid)
                    *tail_valid = tail != nullptr;
                GSLNLS_HIP_OK(hipGetLastError());
                return GSLNLS_SUCCESS;
            }
            if (getenv("GSLNLS_LARGE_PROF"))
                fprintf(stderr, "[mchol] p = %d: the natural-order factorisation met a pivot that is not safely positive; the pivoted routine runs\n", p);
            // not numerically positive definite: the matrix (overwritten by the trailing updates) is formed again
            if (A_host)
                GSLNLS_HIP_OK(hipMemcpy(B.A, A_host, sizeof(double) * (size_t)p * p, hipMemcpyHostToDevice)); // (pageable: the blocking form)
            if (const int e = pivoted_init())
                return e;
        }
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
        hipLaunchKernelGGL(mchol_panel_kernel, dim3(1), dim3(T), lds, sq, a);
        if (kb + a.nb < p)
            hipLaunchKernelGGL(mchol_trail_kernel, dim3(tiles, tiles), dim3(256), 0, sq, a);
    }
    {
        const size_t bl = sizeof(double) * ((size_t)p + MC_BS * (MC_BS + 1) + MC_BS) + sizeof(int) * ((size_t)p + MC_BS) + 64;
        hipLaunchKernelGGL(mchol_backsub_kernel, dim3(1), dim3(T), bl, sq, a, d_sol);
    }
    GSLNLS_HIP_OK(hipMemcpyAsync(sol_host, d_sol, sizeof(double) * p, hipMemcpyDeviceToHost, sq));
    GSLNLS_HIP_OK(hipStreamSynchronize(sq));
    GSLNLS_HIP_OK(hipGetLastError());
    return GSLNLS_SUCCESS;
}

// s_i = sum_j A[j][i] v_j, j ascending, product and sum rounded separately: for the symmetric J^T J (sp_jtj_kernel forms
// both triangles by the same sums) this is the host's row walk sum_j A[i][j] v_j bit for bit, read coalesced.
// Round 5: a workgroup owns 16 columns i; 256 rows j at a time come in through LDS -- every thread has 16 loads in flight,
// the next 256 rows are requested before the current ones are added -- and one thread per column adds them in order.  (One
// thread per column reading its p entries eight at a time was p / 8 dependent round trips: 24.7 us at p = 500, behind every
// lm step's solve.)
constexpr int SYMV_C = 16, SYMV_R = 256;
__global__ __launch_bounds__(256) void mchol_symv_kernel(const double *A, const double *v, int p, double *s)
{
    __shared__ double tile[SYMV_R * (SYMV_C + 1)];
    __shared__ double vs[SYMV_R];
    const int tid = threadIdx.x, ii = tid & 15, jj = tid >> 4, i0 = blockIdx.x * SYMV_C;
    double la[16], lv = 0.0;
    auto fetch = [&](int j0) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
        {
            const int j = j0 + jj + 16 * u;
            la[u] = A[(j < p && i0 + ii < p) ? (size_t)j * p + i0 + ii : 0];
        }
        lv = v[j0 + tid < p ? j0 + tid : 0];
    };
    fetch(0);
    double acc = 0.0;
    for (int j0 = 0; j0 < p; j0 += SYMV_R)
    {
        __syncthreads(); // the rows before have been added
#pragma unroll
        for (int u = 0; u < 16; ++u)
            tile[(jj + 16 * u) * (SYMV_C + 1) + ii] = la[u];
        vs[tid] = lv;
        __syncthreads();
        if (j0 + SYMV_R < p)
            fetch(j0 + SYMV_R);
        if (tid < SYMV_C)
        {
            const int nj = p - j0 < SYMV_R ? p - j0 : SYMV_R;
            for (int j = 0; j < nj; ++j)
                acc = __dadd_rn(acc, __dmul_rn(tile[j * (SYMV_C + 1) + tid], vs[j]));
        }
    }
    if (tid < SYMV_C && i0 + tid < p)
        s[i0 + tid] = acc;
}

int mchol_device_symv(int p, const double *jtj_dev, const double *v_host, double *s_host)
{
    if (p < 1 || !jtj_dev || !v_host || !s_host)
        return GSLNLS_EINVAL;
    MCholBuffers &B = mchol_buffers();
    std::lock_guard<std::mutex> lock(B.mu);
    int dev = 0;
    GSLNLS_HIP_OK(hipGetDevice(&dev));
    if (B.cap < p || B.device != dev || !B.sq || !B.stage)
        return GSLNLS_E_UNSUPPORTED; // (the buffers are the solve's: it runs first)
    // the solve's vectors are free between solves: v where its rhs goes, s where its solution goes
    double *d_v = B.vec + MC_NB_MAX + 3 * (size_t)B.cap + 8, *d_s = d_v + 2 * (size_t)B.cap;
    memcpy(B.stage, v_host, sizeof(double) * p);
    GSLNLS_HIP_OK(hipMemcpyAsync(d_v, B.stage, sizeof(double) * p, hipMemcpyHostToDevice, B.sq));
    hipLaunchKernelGGL(mchol_symv_kernel, dim3((p + SYMV_C - 1) / SYMV_C), dim3(256), 0, B.sq, jtj_dev, d_v, p, d_s);
    double *h_down = B.stage + 2 * (size_t)B.cap;
    GSLNLS_HIP_OK(hipMemcpyAsync(h_down, d_s, sizeof(double) * p, hipMemcpyDeviceToHost, B.sq));
    GSLNLS_HIP_OK(hipStreamSynchronize(B.sq));
    memcpy(s_host, h_down, sizeof(double) * p);
    return GSLNLS_SUCCESS;
}

// The damped solve and, behind it in the same submission, the row sums s = (J^T J) v of the predicted reduction for v = the
// solution (mchol_symv_kernel on the device's copy of it: the bits the host receives) -- they come home with the solution:
// no second upload / kernel / download / synchronisation per trial step (round 5: 35 us of every lm step at p = 500).
// *rows_valid = 0 when the pivoted routine produced the solution (the rows belong to a discarded one).
int mchol_device_solve_resident_symv(int p, const double *jtj_dev, const double *diag_host, double mu, const double *rhs_host,
                                     double *sol_host, double *rows_host, int *rows_valid)
{
    if (!jtj_dev || !diag_host || !rows_host || !rows_valid)
        return GSLNLS_EINVAL;
    if (p > MC_EXTRA_MAX)
        return GSLNLS_E_UNSUPPORTED;
    struct Ctx
    {
        MCholTail tail;
        const double *jtj;
        int p;
    } cx;
    cx.jtj = jtj_dev;
    cx.p = p;
    cx.tail.ctx = &cx;
    cx.tail.extra_n = p;
    cx.tail.extra_host = rows_host;
    cx.tail.enqueue = [](void *ctx, void *stream, const double *d_sol) {
        Ctx &c = *static_cast<Ctx *>(ctx);
        MCholBuffers &B = mchol_buffers(); // (the caller holds its lock; the buffers are final for this solve)
        double *d_s = B.Cg;                // (the back substitution's partial sums: idle behind it)
        hipLaunchKernelGGL(mchol_symv_kernel, dim3((c.p + SYMV_C - 1) / SYMV_C), dim3(256), 0, (hipStream_t)stream, c.jtj, d_sol, c.p, d_s);
        c.tail.extra_dev = d_s;
    };
    return mchol_device_solve_impl(p, nullptr, jtj_dev, diag_host, mu, rhs_host, sol_host, &cx.tail, rows_valid);
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
int mchol_device_solve_resident_tail(int p, const double *jtj_dev, const double *diag_host, double mu, const double *rhs_host,
                                     double *sol_host, const MCholTail *tail, int *tail_valid)
{
    if (!jtj_dev || !diag_host)
        return GSLNLS_EINVAL;
    return mchol_device_solve_impl(p, nullptr, jtj_dev, diag_host, mu, rhs_host, sol_host, tail, tail_valid);
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

// test / measurement hooks: device memory through the runtime this library is linked against (a process may hold a second
// copy of the HIP runtime -- PyTorch ships one -- whose allocations belong to another context)
extern "C" int gslnls_debug_device_alloc(void **p, size_t bytes)
{
    if (!p)
        return GSLNLS_EINVAL;
    *p = nullptr;
    if (hipMalloc(p, bytes ? bytes : 8) != hipSuccess)
    {
        (void)hipGetLastError();
        return GSLNLS_E_NODEVICE;
    }
    return GSLNLS_SUCCESS;
}
extern "C" int gslnls_debug_device_free(void *p)
{
    return (!p || hipFree(p) == hipSuccess) ? GSLNLS_SUCCESS : GSLNLS_E_NODEVICE;
}
extern "C" int gslnls_debug_device_copy(void *dst, const void *src, size_t bytes, int to_device)
{
    if (!dst || !src)
        return GSLNLS_EINVAL;
    if (hipMemcpy(dst, src, bytes, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost) != hipSuccess)
    {
        (void)hipGetLastError();
        return GSLNLS_E_NODEVICE;
    }
    return GSLNLS_SUCCESS;
}

extern "C" void gslnls_debug_mchol_timing(int on) { gslnls::g_mchol_timing.store(on != 0); }

// milliseconds between the upload of the vectors and the download of the solution of the LAST natural-order solve of this
// process (HIP events around its kernels: what the device did, whatever the host was busy with); < 0: not available
extern "C" double gslnls_debug_mchol_last_device_ms(void)
{
    gslnls::MCholBuffers &B = gslnls::mchol_buffers();
    std::lock_guard<std::mutex> lock(B.mu);
    if (B.last_device_ms == -2.f) // the last solve left its two events for whoever asks
    {
        if (!B.ev0 || hipEventSynchronize(B.ev1) != hipSuccess || hipEventElapsedTime(&B.last_device_ms, B.ev0, B.ev1) != hipSuccess)
            B.last_device_ms = -1.f;
    }
    return (double)B.last_device_ms;
}

// test / measurement hook: the same solve with J^T J already in device memory (p x p, row-major, not modified), the way
// the lm step of the large path calls it -- what goes up per solve is diag and rhs, 2 p doubles
extern "C" int gslnls_debug_mchol_solve_resident(int p, const double *jtj_dev, const double *diag, double mu, const double *rhs,
                                                 double *sol)
{
    if (p < 1 || !jtj_dev || !diag || !rhs || !sol)
        return GSLNLS_EINVAL;
    return gslnls::mchol_device_solve_resident(p, jtj_dev, diag, mu, rhs, sol);
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
