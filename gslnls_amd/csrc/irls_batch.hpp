// irls_batch.hpp -- workgroup-per-dataset robust fits (BASELINE config C5: thousands of independent
// data sets, each n ~ 1e4 rows, p = 8, loss = "bisquare").
//
// The reference has no batched form: a user would call gsl_nls(loss = ...) once per data set, and each
// call runs gsl_multifit_nlinear_rho_driver (src/nls_irls.c:412-546): cold LM solve from the original
// start -> unweighted residuals -> sigma = 1.4826 median|r| (full sort) -> new weights -> repeat.
// Here ONE workgroup owns one data set for the whole procedure and never leaves the kernel:
//   * LM loop: every thread streams its rows (x, y, sqrt w from L2/Infinity Cache), workgroup
//     reduction through LDS (fixed order), wavefront 0 runs the same lm_advance() as every other path
//     and broadcasts the next trial point through LDS -- the workgroup barrier replaces the kernel
//     boundary of the grid-per-fit path;
//   * re-weighting: |r_i| bit patterns to a scratch array, 8-pass radix SELECT with an LDS histogram
//     for the median (two order statistics when n is even), psi(r/sigma)/(r/sigma) weights normalised
//     to sum n, sqrt into the weight array of the next solve;
//   * stopping rule test_delta_irls (src/nls_irls.c:343-362) by wavefront 0.
// Data sets are independent => sharding over GPUs is a contiguous split of the batch, no collective.
#pragma once
#include <hip/hip_runtime.h>
#include "dense_kernels.hpp"
#include "irls_core.hpp"

namespace gslnls
{

template <int P>
struct IrlsBatchArgs
{
    const double *x;      // [B][NX][n]
    const double *y;      // [B][n]
    const double *usw;    // [B][n] sqrt(user weights) or nullptr
    double *sw;           // [B][n] scratch: sqrt weights of the current solve
    unsigned long long *keys; // [B][n] scratch: bits of |r_i|
    int n, lo, hi;        // rows per data set; data sets [lo, hi) of the batch
    double start[P];
    double lu[2 * P];
    int has_lu;
    LmParams prm;
    LossCfg loss;
    int irls_maxiter;
    double irls_xtol;
    // outputs, indexed by data set
    double *par;          // [B][P]
    double *scal;         // [B][4]: sigma, ssr (weighted), irls_tol, chisq_init
    int *ints;            // [B][4]: conv, irls_status, irls_niter, niter
};

template <class M, int JAC, int T>
__global__ __launch_bounds__(T) void irls_batch_kernel(IrlsBatchArgs<M::P> a)
{
    constexpr int P = M::P, NX = M::NX;
    using Sums = PassSums<P>;
    constexpr int NV = Sums::NV, NW = T / 64;
    __shared__ double lds_red[NW * NV];
    __shared__ double lds_tot[NV];
    __shared__ LmState<P> lds_state;
    __shared__ unsigned int hist[256];
    __shared__ unsigned long long sel_prefix, sel_mask, sel_k;
    __shared__ double sh_val[2], sh_sigma, sh_scale;
    __shared__ int sh_flag;

    const int d = a.lo + blockIdx.x;
    if (d >= a.hi)
        return;
    const int n = a.n, tid = threadIdx.x;
    const double *xd = a.x + (size_t)d * NX * n;
    const double *yd = a.y + (size_t)d * n;
    const double *ud = a.usw ? a.usw + (size_t)d * n : nullptr;
    double *swd = a.sw + (size_t)d * n;
    unsigned long long *kd = a.keys + (size_t)d * n;

    for (int i = tid; i < n; i += T)
        swd[i] = ud ? ud[i] : 1.0;
    __syncthreads();

    double workp[P];
#pragma unroll
    for (int k = 0; k < P; ++k)
        workp[k] = a.start[k];
    int irls_iter = 0, irls_status = ST_FAILURE, status = ST_CONTINUE;
    double chisq_carry = NAN, chisq_init = NAN, sigma = 1.0;
    LmParams prm = a.prm;
    prm.has_weights = 1;

    for (;;)
    {
        irls_iter += 1;
        // ---------------- cold LM solve from the original start with the current weights ----------------
        if (tid == 0)
        {
            LmState<P> s;
            lm_state_reset<P>(s, a.start, a.has_lu ? a.lu : nullptr);
            lds_state = s;
        }
        prm.chisq_in = (irls_iter > 1) ? chisq_carry : NAN;
        __syncthreads();
        for (int guard = 0; guard < 1000000; ++guard)
        {
            const int phase = lds_state.phase;
            if (phase == PH_DONE)
                break;
            double th[P], vel[P], delta[P];
#pragma unroll
            for (int k = 0; k < P; ++k)
            {
                th[k] = (phase == PH_FVV) ? lds_state.x[k] : lds_state.xt[k];
                vel[k] = lds_state.vel[k];
            }
            fd_deltas<P>(th, prm.h_df, delta);
            Sums acc;
            pass_zero<P>(acc);
            for (int i = tid; i < n; i += T)
            {
                double xr[NX];
#pragma unroll
                for (int c = 0; c < NX; ++c)
                    xr[c] = xd[(size_t)c * n + i];
                double Jrow[P];
                if (phase == PH_FVV)
                {
                    const double fv = row_fvv<M, JAC>(th, vel, delta, prm.h_fvv, prm.fvv_analytic != 0, xr, yd[i],
                                                      swd[i], Jrow, &acc.badj);
#pragma unroll
                    for (int k = 0; k < P; ++k)
                        acc.g[k] += Jrow[k] * fv;
                }
                else
                {
                    const double f = row_fj<M, JAC>(th, delta, xr, yd[i], swd[i], Jrow, &acc.badj);
                    acc_fj<P>(acc, f, Jrow);
                }
            }
            const double tot = block_sum_slots<NV, T>(reinterpret_cast<const double *>(&acc), lds_red);
            if (tid < NV)
                lds_tot[tid] = tot;
            __syncthreads();
            if (tid < 64)
            {
                LmState<P> s = lds_state;
                Sums r;
                double *rf = reinterpret_cast<double *>(&r);
#pragma unroll
                for (int v = 0; v < NV; ++v)
                    rf[v] = lds_tot[v];
                lm_advance<P>(s, r, prm);
                if (tid == 0)
                    lds_state = s;
            }
            __syncthreads();
        }
        status = lds_state.status;
        if (irls_iter == 1)
            chisq_init = lds_state.chisq_init;
        chisq_carry = lds_state.chisq1;
        if (status == ST_EBADFUNC || (status == ST_ENOPROG && irls_iter == 1))
            break;

        // ---------------- re-weighting ----------------
        double th[P];
#pragma unroll
        for (int k = 0; k < P; ++k)
            th[k] = lds_state.x[k];
        for (int i = tid; i < n; i += T)
        {
            double xr[NX];
#pragma unroll
            for (int c = 0; c < NX; ++c)
                xr[c] = xd[(size_t)c * n + i];
            kd[i] = (unsigned long long)__double_as_longlong(fabs(row_resid<M>(th, xr, yd[i], 1.0)));
        }
        __syncthreads();
        const unsigned long long k_lo = (unsigned long long)((n - 1) / 2), k_hi = (unsigned long long)(n / 2);
        const int nsel = (k_lo == k_hi) ? 1 : 2;
        for (int which = 0; which < nsel; ++which)
        {
            if (tid == 0)
            {
                sel_prefix = 0;
                sel_mask = 0;
                sel_k = which == 0 ? k_lo : k_hi;
            }
            for (int pass = 7; pass >= 0; --pass)
            {
                if (tid < 256)
                    hist[tid] = 0;
                __syncthreads();
                const unsigned long long prefix = sel_prefix, mask = sel_mask;
                for (int i = tid; i < n; i += T)
                {
                    const unsigned long long key = kd[i];
                    if ((key & mask) == prefix)
                        atomicAdd(&hist[(key >> (8 * pass)) & 255ull], 1u);
                }
                __syncthreads();
                if (tid == 0)
                {
                    unsigned long long k = sel_k, cum = 0;
                    int bin = 255;
                    for (int b = 0; b < 256; ++b)
                    {
                        const unsigned long long c = hist[b];
                        if (k < cum + c)
                        {
                            bin = b;
                            break;
                        }
                        cum += c;
                    }
                    sel_k = k - cum;
                    sel_prefix = prefix | ((unsigned long long)bin << (8 * pass));
                    sel_mask = mask | (255ull << (8 * pass));
                }
                __syncthreads();
            }
            if (tid == 0)
                sh_val[which] = __longlong_as_double((long long)sel_prefix);
            __syncthreads();
        }
        if (tid == 0)
            sh_sigma = 1.482602218505602 * (nsel == 1 ? sh_val[0] : (sh_val[0] + sh_val[1]) / 2.0);
        __syncthreads();
        sigma = sh_sigma;
        // raw weights (even in r: only |r| is needed), their sum in a fixed order, then the normalised sqrt
        double wsum = 0.0;
        for (int i = tid; i < n; i += T)
        {
            const double rs = __longlong_as_double((long long)kd[i]) / sigma;
            const double w = fmax(irls_psi(rs, a.loss) / rs, DBL_EPSILON);
            swd[i] = w;
            wsum += w;
        }
        wsum = wave_sum(wsum);
        if ((tid & 63) == 0)
            lds_red[tid >> 6] = wsum;
        __syncthreads();
        if (tid == 0)
        {
            double t = 0.0;
            for (int w = 0; w < NW; ++w)
                t += lds_red[w];
            sh_scale = (double)n / t;
            // test_delta_irls on wavefront 0's copy of the iterates
            int st = ST_CONTINUE;
            for (int k = 0; k < P; ++k)
            {
                const double xi = lds_state.x[k], dxi = fabs(workp[k] - xi);
                if (fmin(dxi / fabs(xi), dxi) < a.irls_xtol)
                    st = ST_SUCCESS;
                else
                {
                    st = ST_CONTINUE;
                    break;
                }
            }
            sh_flag = st;
        }
        __syncthreads();
        const double scale = sh_scale;
        irls_status = sh_flag;
        if (irls_status == ST_SUCCESS || irls_iter >= a.irls_maxiter)
            break;
        for (int i = tid; i < n; i += T)
        {
            double w = swd[i] * scale;
            if (ud)
                w = (ud[i] * ud[i]) * w;
            swd[i] = sqrt(w);
        }
#pragma unroll
        for (int k = 0; k < P; ++k)
            workp[k] = lds_state.x[k];
        __syncthreads();
    }

    if (tid == 0)
    {
        int conv = status;
        double irls_tol = 0.0;
        if (!(status == ST_EBADFUNC || (status == ST_ENOPROG && irls_iter == 1)))
        {
            if (irls_iter >= a.irls_maxiter && irls_status != ST_SUCCESS)
            {
                irls_status = ST_EMAXITER;
                conv = ST_EMAXITER;
            }
        }
        const bool ok = (conv == ST_SUCCESS || conv == ST_EMAXITER);
        for (int k = 0; k < P; ++k)
        {
            a.par[(size_t)d * P + k] = ok ? lds_state.x[k] : a.start[k];
            irls_tol = fmax(irls_tol, fabs(workp[k] - lds_state.x[k]));
        }
        a.scal[(size_t)d * 4 + 0] = sigma;
        a.scal[(size_t)d * 4 + 1] = lds_state.chisq1;
        a.scal[(size_t)d * 4 + 2] = irls_tol;
        a.scal[(size_t)d * 4 + 3] = chisq_init;
        a.ints[(size_t)d * 4 + 0] = conv;
        a.ints[(size_t)d * 4 + 1] = irls_status;
        a.ints[(size_t)d * 4 + 2] = irls_iter;
        a.ints[(size_t)d * 4 + 3] = lds_state.niter;
    }
}

} // namespace gslnls
