// dense_kernels.hpp -- grid-per-fit kernels: one nonlinear least-squares problem whose
// n residual rows are spread over the whole chip (BASELINE config C2: n = 1e6, p = 3).
//
// One launch of lm_step_kernel == one trial step of trust_iterate_lu_LD
// (src/trust.c:445-546), with everything the reference does between two model
// evaluations folded into the front of the next pass:
//
//   launch t:  [prologue, every workgroup]  reduce the G partial sums of launch t-1 in a
//              fixed order -> one wavefront runs lm_advance() (rho, accept/reject, mu,
//              D, modified-Cholesky solve, convergence test, bound projection) -> the new
//              trial point is broadcast through LDS;
//              [pass]  every thread streams its rows of (x, y[, sqrt w]) once, computes
//              f_i and the Jacobian row in registers (analytic, forward or central FD)
//              and accumulates ssr, J^T J, J^T f; wavefront shuffle reduction -> LDS
//              -> one partial set per workgroup.
//
// Nothing n-sized is written and nothing crosses PCIe inside the loop.  The kernel
// boundary is the grid-wide barrier (cheaper on gfx950 than an in-kernel barrier:
// MI355X_MICROARCH.md "boundary" 1.45 us vs "barrier-xcd" 4.1 us).  State and partials
// are double-buffered by launch parity so no workgroup reads what another one writes
// in the same launch.  All reductions have a fixed shape => results are run-to-run
// bit-identical.
#pragma once
#include <hip/hip_runtime.h>
#include "lm_core.hpp"
#include "models.hpp"
#include "rowops.hpp"

namespace gslnls
{

constexpr int NX_MAX = 4;

template <int P>
struct DenseCtx
{
    const double *x[NX_MAX]; // regressor columns, each n contiguous doubles
    const double *y;
    const double *sw; // sqrt(weights) or nullptr
    long long n;
    int G; // workgroups == partial sets
    int fresh_parity; // unused by kernels; host bookkeeping
    double *partials[2];  // [NV][G]
    LmState<P> *state[2]; // ping-pong by launch parity
    LmParams prm;
    double *ssrtrace; // maxiter+1, or nullptr
    double *partrace; // (maxiter+1) x P column-major, or nullptr
};

// wavefront (64 lanes) sum, result valid in every lane (butterfly, fixed order)
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
        v += __shfl_xor(v, off, 64);
    return v;
}

// Block-wide sum of NV values held per thread.  lds must hold (T/64) * NV doubles.
// After the call thread v (< NV) of wave 0 holds total v in the return value.
template <int NV, int T>
__device__ __forceinline__ double block_sum_slots(const double *vals, double *lds)
{
    constexpr int NW = T / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int v = 0; v < NV; ++v)
    {
        const double s = wave_sum(vals[v]);
        if (lane == 0)
            lds[wave * NV + v] = s;
    }
    __syncthreads();
    double tot = 0.0;
    if (threadIdx.x < NV)
    {
#pragma unroll
        for (int w = 0; w < NW; ++w)
            tot += lds[w * NV + threadIdx.x];
    }
    return tot;
}

template <class M, int JAC, int T>
__global__ __launch_bounds__(T) void lm_step_kernel(DenseCtx<M::P> ctx, int parity)
{
    constexpr int P = M::P;
    constexpr int NX = M::NX;
    using Sums = PassSums<P>;
    constexpr int NV = Sums::NV;
    constexpr int NW = T / 64;

    __shared__ double lds_red[NW * NV];
    __shared__ double lds_tot[NV];
    __shared__ LmState<P> lds_state;

    const int tid = threadIdx.x;
    const int G = ctx.G;
    const LmState<P> *prev = ctx.state[parity ^ 1];

    // ---------------- prologue: finish the previous launch's reduction, advance ------------
    const int prev_phase = prev->phase;
    const int prev_fresh = prev->bad_steps < 0; // host marks a brand-new state with bad_steps = -1
    if (prev_phase == PH_DONE)
    {
        // fit finished in an earlier launch: keep the final state visible under both parities
        if (blockIdx.x == 0 && tid == 0)
            *ctx.state[parity] = *prev;
        return;
    }

    if (!prev_fresh)
    {
        double vals[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v)
            vals[v] = 0.0;
        const double *pp = ctx.partials[parity ^ 1];
        for (int b = tid; b < G; b += T)
        {
#pragma unroll
            for (int v = 0; v < NV; ++v)
                vals[v] += pp[(size_t)v * G + b];
        }
        const double tot = block_sum_slots<NV, T>(vals, lds_red);
        if (tid < NV)
            lds_tot[tid] = tot;
        __syncthreads();
    }

    if (tid < 64)
    {
        // one wavefront runs the p-sized algebra (uniform across its lanes)
        LmState<P> s = *prev;
        if (prev_fresh)
            s.bad_steps = 0;
        else
        {
            Sums r;
            double *rf = reinterpret_cast<double *>(&r);
#pragma unroll
            for (int v = 0; v < NV; ++v)
                rf[v] = lds_tot[v];
            const int niter_before = s.niter;
            const int phase_before = s.phase;
            lm_advance<P>(s, r, ctx.prm);
            if (blockIdx.x == 0 && tid == 0)
            {
                // callback (src/nls.c:980-995): trace row 0 after init, row niter after each iteration
                if (ctx.ssrtrace)
                {
                    if (phase_before == PH_INIT)
                    {
                        ctx.ssrtrace[0] = s.chisq_init;
                        for (int k = 0; k < P; ++k)
                            ctx.partrace[(size_t)(ctx.prm.maxiter + 1) * k] = s.x[k];
                    }
                    else if (s.niter != niter_before && s.status != ST_EBADFUNC &&
                             !(s.status == ST_ENOPROG && niter_before == 0))
                    {
                        ctx.ssrtrace[s.niter] = s.chisq1;
                        for (int k = 0; k < P; ++k)
                            ctx.partrace[s.niter + (size_t)(ctx.prm.maxiter + 1) * k] = s.x[k];
                    }
                }
            }
        }
        if (tid == 0)
        {
            lds_state = s;
            if (blockIdx.x == 0)
                *ctx.state[parity] = s;
        }
    }
    __syncthreads();

    const int phase = lds_state.phase;
    if (phase == PH_DONE)
        return;

    // ---------------- pass over this workgroup's rows ---------------------------------------
    double th[P], vel[P], delta[P];
#pragma unroll
    for (int k = 0; k < P; ++k)
    {
        th[k] = (phase == PH_FVV) ? lds_state.x[k] : lds_state.xt[k];
        vel[k] = lds_state.vel[k];
    }
    fd_deltas<P>(th, ctx.prm.h_df, delta);

    Sums acc;
    pass_zero<P>(acc);
    const long long n = ctx.n;
    const double *__restrict__ yv = ctx.y;
    const double *__restrict__ swv = ctx.sw;
    const long long stride = (long long)G * T;
    for (long long i = (long long)blockIdx.x * T + tid; i < n; i += stride)
    {
        double xr[NX];
#pragma unroll
        for (int c = 0; c < NX; ++c)
            xr[c] = ctx.x[c][i];
        const double y = yv[i];
        const double sw = swv ? swv[i] : 1.0;
        double Jrow[P];
        if (phase == PH_FVV)
        {
            const double fv = row_fvv<M, JAC>(th, vel, delta, ctx.prm.h_fvv, ctx.prm.fvv_analytic != 0, xr, y, sw,
                                              Jrow, &acc.badj);
#pragma unroll
            for (int k = 0; k < P; ++k)
                acc.g[k] += Jrow[k] * fv;
        }
        else
        {
            const double f = row_fj<M, JAC>(th, delta, xr, y, sw, Jrow, &acc.badj);
            acc_fj<P>(acc, f, Jrow);
        }
    }

    // ---------------- workgroup reduction -> one partial set ---------------------------------
    const double tot = block_sum_slots<NV, T>(reinterpret_cast<const double *>(&acc), lds_red);
    if (tid < NV)
        ctx.partials[parity][(size_t)tid * G + blockIdx.x] = tot;
}

// After the fit: weighted residual and Jacobian at the final point, in the layout C_nls
// returns them (resid n; grad n x p column-major, src/nls.c:695-737), plus (J^T J)^-1.
template <class M, int JAC, int T>
__global__ __launch_bounds__(T) void lm_finalize_kernel(DenseCtx<M::P> ctx, int parity, double *resid, double *grad,
                                                        double *covar)
{
    constexpr int P = M::P;
    constexpr int NX = M::NX;
    const LmState<P> *s = ctx.state[parity];
    double th[P], delta[P];
#pragma unroll
    for (int k = 0; k < P; ++k)
        th[k] = s->x[k];
    fd_deltas<P>(th, ctx.prm.h_df, delta);
    const long long n = ctx.n;
    const long long stride = (long long)gridDim.x * T;
    for (long long i = (long long)blockIdx.x * T + threadIdx.x; i < n; i += stride)
    {
        double xr[NX];
#pragma unroll
        for (int c = 0; c < NX; ++c)
            xr[c] = ctx.x[c][i];
        const double sw = ctx.sw ? ctx.sw[i] : 1.0;
        double Jrow[P], nb = 0.0;
        const double f = row_fj<M, JAC>(th, delta, xr, ctx.y[i], sw, Jrow, &nb);
        if (resid)
            resid[i] = f;
        if (grad)
        {
#pragma unroll
            for (int k = 0; k < P; ++k)
                grad[i + (size_t)n * k] = Jrow[k];
        }
    }
    if (covar && blockIdx.x == 0 && threadIdx.x == 0)
        covar_from_jtj<P>(s->A, covar);
}

} // namespace gslnls
