// sparse_cg.hpp -- the Steihaug-Toint conjugate-gradient step of the large path (GSL multilarge cgst.c, restated in
// large_host.hpp::cgst_step) with its p-sized recurrences ON THE DEVICE.
//
// The host loop of large_host.hpp makes one round trip per CG iteration: u up, J^T J u and ||J u||^2 down
// (~100 us for the p = 500 README problem, two 1.6 MB PCIe hops per product at p = 2e5).  Here z, r, d live in HBM,
// one single-workgroup kernel per iteration does what the host did between two products (alpha, the trust-region
// boundary test and tau, the residual update, the convergence test, beta, the next direction), and the host only
// enqueues: iterations are enqueued in growing chunks, every kernel of an iteration returns at once when the
// `done` word is set, and the host reads that word once per chunk.
//
// Differences from the host loop are rounding only: norms are block reductions (max-scaled like the reference's
// dnrm2, fixed order) instead of a sequential scaled sum; ||J u||^2 is the same 256 block partials added in the
// same order.
#pragma once
#include <hip/hip_runtime.h>

namespace gslnls
{

struct SpCgScal
{
    int done;        // 0 running, 1 finished
    int status;      // ST_SUCCESS / ST_EMAXITER
    long long it;    // CG iterations started
    long long n_notrans, n_trans; // products with J and J^T the reference would have counted
    double norm_g, norm_r, delta;
    double njdx2;    // ||J dx||^2 of the finished step (speculative product for the predicted reduction)
    int njdx2_valid;
    int pad;
};

constexpr int SPCG_NPART = 256;

struct SpCgVecs
{
    const double *g, *diag; // inputs, p
    double *z, *r, *d;      // CG state, p
    double *u;              // input vector of the next product with J (d / diag, or dx at the end)
    const double *Bd;       // J^T J u of the last product
    const double *part;     // SPCG_NPART block partials of ||J u||^2
    double *dx;             // the step
    SpCgScal *s;
    int p;
    long long cgmaxit;
};

// sum / max over the workgroup in a fixed order; every thread gets the result.  lds: blockDim.x / 64 doubles.
__device__ __forceinline__ double spcg_block_sum(double v, double *lds)
{
    v = wave_sum(v);
    __syncthreads(); // lds may still be read from the previous reduction
    if ((threadIdx.x & 63) == 0)
        lds[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w)
        t += lds[w];
    return t;
}
__device__ __forceinline__ double spcg_block_max(double v, double *lds)
{
    for (int off = 32; off > 0; off >>= 1)
        v = fmax(v, __shfl_xor(v, off));
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
        lds[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w)
        t = fmax(t, lds[w]);
    return t;
}

// Euclidean norm of the p values f(i), scaled by their largest magnitude like dnrm2 (no overflow / underflow of the
// squares); +Inf when one of them is infinite
template <class F>
__device__ __forceinline__ double spcg_nrm2(int p, F f, double *lds)
{
    double amax = 0.0;
    for (int i = threadIdx.x; i < p; i += blockDim.x)
        amax = fmax(amax, fabs(f(i)));
    amax = spcg_block_max(amax, lds);
    if (amax == 0.0)
        return 0.0;
    if (isinf(amax))
        return INFINITY;
    double ssq = 0.0;
    for (int i = threadIdx.x; i < p; i += blockDim.x)
    {
        const double a = f(i) / amax;
        ssq = fma(a, a, ssq);
    }
    ssq = spcg_block_sum(ssq, lds);
    return amax * sqrt(ssq);
}

// cgst_step prologue (cgst.c:  z = 0, r = d = -D^-1 g, ||D^-1 g||) and the first product's input
__global__ __launch_bounds__(1024) void spcg_init_kernel(SpCgVecs v, double delta)
{
    __shared__ double lds[16];
    const int p = v.p;
    for (int i = threadIdx.x; i < p; i += blockDim.x)
    {
        const double t = -v.g[i] / v.diag[i];
        v.z[i] = 0.0;
        v.r[i] = t;
        v.d[i] = t;
        v.u[i] = t / v.diag[i];
    }
    const double norm_g = spcg_nrm2(p, [&](int i) { return v.g[i] / v.diag[i]; }, lds);
    if (threadIdx.x == 0)
    {
        SpCgScal *s = v.s;
        s->done = 0;
        s->status = 0;
        s->it = 0;
        s->n_notrans = 0;
        s->n_trans = 0;
        s->norm_g = norm_g;
        s->norm_r = norm_g; // r = -D^-1 g
        s->delta = delta;
        s->njdx2 = 0.0;
        s->njdx2_valid = 0;
    }
}

// what cgst.c does between two products (large_host.hpp::cgst_step, one trip of its loop after ops.jtjv)
__global__ __launch_bounds__(1024) void spcg_update_kernel(SpCgVecs v)
{
    __shared__ double lds[16];
    __shared__ double sh_nw2;
    SpCgScal *s = v.s;
    if (s->done)
        return;
    const int p = v.p;
    const double delta = s->delta, norm_r = s->norm_r, norm_g = s->norm_g;
    // ||J u||^2: the block partials in the order the host adds them
    if (threadIdx.x == 0)
    {
        double t = 0.0;
        for (int k = 0; k < SPCG_NPART; ++k)
            t += v.part[k];
        sh_nw2 = t;
    }
    __syncthreads();
    const double norm_Jd = sqrt(sh_nw2);
    long long n_notrans = s->n_notrans + 1, n_trans = s->n_trans;
    const long long it = s->it + 1;
    __syncthreads(); // every thread has read the scalars before thread 0 rewrites them

    // to the trust-region boundary along d from z:  tau = -t1 + sqrt(t1 u + (delta + |z|)(delta - |z|)) / |d|
    auto to_boundary = [&]() {
        const double norm_p = spcg_nrm2(p, [&](int i) { return v.z[i]; }, lds);
        const double norm_d = spcg_nrm2(p, [&](int i) { return v.d[i]; }, lds);
        double u = 0.0;
        for (int i = threadIdx.x; i < p; i += blockDim.x)
            u = fma(v.z[i], v.d[i], u);
        u = spcg_block_sum(u, lds);
        const double t1 = u / (norm_d * norm_d);
        const double t2 = t1 * u + (delta + norm_p) * (delta - norm_p);
        const double tau = -t1 + sqrt(t2) / norm_d;
        for (int i = threadIdx.x; i < p; i += blockDim.x)
            v.dx[i] = (v.z[i] + tau * v.d[i]) / v.diag[i];
    };
    auto finish = [&](int status) {
        // the step is final: its product with J is what the predicted reduction needs next
        for (int i = threadIdx.x; i < p; i += blockDim.x)
            v.u[i] = v.dx[i];
        if (threadIdx.x == 0)
        {
            s->done = 1;
            s->status = status;
            s->it = it;
            s->n_notrans = n_notrans;
            s->n_trans = n_trans;
        }
    };

    if (norm_Jd == 0.0)
    {
        to_boundary();
        finish(0);
        return;
    }
    double uu = norm_r / norm_Jd;
    const double alpha = uu * uu;
    uu = spcg_nrm2(p, [&](int i) { return v.z[i] + alpha * v.d[i]; }, lds);
    if (uu >= delta)
    {
        to_boundary();
        finish(0);
        return;
    }
    n_trans += 1;
    for (int i = threadIdx.x; i < p; i += blockDim.x)
    {
        v.z[i] = v.z[i] + alpha * v.d[i];
        v.r[i] -= alpha * (v.Bd[i] / v.diag[i]);
    }
    // (each thread reads back only what it wrote itself)
    const double norm_rp1 = spcg_nrm2(p, [&](int i) { return v.r[i]; }, lds);
    if (norm_rp1 / norm_g < 1.0e-6)
    {
        for (int i = threadIdx.x; i < p; i += blockDim.x)
            v.dx[i] = v.z[i] / v.diag[i];
        finish(0);
        return;
    }
    if (it >= v.cgmaxit)
    {
        for (int i = threadIdx.x; i < p; i += blockDim.x)
            v.dx[i] = v.z[i] / v.diag[i];
        finish(ST_EMAXITER);
        return;
    }
    uu = norm_rp1 / norm_r;
    const double beta = uu * uu;
    for (int i = threadIdx.x; i < p; i += blockDim.x)
    {
        const double dn = v.r[i] + beta * v.d[i];
        v.d[i] = dn;
        v.u[i] = dn / v.diag[i];
    }
    if (threadIdx.x == 0)
    {
        s->it = it;
        s->n_notrans = n_notrans;
        s->n_trans = n_trans;
        s->norm_r = norm_rp1;
    }
}

// ||J dx||^2 from the block partials of the product enqueued behind the finished step
__global__ void spcg_pred_kernel(SpCgVecs v)
{
    if (threadIdx.x == 0 && v.s->done && !v.s->njdx2_valid)
    {
        double t = 0.0;
        for (int k = 0; k < SPCG_NPART; ++k)
            t += v.part[k];
        v.s->njdx2 = t;
        v.s->njdx2_valid = 1;
    }
}

} // namespace gslnls
