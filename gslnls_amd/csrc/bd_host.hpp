// bd_host.hpp -- gsl_nls() with the Jacobian as a matrix in HBM: R `function` models of any size (the closures stay on the
// host, as fn / jac / fvv are evaluated with Rf_eval on the R thread in the reference, src/nls.c:815-978) and formulas
// with more than 64 parameters (rows evaluated by a kernel compiled in process for the formula).
//
// Replaces, for these inputs, the same functions as dense_host.hpp / wide_host.hpp: C_nls_internal's single-start branch
// (src/nls.c:533-576, :598-608, :632-753), gsl_multifit_nlinear_driver2 (src/nls_fit.c:40-121), trust_init_LD /
// trust_iterate_lu_LD / lm_step_LD / nielsen_* / trust_calc_rho (src/trust.c), eval_f / eval_df / eval_fvv with weights
// (src/fdf.c:94-233), the difference Jacobians (src/fdjac.c) and fvv (src/fdfvv.c).  The order of operations of one
// iteration is the one of lm_advance<P> (lm_core.hpp) and wide_advance (wide_core.hpp) -- the same state machine a third
// time, with a run-time p up to 4096 and p-vectors on the host: a trial step has to come back to the host anyway when
// the model is a closure, and for p > 64 the p x p factorisation (hundreds of microseconds) hides a round trip.
// Every n x p and p x p operation is a device kernel (bd_kernels.hpp; the damped solve: mchol_device.hip).
#pragma once
#include <atomic>
#include <mutex>
#include <map>
#include <hip/hip_runtime.h>
#include <math.h>
#include <algorithm>
#include <vector>
#include "../../include/gslnls_core.h"
#include "bd_kernels.hpp"
#include "dense_host.hpp"
#include "irls_kernels.hpp"
#include "large_host.hpp"
#include "mstart_driver.hpp"
#include "robust_host.hpp"

namespace gslnls
{

// ---- round 5: the matrix path's buffers are parked between fits ---------------------------------------------------------
// A fit of the matrix path needs about fifteen device buffers (the n x p Jacobian among them) and four pinned ones; allocating
// and freeing them per call was 6.5 ms of a p = 501 call whose LM loop takes 6.  bd_dev_alloc / bd_host_alloc hand back a
// parked block of exactly the requested size on the same device when there is one (a second fit of the same shape allocates
// nothing), bd_*_free park it again; beyond BD_POOL_DEV_BYTES / BD_POOL_HOST_BYTES parked, or BD_POOL_BLOCKS blocks, a
// block is really freed.  Contents are whatever the last user left (hipMalloc promises nothing else either).
// GSLNLS_BD_POOL=0 turns the parking off.  (The pool is never destroyed: no HIP calls during static teardown.)
constexpr size_t BD_POOL_DEV_BYTES = (size_t)4 << 30, BD_POOL_HOST_BYTES = (size_t)256 << 20;
constexpr int BD_POOL_BLOCKS = 96;
struct BdPool
{
    struct Blk
    {
        void *ptr;
        size_t bytes;
        int dev;
        unsigned flags;
        bool host;
    };
    std::mutex mu;
    std::vector<Blk> parked;
    std::map<void *, Blk> live;
    size_t dev_bytes = 0, host_bytes = 0;
    bool on = true;
    BdPool()
    {
        const char *e = getenv("GSLNLS_BD_POOL");
        on = !(e && atoi(e) == 0);
    }
    hipError_t take(void **out, size_t bytes, bool host, unsigned flags)
    {
        int dev = 0;
        (void)hipGetDevice(&dev);
        {
            std::lock_guard<std::mutex> lk(mu);
            for (size_t k = 0; k < parked.size(); ++k)
                if (parked[k].bytes == bytes && parked[k].host == host && parked[k].flags == flags && parked[k].dev == dev)
                {
                    Blk b = parked[k];
                    parked.erase(parked.begin() + (long)k);
                    (host ? host_bytes : dev_bytes) -= bytes;
                    live[b.ptr] = b;
                    *out = b.ptr;
                    return hipSuccess;
                }
        }
        void *q = nullptr;
        const hipError_t e = host ? hipHostMalloc(&q, bytes, flags) : hipMalloc(&q, bytes);
        if (e != hipSuccess)
        {
            // (the parked blocks may be what is in the way: let them go and try once more)
            drain();
            const hipError_t e2 = host ? hipHostMalloc(&q, bytes, flags) : hipMalloc(&q, bytes);
            if (e2 != hipSuccess)
                return e2;
            (void)hipGetLastError();
        }
        std::lock_guard<std::mutex> lk(mu);
        live[q] = Blk{q, bytes, dev, flags, host};
        *out = q;
        return hipSuccess;
    }
    void give(void *q)
    {
        if (!q)
            return;
        Blk b{q, 0, 0, 0, false};
        bool known = false, park = false;
        {
            std::lock_guard<std::mutex> lk(mu);
            auto it = live.find(q);
            if (it != live.end())
            {
                b = it->second;
                live.erase(it);
                known = true;
                park = on && (int)parked.size() < BD_POOL_BLOCKS &&
                       (b.host ? host_bytes + b.bytes <= BD_POOL_HOST_BYTES : dev_bytes + b.bytes <= BD_POOL_DEV_BYTES);
                if (park)
                {
                    parked.push_back(b);
                    (b.host ? host_bytes : dev_bytes) += b.bytes;
                }
            }
        }
        if (!park)
        {
            if (known && b.host)
                (void)hipHostFree(q);
            else
                (void)hipFree(q);
        }
    }
    void drain()
    {
        std::vector<Blk> all;
        {
            std::lock_guard<std::mutex> lk(mu);
            all.swap(parked);
            dev_bytes = host_bytes = 0;
        }
        for (const Blk &b : all)
        {
            if (b.host)
                (void)hipHostFree(b.ptr);
            else
                (void)hipFree(b.ptr);
        }
    }
};
inline BdPool &bd_pool()
{
    static BdPool *pl = new BdPool;
    return *pl;
}
template <class T>
inline hipError_t bd_dev_alloc(T **out, size_t bytes)
{
    return bd_pool().take(reinterpret_cast<void **>(out), bytes, false, 0);
}
template <class T>
inline hipError_t bd_host_alloc(T **out, size_t bytes, unsigned flags)
{
    return bd_pool().take(reinterpret_cast<void **>(out), bytes, true, flags);
}
inline void bd_pool_free(void *q) { bd_pool().give(q); }

// 2-norm condition number of the column-scaled normal matrix for any p: Cholesky of C = S A S, lambda_max by power
// iteration, lambda_min by inverse iteration through the factor (the boundary's solver-routing diagnostic,
// gslnls_solver_served; dense_host.hpp's Jacobi sweeps are cubic per sweep)
inline double bd_scaled_cond(int p, const std::vector<double> &A)
{
    std::vector<double> C((size_t)p * p), s(p);
    for (int i = 0; i < p; ++i)
    {
        const double d = A[(size_t)i * p + i];
        if (!(d > 0.0) || !std::isfinite(d))
            return INFINITY;
        s[i] = 1.0 / sqrt(d);
    }
    for (int i = 0; i < p; ++i)
        for (int j = 0; j < p; ++j)
            C[(size_t)i * p + j] = A[(size_t)i * p + j] * s[i] * s[j];
    std::vector<double> Lf(C);
    if (!lg_chol(p, Lf))
        return INFINITY;
    std::vector<double> v(p, 1.0 / sqrt((double)p)), w(p);
    double lmax = 0.0, lmin_inv = 0.0;
    for (int it = 0; it < 60; ++it)
    {
        double nrm = 0.0;
        for (int i = 0; i < p; ++i)
        {
            double t = 0.0;
            for (int j = 0; j < p; ++j)
                t += C[(size_t)i * p + j] * v[j];
            w[i] = t;
            nrm += t * t;
        }
        nrm = sqrt(nrm);
        if (!(nrm > 0.0))
            break;
        lmax = nrm;
        for (int i = 0; i < p; ++i)
            v[i] = w[i] / nrm;
    }
    for (int i = 0; i < p; ++i)
        v[i] = ((i & 1) ? -1.0 : 1.0) / sqrt((double)p);
    for (int it = 0; it < 60; ++it)
    {
        // w = C^-1 v through L L^T (lg_chol leaves L in the lower triangle)
        for (int i = 0; i < p; ++i)
        {
            double t = v[i];
            for (int k = 0; k < i; ++k)
                t -= Lf[(size_t)i * p + k] * w[k];
            w[i] = t / Lf[(size_t)i * p + i];
        }
        for (int i = p - 1; i >= 0; --i)
        {
            double t = w[i];
            for (int k = i + 1; k < p; ++k)
                t -= Lf[(size_t)k * p + i] * w[k];
            w[i] = t / Lf[(size_t)i * p + i];
        }
        double nrm = 0.0;
        for (int i = 0; i < p; ++i)
            nrm += w[i] * w[i];
        nrm = sqrt(nrm);
        if (!(nrm > 0.0) || !std::isfinite(nrm))
            return INFINITY;
        lmin_inv = nrm;
        for (int i = 0; i < p; ++i)
            v[i] = w[i] / nrm;
    }
    return lmax * lmin_inv;
}

// where the wall time of the last fit on the matrix path went (gslnls_last_matrix_path_profile): milliseconds
struct BdProfile
{
    double setup_ms = 0, loop_ms = 0, solve_ms = 0, jac_ms = 0, resid_ms = 0, covar_ms = 0, down_ms = 0, cond_ms = 0;
    long trial_steps = 0, jacobians = 0;
    int fused = 0, p = 0, n = 0;
};
inline BdProfile g_bd_prof;

// where the model comes from: host closures (R functions) or a kernel compiled for the formula
// r_i = f_i / sqrt(w_i) (the unweighted residual as the reference's IRLS driver recovers it, src/nls_irls.c:455-459) and the
// bit pattern of |r_i|: the key of the radix select that finds the median (irls_kernels.hpp)
static __global__ __launch_bounds__(256) void bd_unweight_keys_kernel(const double *fw, const double *sw, long long n, double *r,
                                                                       unsigned long long *keys)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    {
        const double v = fw[i] / sw[i];
        r[i] = v;
        keys[i] = (unsigned long long)__double_as_longlong(fabs(v));
    }
}

// ---- round 5: (J^T J)^-1 and the solver-routing diagnostic at a fit's end, on the device ----
// The damped solve's kernels factor A = J^T J (mu = 0, zero right-hand side: natural-order blocked Cholesky); behind them,
// in the same submission: X = L^-1 column by column (bd_trinv_kernel), A^-1 = X^T X on the matrix cores (bd_syrk_kernel,
// the kernel of J^T J itself) and, when cond_out is given, the two power iterations of bd_scaled_cond -- 60 products with
// C = S A S, 60 with C^-1 = S^-1 A^-1 S^-1 -- whose norms come home with the solve's one synchronisation.  Until round 5
// both were host loops, cubic in p: 25.6 + 18.5 ms of a 76 ms matrix-path call at p = 501 where the LM loop takes 6, and
// 15 ms of every gsl_nls_large call at p = 500.  d_A: p x p on the device, symmetric, left as it is.  Scratch from the
// parked pool.  Returns non-zero (and writes nothing) when the factorisation was refused -- a pivot not safely positive --
// or the device routine does not take the size: the caller keeps its host routine, whose rank rules then decide.
inline int bd_device_inverse(int p, const double *d_A, double *covar_host, double *cond_out)
{
    if (p < 1 || p > 4096 || !d_A)
        return GSLNLS_E_UNSUPPORTED;
    const size_t pp = (size_t)p * p;
    const BdSyrkGeom geom = bd_syrk_geom(p, p); // (X^T X of the p x p inverse factor)
    struct Scratch
    {
        double *epi = nullptr, *cpart = nullptr;
        ~Scratch()
        {
            bd_pool_free(epi);
            bd_pool_free(cpart);
        }
    } sc;
    if (bd_dev_alloc(&sc.epi, sizeof(double) * (2 * pp + 4 * (size_t)p + 8)) != hipSuccess ||
        bd_dev_alloc(&sc.cpart, sizeof(double) * geom.scratch) != hipSuccess)
    {
        (void)hipGetLastError();
        return GSLNLS_E_NODEVICE;
    }
    const int nblk = (p + 63) / 64;
    const size_t lds_ti = sizeof(double) * ((size_t)64 * nblk + 64 * BD_TI_LD + 64);
    static std::atomic<size_t> lds_set{0};
    if (lds_set.load() < lds_ti)
    {
        if (hipFuncSetAttribute((const void *)bd_trinv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_ti) != hipSuccess)
        {
            (void)hipGetLastError();
            return GSLNLS_E_NODEVICE;
        }
        lds_set.store(lds_ti);
    }
    struct Ctx
    {
        const double *d_A;
        double *epi, *cpart;
        size_t lds_ti;
        BdSyrkGeom geom;
        bool cond;
    } cx{d_A, sc.epi, sc.cpart, lds_ti, geom, cond_out != nullptr};
    auto enq = [](void *ctx, void *stream, const double *d_L, const double *d_dinv, int pq) {
        Ctx &c = *static_cast<Ctx *>(ctx);
        hipStream_t s2 = (hipStream_t)stream;
        const size_t ppq = (size_t)pq * pq;
        double *d_X = c.epi, *d_cov = d_X + ppq, *d_w = d_cov + ppq, *d_nrm = d_w + 4 * (size_t)pq;
        hipLaunchKernelGGL(bd_trinv_kernel, dim3(pq), dim3(BD_T), c.lds_ti, s2, d_L, d_dinv, pq, d_X);
        bd_syrk_launch(d_X, (long long)pq, pq, d_cov, c.cpart, c.geom, s2);
        if (!c.cond)
            return;
        hipLaunchKernelGGL(bd_power_start_kernel, dim3((pq + 255) / 256), dim3(256), 0, s2, pq, d_w, d_w + 2 * (size_t)pq);
        const int gw = (pq + BD_T / 64 - 1) / (BD_T / 64);
        for (int run = 0; run < 2; ++run)
        {
            double *wa = d_w + 2 * (size_t)run * pq, *wb = wa + pq;
            // 60 products; the 61st launch only delivers the norm of the last one
            for (int it = 0; it <= 60; ++it)
                hipLaunchKernelGGL(bd_power_kernel, dim3(gw), dim3(BD_T), sizeof(double) * pq, s2, run ? d_cov : c.d_A, c.d_A, pq, run,
                                   (it & 1) ? wb : wa, (it & 1) ? wa : wb, d_nrm + run);
        }
    };
    std::vector<double> zero(p, 0.0), sol(p, 0.0);
    double nrm[2] = {NAN, NAN};
    MCholTail tail;
    tail.enqueue_factor = enq;
    tail.ctx = &cx;
    tail.extra_dev = sc.epi + 2 * pp + 4 * (size_t)p;
    tail.extra_n = cond_out ? 2 : 0;
    tail.extra_host = nrm;
    int valid = 0;
    const int rc = mchol_device_solve_resident_tail(p, d_A, zero.data(), 0.0, zero.data(), sol.data(), &tail, &valid);
    if (rc)
        return rc;
    if (!valid)
        return GSLNLS_FAILURE;
    if (covar_host)
        GSLNLS_HIP_OK(hipMemcpy(covar_host, sc.epi + pp, sizeof(double) * pp, hipMemcpyDeviceToHost)); // (symmetric: either storage order)
    if (cond_out)
    {
        const double c = nrm[0] * nrm[1];
        *cond_out = std::isfinite(c) && c > 0.0 ? c : INFINITY;
    }
    return GSLNLS_SUCCESS;
}

struct BdModel
{
    virtual ~BdModel() {}
    // model values m(theta) -> d_fval[n] on the device (NOT residuals).  Non-zero: the closure failed (EBADFUNC).
    virtual int values(const double *theta, double *d_fval, hipStream_t st) = 0;
    // analytic Jacobian dm/dtheta -> d_J, n x p column-major, unweighted.  Only called when has_jac.
    virtual int jacobian(const double *theta, double *d_J, hipStream_t st) = 0;
    // analytic second directional derivative D^2 m[v, v] -> d_out[n], unweighted.  Only called when has_fvv.
    virtual int fvv(const double *theta, const double *v, double *d_out, hipStream_t st) = 0;
    bool has_jac = false, has_fvv = false;
    // models whose rows are evaluated by a device kernel (a formula) can take theta from device memory: the trial step then
    // evaluates m(x + dx) behind the damped solve without coming back to the host in between (BdFit::solve, round 5)
    virtual bool theta_on_device() const { return false; }
    virtual int values_dev(const double *d_theta, double *d_fval, hipStream_t st)
    {
        (void)d_theta;
        (void)d_fval;
        (void)st;
        return 1;
    }
    // round 5: f = sqrt(w) (m(theta) - y) and the partial sums of its squares (bd_resid_kernel's, bit for bit) by the kernel that
    // evaluates m, theta on the device.  -1: not offered (values_dev + bd_resid_kernel serve)
    virtual int resid_dev(const double *d_theta, const double *d_y, const double *d_sw, double *d_f, double *d_parts, int g, hipStream_t st)
    {
        (void)d_theta;
        (void)d_y;
        (void)d_sw;
        (void)d_f;
        (void)d_parts;
        (void)g;
        (void)st;
        return -1;
    }
    // round 5: the analytic Jacobian with the rows' sqrt(w) applied and one non-finite flag per workgroup written to d_part
    // (*nparts of them) by the kernel that produces the rows; theta from d_theta when non-null.  -1: not offered (the
    // caller runs jacobian() and bd_weight_kernel), > 0: failure.
    virtual int jacobian_flagged(const double *theta, const double *d_theta, double *d_J, const double *d_sw, double *d_part, int *nparts,
                                 hipStream_t st)
    {
        (void)theta;
        (void)d_theta;
        (void)d_J;
        (void)d_sw;
        (void)d_part;
        (void)nparts;
        (void)st;
        return -1;
    }
};

// xt = x + dx, a component that would leave its bound moved to x + dx / max(|dx|, delta) |x - bound| (trust_trial_step_lu,
// src/trust.c:9-32): the host loop of BdFit::solve, one thread per component
static __global__ __launch_bounds__(256) void bd_trial_kernel(const double *x, const double *dx, const double *lo, const double *up, double delta,
                                                               int has_bounds, int p, double *xt)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= p)
        return;
    const double xk = x[k], dk = dx[k];
    double t = xk + dk;
    if (has_bounds)
    {
        // (product and sum rounded separately, as the host's loop rounds them: no contraction into an fma)
        if (t < lo[k])
            t = __dadd_rn(xk, __dmul_rn(dk / fmax(fabs(dk), delta), fabs(xk - lo[k])));
        else if (t > up[k])
            t = __dadd_rn(xk, __dmul_rn(dk / fmax(fabs(dk), delta), fabs(xk - up[k])));
    }
    xt[k] = t;
}

// bd_trial_kernel and bd_quad_kernel in one launch (round 5): workgroup i takes row i of dx^T (J^T J) dx, its first thread
// component i of the trial point -- each the arithmetic of its own kernel
static __global__ __launch_bounds__(BD_T) void bd_trial_quad_kernel(const double *x, const double *dx, const double *lo, const double *up, double delta,
                                                                     int has_bounds, int p, double *xt, const double *C, double *quad)
{
    __shared__ double red_s[BD_T / 64];
    const int i = blockIdx.x;
    if (threadIdx.x == 0)
    {
        const double xk = x[i], dk = dx[i];
        double t = xk + dk;
        if (has_bounds)
        {
            if (t < lo[i])
                t = __dadd_rn(xk, __dmul_rn(dk / fmax(fabs(dk), delta), fabs(xk - lo[i])));
            else if (t > up[i])
                t = __dadd_rn(xk, __dmul_rn(dk / fmax(fabs(dk), delta), fabs(xk - up[i])));
        }
        xt[i] = t;
    }
    const double *row = C + (size_t)i * p;
    double s = 0.0;
    for (int j = threadIdx.x; j < p; j += BD_T)
        s += row[j] * dx[j];
    s = bd_block_sum(s, red_s);
    if (threadIdx.x == 0)
        quad[i] = s * dx[i];
}

// what a Jacobian evaluation sends home -- g = J^T f, diag(J^T J), the non-finite flags of an analytic Jacobian -- written
// into pinned host memory through its mapping, the sequence number of the evaluation behind a system-scope fence: the host
// polls that word (round 5: a copy engine's pitched copy of the diagonal plus two more copies and the wake-up of a
// thread blocked in hipStreamSynchronize were ~40 us of every accepted point)
static __global__ __launch_bounds__(256) void bd_publish_kernel(const double *g, const double *C, int p, const double *parts, int nparts,
                                                                 double *h_out, unsigned long long seq)
{
    for (int k = threadIdx.x; k < p; k += 256)
    {
        h_out[k] = g[k];
        h_out[p + k] = C[(size_t)k * p + k];
    }
    for (int k = threadIdx.x; k < nparts; k += 256)
        h_out[2 * (size_t)p + k] = parts[k];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0)
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(h_out + 2 * (size_t)p + BD_MAXG), seq, __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_SYSTEM);
}

struct BdFit
{
    int n = 0, p = 0;
    BdModel *model = nullptr;
    hipStream_t st = nullptr;
    double *d_y = nullptr, *d_sw = nullptr, *d_fval = nullptr, *d_f[2] = {nullptr, nullptr}, *d_fp = nullptr, *d_fm = nullptr,
           *d_J = nullptr, *d_C = nullptr, *d_cpart = nullptr, *d_part = nullptr, *d_pv = nullptr, *d_u = nullptr;
    std::vector<double> h_part;
    // round 5: x, bounds and what a fused trial step sends home (x + dx | rows of dx^T J^T J dx | partial sums of ||f||^2) on
    // the device; pinned staging for every small copy (a pageable destination makes an asynchronous copy a blocking one)
    double *d_pub = nullptr, *h_pin = nullptr;
    double *h_xmap = nullptr, *d_xmap = nullptr; // pinned + mapped, 3 p doubles: x | lower | upper bounds, read by bd_trial_kernel in place
    double *h_jmap = nullptr, *d_jmap = nullptr; // pinned + mapped: g | diag(J^T J) | flags (BD_MAXG) | sequence word (bd_publish_kernel)
    unsigned long long jseq = 0;
    hipEvent_t ev_j = nullptr;
    const double *jac_theta_dev = nullptr; // set around a jac_at() whose point is already on the device (the accepted x + dx of a fused trial step)
    int cur = 0;        // d_f[cur]: residual at the current point, d_f[cur ^ 1]: at the trial point
    std::vector<double> last_x;    // where the last solve ended (whatever its status)
    // more of the state the last solve ended in -- what gsl_multistart_driver reads out of the solver workspace after a
    // concentration / local-search fit (src/nls_mstart.c:91-95, :254-258, :324-326): trust_state->diag, chisq0 / chisq1 of
    // driver2, and det(J^T J) of the workspace Jacobian (filled only when the solve was asked for it: `want_det`)
    std::vector<double> end_diag;
    double end_chisq0 = INFINITY, end_chisq1 = INFINITY, end_det = 0.0;
    int end_niter = 0, end_status = ST_CONTINUE;
    const double *last_f = nullptr; // its weighted residual there, on the device
    void *irls_arena = nullptr;
    size_t irls_arena_bytes = 0;
    BdSyrkGeom geom; // how J^T J of this fit's n x p Jacobian is formed (bd_syrk_geom)
    long nevalf = 0, nevaldf = 0, nevalfvv = 0;
    int device_ordinal = -1;

    ~BdFit() { release(); }
    void release()
    {
        if (irls_arena)
            (void)hipFree(irls_arena);
        irls_arena = nullptr;
        irls_arena_bytes = 0;
        double *bufs[] = {d_y, d_sw, d_fval, d_f[0], d_f[1], d_fp, d_fm, d_J, d_C, d_cpart, d_part, d_pv, d_u, d_pub};
        for (double *b : bufs)
            if (b)
                bd_pool_free(b);
        d_y = d_sw = d_fval = d_f[0] = d_f[1] = d_fp = d_fm = d_J = d_C = d_cpart = d_part = d_pv = d_u = nullptr;
        d_pub = nullptr;
        if (h_pin)
            bd_pool_free(h_pin);
        if (h_xmap)
            bd_pool_free(h_xmap);
        if (h_jmap)
            bd_pool_free(h_jmap);
        h_pin = h_xmap = d_xmap = h_jmap = d_jmap = nullptr;
        if (ev_j)
            (void)hipEventDestroy(ev_j);
        ev_j = nullptr;
        if (st)
            (void)hipStreamDestroy(st);
        st = nullptr;
    }

    int init(int n_, int p_, const double *y, const double *swts, BdModel *m)
    {
        n = n_;
        p = p_;
        model = m;
        if (n < 1 || p < 1 || p > 4096 || !y)
            return GSLNLS_EINVAL;
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        {
            fprintf(stderr, "gslnls: no HIP device available -- the MI355X path cannot run (no CPU fallback exists)\n");
            return GSLNLS_E_NODEVICE;
        }
        (void)hipGetDevice(&device_ordinal);
        GSLNLS_HIP_OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        const size_t nb = sizeof(double) * (size_t)n;
        geom = bd_syrk_geom(n, p);
        GSLNLS_HIP_OK(bd_dev_alloc(&d_y, nb));
        GSLNLS_HIP_OK(bd_dev_alloc(&d_fval, nb));
        GSLNLS_HIP_OK(bd_dev_alloc(&d_f[0], nb));
        GSLNLS_HIP_OK(bd_dev_alloc(&d_f[1], nb));
        GSLNLS_HIP_OK(bd_dev_alloc(&d_fp, nb));
        GSLNLS_HIP_OK(bd_dev_alloc(&d_fm, nb));
        GSLNLS_HIP_OK(bd_dev_alloc(&d_u, nb));
        GSLNLS_HIP_OK(bd_dev_alloc(&d_J, nb * p));
        GSLNLS_HIP_OK(bd_dev_alloc(&d_C, sizeof(double) * (size_t)p * p));
        GSLNLS_HIP_OK(bd_dev_alloc(&d_cpart, sizeof(double) * geom.scratch));
        GSLNLS_HIP_OK(bd_dev_alloc(&d_part, sizeof(double) * BD_MAXG));
        GSLNLS_HIP_OK(bd_dev_alloc(&d_pv, sizeof(double) * (size_t)4 * p));
        GSLNLS_HIP_OK(bd_dev_alloc(&d_pub, sizeof(double) * ((size_t)2 * p + BD_MAXG)));
        GSLNLS_HIP_OK(bd_host_alloc(&h_xmap, sizeof(double) * (size_t)3 * p, hipHostMallocMapped));
        GSLNLS_HIP_OK(hipHostGetDevicePointer((void **)&d_xmap, h_xmap, 0));
        GSLNLS_HIP_OK(bd_host_alloc(&h_jmap, sizeof(double) * ((size_t)2 * p + BD_MAXG + 2), hipHostMallocMapped));
        GSLNLS_HIP_OK(hipHostGetDevicePointer((void **)&d_jmap, h_jmap, 0));
        GSLNLS_HIP_OK(hipEventCreateWithFlags(&ev_j, hipEventDisableTiming));
        memset(h_jmap + 2 * (size_t)p + BD_MAXG, 0, 2 * sizeof(double)); // (the sequence word: jseq counts from 1)
        GSLNLS_HIP_OK(bd_host_alloc(&h_pin, sizeof(double) * ((size_t)4 * p + 2 * BD_MAXG), hipHostMallocDefault));
        h_part.resize(BD_MAXG);
        GSLNLS_HIP_OK(hipMemcpyAsync(d_y, y, nb, hipMemcpyHostToDevice, st));
        if (swts)
        {
            GSLNLS_HIP_OK(bd_dev_alloc(&d_sw, nb));
            GSLNLS_HIP_OK(hipMemcpyAsync(d_sw, swts, nb, hipMemcpyHostToDevice, st));
        }
        GSLNLS_HIP_OK(hipStreamSynchronize(st));
        return GSLNLS_SUCCESS;
    }

    int grid_n() const
    {
        long long g = ((long long)n + BD_T - 1) / BD_T;
        return (int)(g < 1 ? 1 : (g > BD_MAXG ? BD_MAXG : g));
    }
    int sum_parts(int g, double *out)
    {
        GSLNLS_HIP_OK(hipMemcpyAsync(h_part.data(), d_part, sizeof(double) * g, hipMemcpyDeviceToHost, st));
        GSLNLS_HIP_OK(hipStreamSynchronize(st));
        double s = 0.0;
        for (int k = 0; k < g; ++k)
            s += h_part[k];
        *out = s;
        return GSLNLS_SUCCESS;
    }

    // weighted residual at theta into d_dst (and its sum of squares when ssr != nullptr)
    int resid_at(const double *theta, double *d_dst, double *ssr)
    {
        if (model->values(theta, d_fval, st))
            return GSLNLS_EBADFUNC;
        const int g = grid_n();
        hipLaunchKernelGGL(bd_resid_kernel, dim3(g), dim3(BD_T), 0, st, d_fval, d_y, d_sw, (long long)n, d_dst, ssr ? d_part : nullptr);
        if (ssr)
            return sum_parts(g, ssr);
        return GSLNLS_SUCCESS;
    }

    // Jacobian at theta (residual at theta in d_fbase) -> d_J weighted, then J^T J -> d_C, g = J^T f_base, diag(J^T J).
    // *badj != 0: an analytic Jacobian with a non-finite entry.  jtj_host (p x p row-major) is filled when not null.
    int jac_at(const double *theta, const double *d_fbase, const LmParams &prm, double *g_out, double *djj_out, double *badj,
               double *jtj_host)
    {
        *badj = 0.0;
        int nbadparts = 0;
        int fl = -1;
        if (prm.jac_analytic && !getenv("GSLNLS_BD_STEPWISE"))
        {
            fl = model->jacobian_flagged(theta, jac_theta_dev, d_J, d_sw, d_part, &nbadparts, st);
            if (fl > 0)
                return GSLNLS_EBADFUNC;
        }
        if (fl == 0)
            ; // (rows, weights and flags by one kernel)
        else if (prm.jac_analytic)
        {
            if (model->jacobian(theta, d_J, st))
                return GSLNLS_EBADFUNC;
            const long long tot = (long long)n * p;
            long long gl = (tot + BD_T - 1) / BD_T;
            const int g = (int)(gl > BD_MAXG ? BD_MAXG : gl);
            hipLaunchKernelGGL(bd_weight_kernel, dim3(g), dim3(BD_T), 0, st, d_J, d_sw, (long long)n, p, d_part);
            // (its flags travel with g and the diagonal below: one synchronisation per Jacobian, not two)
            nbadparts = g;
        }
        else
        {
            // src/fdjac.c: delta_j = h |x_j| (0 -> h); forward (f(x + delta e_j) - f(x)) / delta, central over +- delta / 2
            std::vector<double> th(theta, theta + p);
            const int g = grid_n();
            for (int j = 0; j < p; ++j)
            {
                double d = prm.h_df * fabs(theta[j]);
                if (d == 0.0)
                    d = prm.h_df;
                if (prm.fdtype == 0)
                {
                    th[j] = theta[j] + d;
                    if (int rc = resid_at(th.data(), d_fp, nullptr))
                        return rc;
                    hipLaunchKernelGGL(bd_fdcol_kernel, dim3(g), dim3(BD_T), 0, st, d_fp, d_fbase, 1.0 / d, d_J + (size_t)j * n,
                                       (long long)n);
                }
                else
                {
                    th[j] = theta[j] + 0.5 * d;
                    if (int rc = resid_at(th.data(), d_fp, nullptr))
                        return rc;
                    th[j] = theta[j] - 0.5 * d;
                    if (int rc = resid_at(th.data(), d_fm, nullptr))
                        return rc;
                    hipLaunchKernelGGL(bd_fdcol_kernel, dim3(g), dim3(BD_T), 0, st, d_fp, d_fm, 1.0 / d, d_J + (size_t)j * n,
                                       (long long)n);
                }
                th[j] = theta[j];
            }
        }
        if (getenv("GSLNLS_BD_STEPWISE"))
        {
            bd_syrk_launch(d_J, (long long)n, p, d_C, d_cpart, geom, st);
            hipLaunchKernelGGL(bd_gemv_t_kernel, dim3(p), dim3(BD_T), 0, st, d_J, d_fbase, (long long)n, p, d_pv);
        }
        else
            bd_syrk_gemv_launch(d_J, (long long)n, p, d_C, d_cpart, geom, d_fbase, d_pv, st);
        if (jtj_host) // (the host factorises: the whole matrix comes down, and the stream's end covers everything)
        {
            GSLNLS_HIP_OK(hipMemcpyAsync(h_pin, d_pv, sizeof(double) * p, hipMemcpyDeviceToHost, st));
            if (nbadparts)
                GSLNLS_HIP_OK(hipMemcpyAsync(h_pin + 4 * (size_t)p, d_part, sizeof(double) * nbadparts, hipMemcpyDeviceToHost, st));
            GSLNLS_HIP_OK(hipMemcpyAsync(jtj_host, d_C, sizeof(double) * (size_t)p * p, hipMemcpyDeviceToHost, st));
            GSLNLS_HIP_OK(hipStreamSynchronize(st));
            memcpy(g_out, h_pin, sizeof(double) * p);
            for (int j = 0; j < p; ++j)
                djj_out[j] = jtj_host[(size_t)j * p + j];
            if (nbadparts)
            {
                double b = 0.0;
                for (int k = 0; k < nbadparts; ++k)
                    b += h_pin[4 * (size_t)p + k];
                *badj = (b == 0.0) ? 0.0 : 1.0;
            }
            GSLNLS_HIP_OK(hipGetLastError());
            return GSLNLS_SUCCESS;
        }
        // g, diag(J^T J) and the flags through mapped host memory; the host polls the sequence word behind them
        jseq += 1;
        volatile unsigned long long *word = reinterpret_cast<volatile unsigned long long *>(h_jmap + 2 * (size_t)p + BD_MAXG);
        hipLaunchKernelGGL(bd_publish_kernel, dim3(1), dim3(256), 0, st, d_pv, d_C, p, d_part, nbadparts, d_jmap, jseq);
        // (the word is polled; the stream is asked only now and then -- to notice a launch failure or a device fault, which
        // would never write the word.  Until round 5 an event was recorded behind the kernel and queried in every turn of
        // the loop: a marker packet on the device and a runtime call per poll on the host, for nothing in the good case)
        for (unsigned spin = 1;; ++spin)
        {
            if (*word == jseq)
                break;
            if (spin & 255u)
            {
                __builtin_ia32_pause();
                continue;
            }
            const hipError_t q = hipStreamQuery(st);
            if (q == hipSuccess)
            {
                if (*word != jseq)
                    GSLNLS_HIP_OK(hipStreamSynchronize(st));
                break;
            }
            if (q != hipErrorNotReady)
            {
                (void)hipGetLastError();
                return GSLNLS_E_NODEVICE;
            }
        }
        __sync_synchronize();
        memcpy(g_out, h_jmap, sizeof(double) * p);
        memcpy(djj_out, h_jmap + p, sizeof(double) * p);
        if (nbadparts)
        {
            double b = 0.0;
            for (int k = 0; k < nbadparts; ++k)
                b += h_jmap[2 * (size_t)p + k];
            *badj = (b == 0.0) ? 0.0 : 1.0;
        }
        return GSLNLS_SUCCESS;
    }

    // J^T fvv at x along v (src/fdf.c:200-233; by differences src/fdfvv.c:35-77); *bad != 0: non-finite analytic fvv
    int fvv_at(const double *x, const double *v, const LmParams &prm, double *out, double *bad)
    {
        *bad = 0.0;
        const int g = grid_n();
        if (prm.fvv_analytic)
        {
            if (model->fvv(x, v, d_fp, st))
                return GSLNLS_EBADFUNC;
            hipLaunchKernelGGL(bd_weight_vec_kernel, dim3(g), dim3(BD_T), 0, st, d_fp, d_sw, (long long)n, d_part);
            double b = 0.0;
            if (int rc = sum_parts(g, &b))
                return rc;
            *bad = (b == 0.0) ? 0.0 : 1.0;
        }
        else
        {
            std::vector<double> xh(p);
            for (int k = 0; k < p; ++k)
                xh[k] = x[k] + prm.h_fvv * v[k];
            if (int rc = resid_at(xh.data(), d_fm, nullptr))
                return rc;
            GSLNLS_HIP_OK(hipMemcpyAsync(d_pv + p, v, sizeof(double) * p, hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(bd_gemv_n_kernel, dim3(g), dim3(BD_T), 0, st, d_J, d_pv + p, (long long)n, p, d_u);
            hipLaunchKernelGGL(bd_fdfvv_kernel, dim3(g), dim3(BD_T), 0, st, d_fm, d_f[cur], d_u, prm.h_fvv, d_fp, (long long)n);
        }
        hipLaunchKernelGGL(bd_gemv_t_kernel, dim3(p), dim3(BD_T), 0, st, d_J, d_fp, (long long)n, p, d_pv);
        GSLNLS_HIP_OK(hipMemcpyAsync(out, d_pv, sizeof(double) * p, hipMemcpyDeviceToHost, st));
        GSLNLS_HIP_OK(hipStreamSynchronize(st));
        return GSLNLS_SUCCESS;
    }

    // v^T (J^T J) v with J^T J on the device (p >= the device threshold)
    int quad_device(const double *v, double *out)
    {
        GSLNLS_HIP_OK(hipMemcpyAsync(d_pv + p, v, sizeof(double) * p, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(bd_quad_kernel, dim3(p), dim3(BD_T), 0, st, d_C, d_pv + p, p, d_pv + 2 * p);
        std::vector<double> r(p);
        GSLNLS_HIP_OK(hipMemcpyAsync(r.data(), d_pv + 2 * p, sizeof(double) * p, hipMemcpyDeviceToHost, st));
        GSLNLS_HIP_OK(hipStreamSynchronize(st));
        double s = 0.0;
        for (int i = 0; i < p; ++i)
            s += r[i];
        *out = s;
        return GSLNLS_SUCCESS;
    }

    // det_cholesky_jtj (src/nls_utils.c:55-73): (prod L_ii)^2 of the plain Cholesky factor, 0 when it does not exist
    // (J^T J)^-1 and the solver-routing diagnostic at a fit's end, on the device: bd_device_inverse below
    int device_epilogue(double *covar_out, double *cond_out) { return bd_device_inverse(p, d_C, covar_out, cond_out); }

    static double det_chol(std::vector<double> &M)
    {
        const int q = (int)llround(sqrt((double)M.size()));
        if (!lg_chol(q, M))
            return 0.0;
        double det = 1.0;
        for (int i = 0; i < q; ++i)
            det *= M[(size_t)i * q + i];
        return det * det;
    }
    // det_eval_jtj (src/nls_utils.c:23-53): f and J at x, det(J^T J); 0 when either evaluation fails.  *ssr = ||f||^2.
    int det_at(const double *x, int jac, const int *ci, const double *cd, double *det, double *ssr)
    {
        const LmParams prm = make_params(ci, cd, jac, 0, false, d_sw != nullptr);
        *det = 0.0;
        *ssr = INFINITY;
        int rc = resid_at(x, d_f[cur], ssr);
        if (rc == GSLNLS_EBADFUNC)
            return GSLNLS_SUCCESS;
        if (rc)
            return rc;
        std::vector<double> gtmp(p), dtmp(p), M((size_t)p * p);
        double badj = 0.0;
        rc = jac_at(x, d_f[cur], prm, gtmp.data(), dtmp.data(), &badj, M.data());
        if (rc == GSLNLS_EBADFUNC || badj != 0.0)
            return GSLNLS_SUCCESS;
        if (rc)
            return rc;
        *det = det_chol(M);
        return GSLNLS_SUCCESS;
    }

    // The fit.  Same decisions in the same order as lm_advance<P> / wide_advance (which cite the reference line by line).
    // chisq_in: the chi^2 the convergence test of the first iteration compares with, when this solve is a re-solve of the
    // IRLS driver (NaN: the start's own, as a plain fit)
    // point_fit_maxiter >= 0: one of the multi-start driver's short fits (src/nls_mstart.c:91, :254): that many iterations at
    // most, gtol = 1e-3, no result vectors -- only the end state above, with det(J^T J) where the fit ended
    int solve(int jac, int fvv, const double *start, const double *lupars, const int *ci, const double *cd, gslnls_result *out,
              double chisq_in = NAN, int point_fit_maxiter = -1)
    {
        if (ci[2] > 1)
            return GSLNLS_E_UNSUPPORTED; // dogleg family: not lowered
        if ((jac && !model->has_jac) || (fvv && !model->has_fvv))
            return GSLNLS_EINVAL;
        LmParams prm_ = make_params(ci, cd, jac, fvv, lupars != nullptr, d_sw != nullptr);
        const bool point_fit = point_fit_maxiter >= 0;
        if (point_fit)
        {
            prm_.maxiter = point_fit_maxiter;
            prm_.gtol = 1e-3;
        }
        const LmParams prm = prm_;
        const int maxiter = prm.maxiter;
        const bool trace = !point_fit && ci[1] != 0 && out->ssrtrace && out->partrace;
        // (read per fit: the tests run one problem through both factorisations)
        // (round 5: the device from p = 65 on -- "the damped solve resident on the device", BASELINE north_star; measured with
        // the round-5 kernels, profiles/r05_large_chol_solve_times.txt: device 0.063 / 0.066 / 0.123 / 0.196 ms at p = 100 / 128 /
        // 200 / 333 against the host's 0.039 / 0.076 / 0.219 / 0.780)
        const char *dev_env = getenv("GSLNLS_LARGE_CHOL_DEVICE_MIN");
        const int dev_min = dev_env ? atoi(dev_env) : 65;
        const bool dev_solve = dev_min > 0 && p >= dev_min;
        std::vector<double> x(start, start + p), xt(p), dx(p, 0.0), vel(p, 0.0), acc(p, 0.0), g(p), gt(p), diag(p, 1.0), djj(p),
            djjt(p), lo(p, -INFINITY), up(p, INFINITY), rhs(p), gfvv(p);
        std::vector<double> A(dev_solve ? 0 : (size_t)p * p), Adamp;
        if (lupars)
            for (int k = 0; k < p; ++k)
            {
                lo[k] = std::isfinite(lupars[2 * k]) ? lupars[2 * k] : -INFINITY;
                up[k] = std::isfinite(lupars[2 * k + 1]) ? lupars[2 * k + 1] : INFINITY;
            }
        nevalf = nevaldf = nevalfvv = 0;
        double fnorm2 = INFINITY, mu = 0.0, nu = 2.0, delta = 0.0, avratio = 0.0, chisq0 = INFINITY, chisq1 = INFINITY,
               chisq_init = INFINITY;
        int niter = 0, status = ST_CONTINUE, info = ST_CONTINUE, bad_steps = 0;
        const double t_begin = now_s();
        if (trace)
        {
            for (int k = 0; k <= maxiter; ++k)
                out->ssrtrace[k] = NAN;
            for (size_t k = 0; k < (size_t)(maxiter + 1) * p; ++k)
                out->partrace[k] = NAN;
        }
        auto scale = [&](bool init) { // GSL scaling.c
            for (int j = 0; j < p; ++j)
            {
                if (prm.scale == 1)
                {
                    if (init)
                        diag[j] = 1.0;
                    continue;
                }
                double norm = sqrt(djj[j]);
                if (norm == 0.0)
                    norm = 1.0;
                if (init || prm.scale == 2)
                    diag[j] = norm;
                else
                    diag[j] = fmax(diag[j], norm);
            }
        };
        // (J^T J + mu D^2) sol = b: gsl_linalg_mcholesky on the device (every p of this path unless the threshold is moved), else the host routine
        auto damped_solve = [&](const std::vector<double> &b, std::vector<double> &sol) -> int {
            sol.assign(p, 0.0);
            if (dev_solve)
            {
                const int rc = mchol_device_solve_resident(p, d_C, diag.data(), mu, b.data(), sol.data());
                if (rc != GSLNLS_E_UNSUPPORTED)
                    return rc;
                std::vector<double> Ah((size_t)p * p);
                GSLNLS_HIP_OK(hipMemcpy(Ah.data(), d_C, sizeof(double) * (size_t)p * p, hipMemcpyDeviceToHost));
                for (int i = 0; i < p; ++i)
                    Ah[(size_t)i * p + i] += mu * diag[i] * diag[i];
                lg_mchol_solve(p, Ah, b, sol);
                return GSLNLS_SUCCESS;
            }
            Adamp = A;
            for (int i = 0; i < p; ++i)
                Adamp[(size_t)i * p + i] += mu * diag[i] * diag[i];
            lg_mchol_solve(p, Adamp, b, sol);
            return GSLNLS_SUCCESS;
        };
        auto test = [&](int *inf) -> int { // gsl_multifit_nlinear_test
            bool small = true;
            for (int i = 0; i < p && small; ++i)
                if (!(fabs(dx[i]) < prm.xtol * prm.xtol + prm.xtol * fabs(x[i])))
                    small = false;
            if (small)
            {
                *inf = 1;
                return ST_SUCCESS;
            }
            double gnorm = 0.0;
            for (int i = 0; i < p; ++i)
                gnorm = fmax(gnorm, fabs(fmax(x[i], 1.0) * g[i]));
            if (gnorm <= prm.gtol * fmax(0.5 * fnorm2, 1.0))
            {
                *inf = 2;
                return ST_SUCCESS;
            }
            *inf = 0;
            return ST_CONTINUE;
        };
        bool done = false;
        auto trace_row = [&](int row, double ssr) {
            if (!trace)
                return;
            out->ssrtrace[row] = ssr;
            for (int k = 0; k < p; ++k)
                out->partrace[row + (size_t)(maxiter + 1) * k] = x[k];
        };
        auto end_iteration = [&](int itstatus) -> bool { // lm_end_iteration: true = another iteration follows
            const int iter = niter;
            niter += 1;
            chisq1 = fnorm2;
            if (itstatus == ST_EBADFUNC || (itstatus == ST_ENOPROG && iter == 0))
            {
                info = itstatus;
                status = itstatus;
                done = true;
                return false;
            }
            trace_row(niter, chisq1);
            int inf = 0;
            const int t = test(&inf);
            info = inf;
            if (t == ST_SUCCESS)
            {
                status = ST_SUCCESS;
                done = true;
                return false;
            }
            if (niter >= maxiter)
            {
                status = ST_EMAXITER;
                done = true;
                return false;
            }
            chisq0 = chisq1;
            bad_steps = 0;
            return true;
        };

        // ---- trust_init_LD ----
        int rc = resid_at(x.data(), d_f[cur], &fnorm2);
        if (rc == GSLNLS_EBADFUNC)
        {
            status = info = ST_EBADFUNC;
            done = true;
        }
        else if (rc)
            return rc;
        double badj = 0.0;
        if (!done)
        {
            nevalf += 1;
            rc = jac_at(x.data(), d_f[cur], prm, g.data(), djj.data(), &badj, dev_solve ? nullptr : A.data());
            if (rc == GSLNLS_EBADFUNC)
                badj = 1.0;
            else if (rc)
                return rc;
            if (prm.jac_analytic)
                nevaldf += 1;
            else
                nevalf += lm_fd_cost(prm, p);
            if (badj != 0.0)
            {
                chisq_init = chisq0 = chisq1 = fnorm2;
                status = info = ST_EBADFUNC;
                done = true;
            }
        }
        const bool have_jtj = !done; // J^T J of the start point exists (the fit's matrices are valid from here on)
        if (!done)
        {
            scale(true);
            double Dx2 = 0.0, mx = -1.0;
            for (int j = 0; j < p; ++j)
            {
                const double u = diag[j] * x[j];
                Dx2 += u * u;
                mx = fmax(mx, sqrt(djj[j]) / diag[j]);
            }
            delta = 0.3 * fmax(1.0, sqrt(Dx2));
            mu = 1.0e-3 * mx * mx;
            nu = 2.0;
            avratio = 0.0;
            chisq_init = fnorm2;
            chisq0 = chisq1 = (chisq_in == chisq_in) ? chisq_in : fnorm2; // (lm_core.hpp: the same rule)
            niter = 0;
            bad_steps = 0;
            trace_row(0, chisq_init);
        }
        // Round 5, models evaluated by a device kernel (formulas), algorithm lm, device solve: ONE host synchronisation per
        // trial step.  Behind the back substitution, on the solve's stream: x + dx (bounds applied), the model there, its
        // weighted residual with the partial sums of ||f||^2, the rows of dx^T (J^T J) dx -- published with the solution.
        // The host adds the partial sums in the order sum_parts / quad_device add them: the same numbers as the stepwise path.
        struct TrialTail
        {
            BdFit *fit;
            double delta;
            int has_bounds, g;
        } tt{this, 0.0, prm.has_bounds ? 1 : 0, grid_n()};
        const bool fused_trial = dev_solve && prm.trs == 0 && model->theta_on_device() && !getenv("GSLNLS_BD_STEPWISE");
        if (fused_trial)
            for (int k = 0; k < p; ++k)
            {
                h_xmap[k] = x[k];
                h_xmap[p + k] = lo[k];
                h_xmap[2 * (size_t)p + k] = up[k];
            }
        auto tail_enqueue = [](void *ctx, void *stream, const double *d_sol) {
            TrialTail &t = *static_cast<TrialTail *>(ctx);
            BdFit &f = *t.fit;
            hipStream_t s2 = (hipStream_t)stream;
            const int p = f.p;
            double *d_xt = f.d_pub, *d_quad = f.d_pub + p, *d_parts = f.d_pub + 2 * (size_t)p;
            // two launches: [trial point | rows of the quadratic form], then [model value, residual, partial sums] -- four until
            // round 5's kernel timeline (profiles/r05_matrix_step_timeline.txt: 5 us each); a model without the fused
            // residual kernel keeps values_dev + bd_resid_kernel
            hipLaunchKernelGGL(bd_trial_quad_kernel, dim3(p), dim3(BD_T), 0, s2, f.d_xmap, d_sol, f.d_xmap + p, f.d_xmap + 2 * (size_t)p, t.delta,
                               t.has_bounds, p, d_xt, f.d_C, d_quad);
            if (f.model->resid_dev(d_xt, f.d_y, f.d_sw, f.d_f[f.cur ^ 1], d_parts, t.g, s2) < 0)
            {
                (void)f.model->values_dev(d_xt, f.d_fval, s2);
                hipLaunchKernelGGL(bd_resid_kernel, dim3(t.g), dim3(BD_T), 0, s2, f.d_fval, f.d_y, f.d_sw, (long long)f.n, f.d_f[f.cur ^ 1], d_parts);
            }
        };
        std::vector<double> pub(fused_trial ? (size_t)2 * p + tt.g : 0);
        double prof_solve = 0.0, prof_jac = 0.0, prof_resid = 0.0;
        long prof_njac = 0;
        // ---- driver2 / trust_iterate_lu_LD: one pass of the loop = one trial step ----
        long steps = 0;
        const long max_steps = ((long)maxiter * 17 + 2) * (prm.trs ? 2 : 1) + 2;
        while (!done)
        {
            if (g_interrupt_hook && g_interrupt_hook())
                return GSLNLS_E_INTERRUPTED;
            if (++steps > max_steps)
                return GSLNLS_FAILURE;
            // lm_begin_step: velocity of the damped system
            for (int k = 0; k < p; ++k)
                rhs[k] = -g[k];
            bool trial_done = false; // the fused tail delivered x + dx, ||f(x + dx)||^2 and dx^T J^T J dx with the solution
            const double t_s0 = now_s();
            if (fused_trial)
            {
                tt.delta = delta;
                MCholTail tail;
                tail.enqueue = tail_enqueue;
                tail.ctx = &tt;
                tail.extra_dev = d_pub;
                tail.extra_n = 2 * p + tt.g;
                tail.extra_host = pub.data();
                int valid = 0;
                vel.assign(p, 0.0);
                rc = mchol_device_solve_resident_tail(p, d_C, diag.data(), mu, rhs.data(), vel.data(), &tail, &valid);
                if (rc == GSLNLS_E_UNSUPPORTED)
                    rc = damped_solve(rhs, vel); // (a size the device routine does not take)
                if (rc)
                    return rc;
                trial_done = valid != 0;
            }
            else if ((rc = damped_solve(rhs, vel)))
                return rc;
            prof_solve += now_s() - t_s0;
            if (prm.trs == 1)
            {
                // geodesic acceleration (src/trust.c:252-286)
                double badv = 0.0;
                rc = fvv_at(x.data(), vel.data(), prm, gfvv.data(), &badv);
                if (rc == GSLNLS_EBADFUNC)
                    badv = 1.0;
                else if (rc)
                    return rc;
                if (prm.fvv_analytic)
                    nevalfvv += 1;
                else
                    nevalf += 1;
                if (prm.fvv_analytic && badv != 0.0)
                {
                    // a failed fvv counts as a rejected step (src/trust.c:452-483, :530-545)
                    delta /= prm.factor_down;
                    lmd_nielsen_reject(mu, nu);
                    const int itstatus = (++bad_steps > LMD_MAX_REJECTS) ? ST_ENOPROG : ST_CONTINUE;
                    if (itstatus != ST_CONTINUE)
                        (void)end_iteration(itstatus);
                    continue;
                }
                for (int k = 0; k < p; ++k)
                    rhs[k] = -gfvv[k];
                if ((rc = damped_solve(rhs, acc)))
                    return rc;
                double an = 0.0, vn = 0.0;
                for (int k = 0; k < p; ++k)
                {
                    an += acc[k] * acc[k];
                    vn += vel[k] * vel[k];
                }
                avratio = sqrt(an) / sqrt(vn);
                for (int k = 0; k < p; ++k)
                    dx[k] = vel[k] + 0.5 * acc[k];
            }
            else
                for (int k = 0; k < p; ++k)
                {
                    acc[k] = 0.0;
                    dx[k] = vel[k];
                }
            // trust_trial_step_lu (src/trust.c:9-32)
            for (int k = 0; k < p; ++k)
            {
                double t = x[k] + dx[k];
                if (prm.has_bounds)
                {
                    if (t < lo[k])
                        t = x[k] + (dx[k] / fmax(fabs(dx[k]), delta) * fabs(x[k] - lo[k]));
                    else if (t > up[k])
                        t = x[k] + (dx[k] / fmax(fabs(dx[k]), delta) * fabs(x[k] - up[k]));
                }
                xt[k] = t;
            }
            // trust_eval_step + radius / mu updates (src/trust.c:474-545)
            double ssr_t = INFINITY, vAv_tail = 0.0;
            if (trial_done)
            {
                // (x + dx as the device formed it: the same expression on the same values; the sums in sum_parts' order)
                for (int k = 0; k < p; ++k)
                    xt[k] = pub[k];
                double s0 = 0.0;
                for (int k = 0; k < tt.g; ++k)
                    s0 += pub[2 * (size_t)p + k];
                ssr_t = s0;
                for (int i = 0; i < p; ++i)
                    vAv_tail += pub[(size_t)p + i];
            }
            else
            {
                const double t_r0 = now_s();
                rc = resid_at(xt.data(), d_f[cur ^ 1], &ssr_t);
                prof_resid += now_s() - t_r0;
                if (rc == GSLNLS_EBADFUNC)
                    ssr_t = INFINITY;
                else if (rc)
                    return rc;
            }
            nevalf += 1;
            double rho;
            if (!(ssr_t < fnorm2))
                rho = -1.0;
            else
            {
                double vAv = 0.0, Dv2 = 0.0;
                if (trial_done)
                    vAv = vAv_tail;
                else if (dev_solve)
                {
                    if ((rc = quad_device(vel.data(), &vAv)))
                        return rc;
                }
                else
                    for (int i = 0; i < p; ++i)
                    {
                        double row = 0.0;
                        for (int j = 0; j < p; ++j)
                            row += A[(size_t)i * p + j] * vel[j];
                        vAv += row * vel[i];
                    }
                for (int i = 0; i < p; ++i)
                {
                    const double ud = diag[i] * vel[i];
                    Dv2 += ud * ud;
                }
                rho = lmd_rho_of(ssr_t, fnorm2, vAv, Dv2, mu);
            }
            const bool found = lmd_step_found(rho, prm.trs, avratio, prm.avmax);
            lmd_radius(rho, prm.factor_up, prm.factor_down, delta);
            int itstatus = ST_CONTINUE;
            if (found)
            {
                itstatus = ST_SUCCESS;
                const double t_j0 = now_s();
                jac_theta_dev = trial_done ? d_pub : nullptr; // (x + dx as bd_trial_kernel left it: the bits xt was read from)
                rc = jac_at(xt.data(), d_f[cur ^ 1], prm, gt.data(), djjt.data(), &badj, dev_solve ? nullptr : A.data());
                jac_theta_dev = nullptr;
                prof_jac += now_s() - t_j0;
                prof_njac += 1;
                if (rc == GSLNLS_EBADFUNC)
                    badj = 1.0;
                else if (rc)
                    return rc;
                if (prm.jac_analytic)
                    nevaldf += 1;
                else
                    nevalf += lm_fd_cost(prm, p);
                // a failed eval_df -- a non-finite analytic entry, or the closure failing at one of the difference points
                // (forward_jac_LD / center_jac_LD return eval_f's status, src/fdjac.c:46-48, :104-112) -- ends the iteration with
                // that status (src/trust.c:496-510), whatever the Jacobian's kind
                if (badj != 0.0)
                    itstatus = ST_EBADFUNC;
                if (itstatus == ST_SUCCESS)
                {
                    x = xt;
                    if (fused_trial)
                        memcpy(h_xmap, x.data(), sizeof(double) * p); // (read in place by the next trial step's kernel)
                    g = gt;
                    djj = djjt;
                    cur ^= 1;
                    fnorm2 = ssr_t;
                    scale(false);
                    lmd_nielsen_accept(rho, mu, nu);
                    bad_steps = 0;
                }
            }
            else
            {
                lmd_nielsen_reject(mu, nu);
                if (++bad_steps > LMD_MAX_REJECTS)
                    itstatus = ST_ENOPROG;
            }
            if (itstatus != ST_CONTINUE)
                (void)end_iteration(itstatus);
        }
        const double loop_ms = 1e3 * (now_s() - t_begin);
        end_diag = diag;
        end_chisq0 = chisq0;
        end_chisq1 = chisq1;
        end_niter = niter;
        end_status = status;
        if (point_fit)
        {
            // what the multi-start driver takes from the workspace (src/nls_mstart.c:91-95): nothing n-sized, and
            // det_cholesky_jtj of the Jacobian the solver holds -- the one of the last accepted point
            end_det = 0.0;
            if (have_jtj)
            {
                std::vector<double> M;
                if (dev_solve)
                {
                    M.resize((size_t)p * p);
                    GSLNLS_HIP_OK(hipMemcpy(M.data(), d_C, sizeof(double) * (size_t)p * p, hipMemcpyDeviceToHost));
                }
                else
                    M = A;
                end_det = det_chol(M);
            }
            last_x = x;
            last_f = d_f[cur];
            out->niter = niter;
            out->conv = status;
            out->info = info;
            out->ssr = chisq1;
            out->ssrtol = chisq0 - chisq1;
            out->neval[0] = (int)nevalf;
            out->neval[1] = (int)nevaldf;
            out->neval[2] = (int)nevalfvv;
            out->chisq_init = chisq_init;
            out->loop_ms = (float)loop_ms;
            out->n_steps = (int)steps;
            out->code_path = 4;
            return status;
        }
        // ---- result (src/nls.c:648-753) ----
        const bool ok = (status == ST_SUCCESS || status == ST_EMAXITER);
        const double t_res0 = now_s();
        double t_res1 = t_res0, t_res2 = t_res0;
        for (int k = 0; k < p; ++k)
            if (out->par)
                out->par[k] = ok ? x[k] : start[k];
        // NOTE: after a failed last Jacobian (EBADFUNC at an accepted point) d_J / d_C hold that point's matrices; the
        // result is NaN-filled then, as the reference's is
        // round 5: both from the device when the damped solve runs there (device_epilogue above); the host routines below
        // when it does not, or when the natural-order factorisation refuses the matrix (GSLNLS_BD_HOST_EPILOGUE=1 forces them)
        bool epi_dev = false;
        double cond_dev = NAN;
        if (ok && dev_solve && !getenv("GSLNLS_BD_HOST_EPILOGUE"))
            epi_dev = device_epilogue(out->covar, &cond_dev) == GSLNLS_SUCCESS;
        std::vector<double> Afin; // J^T J at the final point: covariance and the solver-routing diagnostic
        if (ok && !epi_dev)
        {
            Afin.resize((size_t)p * p);
            if (dev_solve)
                GSLNLS_HIP_OK(hipMemcpy(Afin.data(), d_C, sizeof(double) * (size_t)p * p, hipMemcpyDeviceToHost));
            else
                Afin = A;
        }
        if (out->covar && !epi_dev)
        {
            bool good = ok;
            std::vector<double> Ci;
            if (good)
            {
                Ci = Afin;
                good = lg_chol(p, Ci);
                if (good)
                    lg_chol_invert(p, Ci);
            }
            for (int i = 0; i < p; ++i)
                for (int j = 0; j < p; ++j)
                    out->covar[i + (size_t)p * j] = good ? Ci[(size_t)i * p + j] : NAN;
        }
        t_res1 = now_s();
        if (out->resid)
        {
            if (ok)
                GSLNLS_HIP_OK(hipMemcpy(out->resid, d_f[cur], sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
            else
                for (int i = 0; i < n; ++i)
                    out->resid[i] = NAN;
        }
        if (out->grad)
        {
            if (ok)
                GSLNLS_HIP_OK(hipMemcpy(out->grad, d_J, sizeof(double) * (size_t)n * p, hipMemcpyDeviceToHost));
            else
                for (size_t i = 0; i < (size_t)n * p; ++i)
                    out->grad[i] = NAN;
        }
        t_res2 = now_s();
        last_x = x;
        last_f = d_f[cur];
        out->niter = niter;
        out->conv = status;
        out->info = info;
        out->ssr = chisq1;
        out->ssrtol = chisq0 - chisq1;
        out->neval[0] = (int)nevalf;
        out->neval[1] = (int)nevaldf;
        out->neval[2] = (int)nevalfvv;
        out->chisq_init = chisq_init;
        out->loop_ms = (float)loop_ms;
        out->n_launches = 0;
        out->n_steps = (int)steps;
        out->jtj_cond = !ok ? NAN : (epi_dev ? cond_dev : bd_scaled_cond(p, Afin));
        if (!point_fit)
        {
            g_bd_prof.loop_ms = loop_ms;
            g_bd_prof.solve_ms = 1e3 * prof_solve;
            g_bd_prof.jac_ms = 1e3 * prof_jac;
            g_bd_prof.resid_ms = 1e3 * prof_resid;
            g_bd_prof.covar_ms = 1e3 * (t_res1 - t_res0);
            g_bd_prof.down_ms = 1e3 * (t_res2 - t_res1);
            g_bd_prof.cond_ms = 1e3 * (now_s() - t_res2);
            g_bd_prof.trial_steps = steps;
            g_bd_prof.jacobians = prof_njac;
            g_bd_prof.fused = fused_trial ? 1 : 0;
            g_bd_prof.p = p;
            g_bd_prof.n = n;
        }
        if (getenv("GSLNLS_LARGE_PROF"))
            fprintf(stderr, "[bd] p = %d n = %d: loop %.2f ms (%ld trial steps), covariance %.2f ms, resid + grad down %.2f ms, condition %.2f ms\n", p, n,
                    loop_ms, steps, 1e3 * (t_res1 - t_res0), 1e3 * (t_res2 - t_res1), 1e3 * (now_s() - t_res2));
        out->code_path = 4;
        return status;
    }

    // gsl_multifit_nlinear_rho_driver (src/nls_irls.c:412-546) around the matrix-path solve: the driver of wide_host.hpp /
    // irls_host.hpp (cold re-start from the ORIGINAL start with the current weights, radix-select median of |r| on the
    // device, psi family, n / sum(w) scaling, user weights multiplied in, test_delta_irls).  The unweighted residual is
    // the last solve's weighted one divided by its weights, as the reference takes it (no extra evaluation of the model).
    int irls(int jac, int fvv, const double *start, const double *lupars, const int *ci, const double *cd, int loss_rho,
             const double *loss_cc, gslnls_result *out)
    {
        if (ci[2] > 1)
            return GSLNLS_E_UNSUPPORTED;
        const int irls_maxiter = ci[14];
        const double irls_xtol = cd[10];
        LossCfg Lc;
        Lc.rho = loss_rho;
        {
            static const int ncc[9] = {0, 1, 2, 1, 1, 1, 1, 3, 3};
            for (int k = 0; k < 3; ++k)
                Lc.cc[k] = (loss_rho >= 1 && loss_rho <= 8 && k < ncc[loss_rho]) ? loss_cc[k] : 0.0;
        }
        const size_t nb = sizeof(double) * (size_t)n;
        constexpr int TW = 256;
        int nblk = (int)(((long long)n + TW - 1) / TW);
        if (nblk > 1024)
            nblk = 1024;
        auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
        const size_t need = 6 * up(nb) + up(sizeof(unsigned long long) * (size_t)n) + up(sizeof(double) * nblk) +
                            up(sizeof(SelectState) * 2) + up(sizeof(IrlsScalars));
        if (irls_arena_bytes < need)
        {
            (void)hipFree(irls_arena);
            irls_arena = nullptr;
            irls_arena_bytes = 0;
            GSLNLS_HIP_OK(hipMalloc(&irls_arena, need));
            irls_arena_bytes = need;
        }
        char *q = static_cast<char *>(irls_arena);
        auto take = [&](size_t b) {
            char *r = q;
            q += up(b);
            return r;
        };
        double *d_r = reinterpret_cast<double *>(take(nb)), *d_wt = reinterpret_cast<double *>(take(nb)),
               *d_psi = reinterpret_cast<double *>(take(nb)), *d_psip = reinterpret_cast<double *>(take(nb)),
               *d_swA = reinterpret_cast<double *>(take(nb)), *d_swB = reinterpret_cast<double *>(take(nb));
        unsigned long long *d_keys = reinterpret_cast<unsigned long long *>(take(sizeof(unsigned long long) * (size_t)n));
        double *d_pw = reinterpret_cast<double *>(take(sizeof(double) * nblk));
        SelectState *d_sel = reinterpret_cast<SelectState *>(take(sizeof(SelectState) * 2));
        IrlsScalars *d_sc = reinterpret_cast<IrlsScalars *>(take(sizeof(IrlsScalars)));
        double *const user_sw = d_sw;
        // the fit's own weights are put back however this function is left
        struct RestoreSw
        {
            double *&ref;
            double *val;
            ~RestoreSw() { ref = val; }
        } restore_sw{d_sw, user_sw};
        double *sw_now = d_swA, *sw_next = d_swB;
        if (user_sw)
            GSLNLS_HIP_OK(hipMemcpyAsync(sw_now, user_sw, nb, hipMemcpyDeviceToDevice, st));
        else
        {
            std::vector<double> ones((size_t)n, 1.0);
            GSLNLS_HIP_OK(hipMemcpy(sw_now, ones.data(), nb, hipMemcpyHostToDevice));
        }
        std::vector<double> workp(p);
        int irls_iter = 0, irls_status = ST_FAILURE, status = ST_CONTINUE;
        double chisq_init = NAN, chisq_carry = NAN, sigma = 1.0;
        long long nf = 0, ndf = 0, nfvv = 0;
        double loop_ms = 0.0;
        const int gf = (int)std::min<long long>(2048, ((long long)n + 255) / 256);
        do
        {
            irls_iter += 1;
            if (irls_iter > 1)
            {
                std::swap(sw_now, sw_next);
                workp = last_x;
            }
            else
                workp.assign(start, start + p);
            d_sw = sw_now;
            const int rc = solve(jac, fvv, start, lupars, ci, cd, out, irls_iter > 1 ? chisq_carry : NAN);
            if (rc < 0 && rc != ST_EBADFUNC)
                return rc; // (a library error, not a GSL status)
            nf += out->neval[0];
            ndf += out->neval[1];
            nfvv += out->neval[2];
            loop_ms += out->loop_ms;
            status = out->conv;
            if (irls_iter == 1)
                chisq_init = out->chisq_init;
            chisq_carry = out->ssr;
            trace_printf("IRLS iter: %3d, weighted ssr: %g, par: (", irls_iter, out->ssr); // (src/nls_irls.c:466-472)
            trace_vector(last_x.data(), p);
            if (status == ST_EBADFUNC || (status == ST_ENOPROG && irls_iter == 1))
                break;
            // ---- re-weighting chain, all on the device ----
            hipLaunchKernelGGL(bd_unweight_keys_kernel, dim3(gf), dim3(256), 0, st, last_f, sw_now, (long long)n, d_r, d_keys);
            const unsigned long long k_lo = (unsigned long long)((n - 1) / 2), k_hi = (unsigned long long)(n / 2);
            const int nsel = (k_lo == k_hi) ? 1 : 2;
            for (int which = 0; which < nsel; ++which)
            {
                SelectState *ss = d_sel + which;
                hipLaunchKernelGGL(select_init_kernel, dim3(1), dim3(1), 0, st, ss, which == 0 ? k_lo : k_hi);
                for (int pass = 7; pass >= 0; --pass)
                {
                    hipLaunchKernelGGL(select_hist_kernel, dim3(std::min(1024, (int)((n + 255) / 256))), dim3(256), 0, st, d_keys,
                                       (long long)n, pass, ss);
                    hipLaunchKernelGGL(select_pick_kernel, dim3(1), dim3(1), 0, st, pass, ss);
                }
            }
            hipLaunchKernelGGL(irls_sigma_kernel, dim3(1), dim3(1), 0, st, d_sel, d_sel + (nsel - 1), d_sc);
            hipLaunchKernelGGL((irls_weight_kernel<TW>), dim3(nblk), dim3(TW), 0, st, d_r, (long long)n, Lc, d_sc, d_wt, d_psi,
                               d_psip, d_pw);
            hipLaunchKernelGGL(irls_scale_kernel, dim3(1), dim3(1), 0, st, d_pw, nblk, (long long)n, d_sc);
            hipLaunchKernelGGL((irls_apply_kernel<TW>), dim3(nblk), dim3(TW), 0, st, d_wt, (long long)n, d_sc, user_sw, sw_next);
            IrlsScalars hsc;
            GSLNLS_HIP_OK(hipMemcpyAsync(&hsc, d_sc, sizeof(hsc), hipMemcpyDeviceToHost, st));
            GSLNLS_HIP_OK(hipStreamSynchronize(st));
            sigma = hsc.sigma;
            // test_delta_irls (src/nls_irls.c:343-362)
            irls_status = ST_CONTINUE;
            for (int k = 0; k < p; ++k)
            {
                const double xi = last_x[k], dxi = fabs(workp[k] - xi);
                if (fmin(dxi / fabs(xi), dxi) < irls_xtol)
                    irls_status = ST_SUCCESS;
                else
                {
                    irls_status = ST_CONTINUE;
                    break;
                }
            }
            if (irls_status == ST_SUCCESS)
                break;
        } while (irls_status == ST_CONTINUE && irls_iter < irls_maxiter);
        if (!(status == ST_EBADFUNC || (status == ST_ENOPROG && irls_iter == 1)))
        {
            if (irls_iter >= irls_maxiter && irls_status != ST_SUCCESS)
            {
                irls_status = ST_EMAXITER;
                status = ST_EMAXITER;
                out->conv = ST_EMAXITER;
                out->info = ST_EMAXITER;
            }
        }
        // (resid / grad / covar in `out` are the LAST solve's, with its weights: src/nls.c:695-737)
        out->chisq_init = chisq_init;
        out->neval[0] = (int)nf;
        out->neval[1] = (int)ndf;
        out->neval[2] = (int)nfvv;
        out->loop_ms = (float)loop_ms;
        const bool ok = (status == ST_SUCCESS || status == ST_EMAXITER);
        double irls_delta = 0.0;
        for (int k = 0; k < p; ++k)
            irls_delta = fmax(irls_delta, fabs(workp[k] - last_x[k]));
        out->irls_sigma = sigma;
        out->irls_status = irls_status;
        out->irls_niter = irls_iter;
        out->irls_tol = irls_delta;
        if (ok)
        {
            if (out->irls_weights)
                GSLNLS_HIP_OK(hipMemcpy(out->irls_weights, d_wt, nb, hipMemcpyDeviceToHost));
            if (out->irls_psi)
                GSLNLS_HIP_OK(hipMemcpy(out->irls_psi, d_psi, nb, hipMemcpyDeviceToHost));
            if (out->irls_dpsi)
                GSLNLS_HIP_OK(hipMemcpy(out->irls_dpsi, d_psip, nb, hipMemcpyDeviceToHost));
        }
        else
            for (int i = 0; i < n; ++i)
            {
                if (out->irls_weights)
                    out->irls_weights[i] = NAN;
                if (out->irls_psi)
                    out->irls_psi[i] = NAN;
                if (out->irls_dpsi)
                    out->irls_dpsi[i] = NAN;
            }
        return status;
    }
};

// ---- multi-start on the matrix path (round 5) ------------------------------------------------------------------------
// The per-point work of gsl_multistart_driver (src/nls_mstart.c:42-128, :236-349) for function models and for formulas
// beyond 64 parameters: quasi-random point -> sampling range, det_eval_jtj, the short fit, det_cholesky_jtj again -- one
// point after the other (a closure is evaluated on the calling thread, one parameter vector at a time, exactly as the
// reference's loop does), every n x p and p x p operation of each of them on the device.  The record layout is the one of
// batch_core.hpp (MsRecord), so the commit of mstart_driver.hpp replays the reference's loop from these records as it
// does for the lane kernel and the wide path.
struct BdMsEvaluator : MsEvaluator
{
    BdFit &fit;
    SobolTable tab;
    int jac = 0, fvv = 0;
    const int *ci = nullptr;
    const double *cd = nullptr;
    const double *lupars = nullptr;
    std::vector<double> staged;
    explicit BdMsEvaluator(BdFit &f) : fit(f) { sobol_build(tab, f.p); }

    int run_host(MsBatch &b, int lo, int hi, double *out)
    {
        const int p = fit.p, K = 3 * p + 8;
        if (b.p != p || b.K != K)
            return GSLNLS_EINVAL;
        std::vector<double> st(p);
        gslnls_result scratch;
        for (int idx = lo; idx < hi; ++idx)
        {
            if (g_interrupt_hook && g_interrupt_hook())
                return GSLNLS_E_INTERRUPTED;
            for (int k = 0; k < p; ++k)
                st[k] = b.draw[idx] >= 0 ? sobol_to_range(sobol_coord(tab, (unsigned int)b.draw[idx], k), b.range[2 * k],
                                                           b.range[2 * k + 1], b.kd[k])
                                         : b.start[(size_t)idx * p + k];
            double *rec = out + (size_t)(idx - lo) * K;
            double *rx = rec, *rdiag = rec + p, *rx0 = rec + 2 * p, *sc = rec + 3 * p;
            double det0 = 0.0, ssr0 = INFINITY;
            int rc = fit.det_at(st.data(), jac, ci, cd, &det0, &ssr0);
            if (rc)
                return rc;
            for (int k = 0; k < p; ++k)
                rx0[k] = st[k];
            sc[2] = det0;
            sc[4] = ssr0;
            if (b.always_fit || det0 > b.dtol)
            {
                memset(&scratch, 0, sizeof(scratch));
                rc = fit.solve(jac, fvv, st.data(), lupars, ci, cd, &scratch, NAN, b.maxiter);
                if (rc < 0)
                    return rc; // (a library error or the step guard, not a GSL status: the end state was not written)
                for (int k = 0; k < p; ++k)
                {
                    rx[k] = fit.last_x[k];
                    rdiag[k] = fit.end_diag[k];
                }
                sc[0] = fit.end_chisq0;
                sc[1] = fit.end_chisq1;
                sc[3] = fit.end_det;
                sc[5] = (double)fit.end_niter;
                sc[6] = (double)fit.end_status;
                sc[7] = (double)fit.nevalf;
            }
            else
            {
                for (int k = 0; k < p; ++k)
                {
                    rx[k] = st[k];
                    rdiag[k] = 1.0;
                }
                sc[0] = INFINITY;
                sc[1] = ssr0;
                sc[3] = 0.0;
                sc[5] = 0.0;
                sc[6] = (double)ST_CONTINUE;
                sc[7] = 1.0;
            }
        }
        return 0;
    }
    int run_async(MsBatch &b, int lo, int hi, double *dev_out) override
    {
        const size_t nd = (size_t)(hi - lo) * (3 * fit.p + 8);
        staged.assign(nd, 0.0);
        const int rc = run_host(b, lo, hi, staged.data());
        if (rc)
            return rc;
        // (`staged` lives as long as the evaluator: the copy is complete before the collective that follows it on the stream)
        GSLNLS_HIP_OK(hipMemcpyAsync(dev_out, staged.data(), sizeof(double) * nd, hipMemcpyHostToDevice, fit.st));
        return 0;
    }
    int run(MsBatch &b, int lo, int hi, double *out, bool out_on_device) override
    {
        if (!out)
            return GSLNLS_E_UNSUPPORTED; // (records are produced on the host: somebody has to take them)
        if (!out_on_device)
            return run_host(b, lo, hi, out);
        const int rc = run_async(b, lo, hi, out);
        if (rc)
            return rc;
        GSLNLS_HIP_OK(hipStreamSynchronize(fit.st));
        return 0;
    }
    void *stream() override { return (void *)fit.st; }
    int fetch(const double *src, bool src_on_device, double *dst, size_t nd) override
    {
        if (src_on_device)
            GSLNLS_HIP_OK(hipMemcpy(dst, src, sizeof(double) * nd, hipMemcpyDeviceToHost));
        else
            memcpy(dst, src, sizeof(double) * nd);
        return 0;
    }
    int fetch_stream(const double *dev_src, double *dst, size_t nd) override
    {
        if (nd)
            GSLNLS_HIP_OK(hipMemcpyAsync(dst, dev_src, sizeof(double) * nd, hipMemcpyDeviceToHost, fit.st));
        GSLNLS_HIP_OK(hipStreamSynchronize(fit.st));
        return 0;
    }
    int poke(double *dev_dst, double value) override
    {
        GSLNLS_HIP_OK(hipMemcpyAsync(dev_dst, &value, sizeof(double), hipMemcpyHostToDevice, fit.st));
        GSLNLS_HIP_OK(hipStreamSynchronize(fit.st));
        return 0;
    }
};

// robust second pass (src/nls.c:401-443): Cook's distances at the first pass's optimum from the resident residual and
// Jacobian; observations with D_i > min(4 / n, 5 MAD(D)) get weight zero.  1: a second pass has to run (d_sw_robust
// filled), 0: not (also when hat_values fails: J^T J singular), < 0: error
inline int bd_robust_weights(BdFit &fit, int jac, const double *mpopt, const int *ci, const double *cd, double *d_sw_robust)
{
    const int n = fit.n, p = fit.p;
    const LmParams prm = make_params(ci, cd, jac, 0, false, fit.d_sw != nullptr);
    double ssr = INFINITY, badj = 0.0;
    int rc = fit.resid_at(mpopt, fit.d_f[fit.cur], &ssr);
    if (rc == GSLNLS_EBADFUNC)
        return 0;
    if (rc)
        return rc < 0 ? rc : -1;
    std::vector<double> gtmp(p), dtmp(p), A((size_t)p * p);
    rc = fit.jac_at(mpopt, fit.d_f[fit.cur], prm, gtmp.data(), dtmp.data(), &badj, A.data());
    if (rc == GSLNLS_EBADFUNC || badj != 0.0)
        return 0;
    if (rc)
        return rc < 0 ? rc : -1;
    if (!lg_chol(p, A)) // cooks_d -> hat_values fails: no second pass (src/nls.c:419-421)
        return 0;
    lg_chol_invert(p, A);
    const double s2 = ssr / (n - p);
    hipStream_t st = fit.st;
    // (J^T J)^-1 takes the place of J^T J on the device: the next evaluation of a Jacobian rewrites it
    GSLNLS_HIP_OK(hipMemcpyAsync(fit.d_C, A.data(), sizeof(double) * (size_t)p * p, hipMemcpyHostToDevice, st));
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t nb8 = up(sizeof(double) * (size_t)n);
    char *arena = nullptr;
    GSLNLS_HIP_OK(hipMalloc(&arena, 2 * nb8 + up(sizeof(SelectState) * 2) + 256));
    double *d_d = reinterpret_cast<double *>(arena);
    unsigned long long *d_keys = reinterpret_cast<unsigned long long *>(arena + nb8);
    SelectState *d_sel = reinterpret_cast<SelectState *>(arena + 2 * nb8);
    int *d_cnt = reinterpret_cast<int *>(arena + 2 * nb8 + up(sizeof(SelectState) * 2));
    if (hipMemsetAsync(d_cnt, 0, sizeof(int), st) != hipSuccess)
    {
        (void)hipFree(arena);
        return GSLNLS_E_NODEVICE;
    }
    int gf = (int)(((long long)n + 255) / 256);
    gf = gf > 2048 ? 2048 : (gf < 1 ? 1 : gf);
    hipLaunchKernelGGL(bd_cooks_kernel, dim3(gf), dim3(BD_T), 0, st, fit.d_f[fit.cur], fit.d_J, (long long)n, p, fit.d_C, s2, d_d,
                       d_keys, (double *)nullptr);
    double med = 0.0, med2 = 0.0;
    rc = device_median(st, d_keys, n, d_sel, &med);
    if (!rc)
    {
        hipLaunchKernelGGL(absdev_keys_kernel, dim3(gf), dim3(256), 0, st, d_d, (long long)n, med, d_keys);
        rc = device_median(st, d_keys, n, d_sel, &med2);
    }
    int noutlier = 0;
    if (!rc)
    {
        const double mad = 1.482602218505602 * med2;
        const double thresh = fmin(4.0 / n, 5 * mad);
        hipLaunchKernelGGL(outlier_weights_kernel, dim3(gf), dim3(256), 0, st, d_d, (long long)n, thresh, fit.d_sw, d_sw_robust, d_cnt);
        (void)hipMemcpyAsync(&noutlier, d_cnt, sizeof(int), hipMemcpyDeviceToHost, st);
        (void)hipStreamSynchronize(st);
    }
    (void)hipFree(arena);
    if (rc)
        return rc < 0 ? rc : -1;
    return (noutlier > 0 && noutlier < (n - p)) ? 1 : 0;
}

// multi-start branch of C_nls (src/nls.c:274-532) on the matrix path: the host driver of mstart_driver.hpp (the
// reference's commit order) around BdMsEvaluator, the robust second pass when a loss function is set, the final solve
inline int bd_mstart(BdFit &fit, int jac, int fvv, const double *start2p, const double *lupars, const int *ci, const double *cd,
                     const int *has_start, const MsComm &comm, int loss_rho, const double *loss_cc, gslnls_result *out)
{
    if (ci[2] > 1)
        return GSLNLS_E_UNSUPPORTED;
    if ((jac && !fit.model->has_jac) || (fvv && !fit.model->has_fvv))
        return GSLNLS_EINVAL;
    if (ci[6] < 1 || ci[8] < 0 || ci[8] > ci[6])
        return GSLNLS_EINVAL;
    const int p = fit.p;
    MsState m;
    ms_init(m, p, ci, cd, start2p, has_start, lupars);
    BdMsEvaluator ev(fit);
    ev.jac = jac;
    ev.fvv = fvv;
    ev.ci = ci;
    ev.cd = cd;
    ev.lupars = lupars;
    int rc = ms_major_loop(m, ev, comm, start2p);
    if (rc)
        return rc < 0 && rc > -100 ? GSLNLS_FAILURE : rc;
    if (loss_rho != 0)
    {
        if (m.mssropt[1] < m.mssropt[0])
            m.mpopt = m.mpopt1;
        double *d_sw_robust = nullptr;
        GSLNLS_HIP_OK(hipMalloc(&d_sw_robust, sizeof(double) * (size_t)fit.n));
        const int second = bd_robust_weights(fit, jac, m.mpopt.data(), ci, cd, d_sw_robust);
        if (second < 0)
        {
            (void)hipFree(d_sw_robust);
            return second;
        }
        if (second == 1)
        {
            double *keep_sw = fit.d_sw;
            fit.d_sw = d_sw_robust;
            m.next_draw = 0; // gsl_qrng_init
            m.second_pass = true;
            m.mstop = ST_CONTINUE;
            m.mstarts = m.nsp = m.nwsp = 0;
            m.dtol = 1.0e-6;
            m.rejectscl = 1.25;
            m.mssropt[0] = m.mssropt[1] = INFINITY;
            m.ssrconv[0] = m.ssrconv[1] = 1.0;
            std::fill(m.ntix.begin(), m.ntix.end(), 0);
            std::fill(m.luchange.begin(), m.luchange.end(), 0);
            rc = ms_major_loop(m, ev, comm, start2p);
            fit.d_sw = keep_sw; // "reset original weights" (src/nls.c:490-507)
        }
        (void)hipFree(d_sw_robust);
        if (rc)
            return rc < 0 && rc > -100 ? GSLNLS_FAILURE : rc;
    }
    ms_trace_finished(m);
    // src/nls.c:518-531
    if (m.mssropt[1] < m.mssropt[0])
    {
        m.mssropt[0] = m.mssropt[1];
        m.ssrconv[0] = m.ssrconv[1];
        m.mpopt = m.mpopt1;
    }
    const double ftol = cd[6];
    if (m.mssropt[0] < ftol || m.ssrconv[0] < ftol)
    {
        if (lupars)
            m.mpopt[0] = fmin(m.mpopt[0] + 1.0e-4, std::isfinite(lupars[1]) ? lupars[1] : INFINITY);
        else
            m.mpopt[0] = m.mpopt[0] + 1.0e-4;
    }
    if (loss_rho != 0)
        rc = fit.irls(jac, fvv, m.mpopt.data(), lupars, ci, cd, loss_rho, loss_cc, out);
    else
        rc = fit.solve(jac, fvv, m.mpopt.data(), lupars, ci, cd, out);
    out->mstart_nsp = m.nsp;
    out->mstart_nwsp = m.nwsp;
    out->mstart_iters = m.mstarts;
    out->mstart_stop = m.mstop;
    out->mstart_ssropt = m.mssropt[0];
    return rc;
}

// measurement hook: device milliseconds of one J^T J (bd_syrk_kernel + bd_syrk_reduce_kernel) on an n x p matrix of noise,
// by HIP events over `reps` repetitions; < 0 on error
inline double bd_time_syrk(int n, int p, int reps)
{
    if (n < 1 || p < 1 || p > 4096 || reps < 1)
        return -1.0;
    const BdSyrkGeom geom = bd_syrk_geom(n, p);
    double *dJ = nullptr, *dC = nullptr, *dpart = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipStream_t st = nullptr;
    double ms = -1.0;
    if (hipMalloc(&dJ, sizeof(double) * (size_t)n * p) == hipSuccess && hipMalloc(&dC, sizeof(double) * (size_t)p * p) == hipSuccess &&
        hipMalloc(&dpart, sizeof(double) * geom.scratch) == hipSuccess && hipEventCreate(&e0) == hipSuccess &&
        hipEventCreate(&e1) == hipSuccess && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess)
    {
        std::vector<double> h((size_t)n * p);
        unsigned long long sd = 88172645463325252ull;
        for (double &v : h)
        {
            sd ^= sd << 13;
            sd ^= sd >> 7;
            sd ^= sd << 17;
            v = (double)(sd >> 11) * (1.0 / 9007199254740992.0) - 0.5;
        }
        (void)hipMemcpy(dJ, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice);
        for (int r = 0; r < reps + 2; ++r)
        {
            if (r == 2)
                (void)hipEventRecord(e0, st);
            bd_syrk_launch(dJ, (long long)n, p, dC, dpart, geom, st);
        }
        (void)hipEventRecord(e1, st);
        float f = 0.f;
        if (hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&f, e0, e1) == hipSuccess)
            ms = (double)f / reps;
    }
    (void)hipGetLastError();
    if (st)
        (void)hipStreamDestroy(st);
    if (e0)
        (void)hipEventDestroy(e0);
    if (e1)
        (void)hipEventDestroy(e1);
    (void)hipFree(dJ);
    (void)hipFree(dC);
    (void)hipFree(dpart);
    return ms;
}

// host closures as the model: gslnls_nls_fn (capi.hip)
struct BdCallbackModel : BdModel
{
    int n = 0, p = 0;
    gslnls_fn_cb f = nullptr;
    gslnls_jac_cb jac = nullptr;
    gslnls_fvv_cb fv = nullptr;
    void *user = nullptr;
    std::vector<double> hbuf, hJ;
    int values(const double *theta, double *d_fval, hipStream_t st) override
    {
        hbuf.resize(n);
        if (f(theta, p, hbuf.data(), n, user))
            return 1;
        if (hipMemcpyAsync(d_fval, hbuf.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice, st) != hipSuccess)
            return 1;
        return hipStreamSynchronize(st) == hipSuccess ? 0 : 1; // (hbuf is rewritten by the next call)
    }
    int jacobian(const double *theta, double *d_J, hipStream_t st) override
    {
        hJ.resize((size_t)n * p);
        if (jac(theta, p, hJ.data(), n, user))
            return 1;
        if (hipMemcpyAsync(d_J, hJ.data(), sizeof(double) * (size_t)n * p, hipMemcpyHostToDevice, st) != hipSuccess)
            return 1;
        return hipStreamSynchronize(st) == hipSuccess ? 0 : 1;
    }
    int fvv(const double *theta, const double *v, double *d_out, hipStream_t st) override
    {
        hbuf.resize(n);
        if (fv(theta, v, p, hbuf.data(), n, user))
            return 1;
        if (hipMemcpyAsync(d_out, hbuf.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice, st) != hipSuccess)
            return 1;
        return hipStreamSynchronize(st) == hipSuccess ? 0 : 1;
    }
};

} // namespace gslnls
