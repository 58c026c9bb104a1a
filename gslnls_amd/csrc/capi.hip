// capi.hip -- the C ABI of libgslnls_hip.so (declarations: include/gslnls_core.h).
//
// gslnls_nls() is the numeric body behind .Call(C_nls) (src/nls.c:54-813); the R shim in
// integration/r_shim/ only unpacks SEXPs and packs the returned list.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/gslnls_core.h"
#include "trace_log.hpp"
#include "dense_host.hpp"
#include "mstart_host.hpp"
#include "formula.hpp"
#include "irls_host.hpp"
#include "large_host.hpp"
#include "sparse_large.hpp"
#include "irls_batch.hpp"
#include "robust_host.hpp"
#include "rccl_comm.hpp"

using namespace gslnls;

struct gslnls_dense
{
    DenseBase *impl;
};

// process-wide communicator of the multi-start sharding (one process per GPU)
static MsComm g_comm;
static RcclComm g_rccl;

namespace gslnls
{
// the multi-rank form of one batch with the in-library collective: kernel -> ncclAllGather -> D2H, all on the
// evaluator's stream; the host only enqueues and then waits once.  Collective safety: between the point where the
// ranks are known to agree (after ensure_together) and ncclAllGather there is no return -- a rank whose shard fails
// says so in the status slot of its first record, enters the collective like the others, and all ranks fail
// together after it.
int ms_rccl_run_batch(MsEvaluator &ev, const MsComm &comm, MsBatch &b, int per, int lo, int hi)
{
    RcclComm &rc = *comm.rccl;
    const size_t shard_doubles = (size_t)per * b.K;
    hipStream_t st = (hipStream_t)ev.stream();
    int e = rc.ensure_together(shard_doubles, st); // all ranks or none: the outcome is itself all-gathered
    if (e)
        return e;
    const int status_slot = 3 * b.p + 6;
    int run_rc = 0;
    if (hi > lo)
        run_rc = ev.run_async(b, lo, hi, rc.shard);
    if (run_rc && hi > lo)
        (void)ev.poke(rc.shard + status_slot, MS_SHARD_FAILED); // fail together, after the collective
    e = rc.allgather(shard_doubles, st);
    if (e)
        return e; // the collective itself could not be enqueued: RCCL is unusable for every rank of the communicator
    const double *status = nullptr;
    size_t status_stride = 0;
    if (!b.host_records)
    {
        // records stay in HBM: only the status word of every rank's first record crosses (one strided copy)
        const bool ok = hipMemcpy2DAsync(rc.h_flags, sizeof(double), rc.all + status_slot, sizeof(double) * shard_doubles,
                                         sizeof(double), (size_t)rc.world, hipMemcpyDeviceToHost, st) == hipSuccess;
        e = ev.fetch_stream(rc.all, nullptr, 0); // waits for the stream
        if (e || !ok)
            return e ? e : GSLNLS_E_NODEVICE;
        status = rc.h_flags;
        status_stride = 1;
    }
    else
    {
        e = ev.fetch_stream_view(rc.all, (size_t)b.count * b.K, b, &b.rec); // pinned destination, consumed in place
        if (e)
            return e;
        status = b.rec + status_slot;
        status_stride = shard_doubles;
    }
    for (int r = 0; r < rc.world; ++r)
        if ((long long)r * per < b.count && status[(size_t)r * status_stride] == MS_SHARD_FAILED)
            return run_rc ? run_rc : GSLNLS_FAILURE;
    return run_rc; // (0 unless this rank's failure could not even be written into its shard)
}
} // namespace gslnls

namespace gslnls
{
DenseBase *make_dense_expr(const gslnls_model *fn, const double *y, int n, const double *swts, int *err); // vm_models.hip
int bd_formula_nls(const gslnls_model *fn, const double *y, int n, int jac, int fvv, const double *start, int start_is_matrix,
                   const int *has_start, const MsComm &comm, const double *swts, const double *lupars, const int *ci,
                   const double *cd, int loss_rho, const double *loss_cc, gslnls_result *out);
int bd_callback_nls(int n, int p, const double *y, gslnls_fn_cb f, gslnls_jac_cb jac, gslnls_fvv_cb fvv, void *user,
                    const double *start, int start_is_matrix, const int *has_start, const MsComm &comm, const double *swts,
                    const double *lupars, const int *ci, const double *cd, int loss_rho, const double *loss_cc,
                    gslnls_result *out); // bd_models.hip
int bd_last_profile(double *v, int cap);     // bd_models.hip
double bd_syrk_ms(int n, int p, int reps);   // bd_models.hip
void bd_trim_pool();                          // bd_models.hip: the matrix path's parked buffers
int bd_spd_inverse(int p, const double *d_A, const double *A_host, double *covar_host); // bd_models.hip
void trim_dense_expr();                                                                                         // vm_models.hip
}

// destroyed problems of the hand-written models are parked for the next create of the same model (dense_host.hpp)
static void release_dense(DenseBase *b)
{
    if (b && !b->park())
        delete b;
}

static DenseBase *make_dense(const gslnls_model *fn, const double *y, int n, const double *swts, int *err)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    {
        fprintf(stderr, "gslnls: no HIP device available -- the MI355X path cannot run (no CPU fallback exists)\n");
        *err = GSLNLS_E_NODEVICE;
        return nullptr;
    }
    DenseBase *b = nullptr;
    int rc = GSLNLS_E_UNSUPPORTED;
#define GSLNLS_MAKE(MODEL)                                          \
    {                                                               \
        if (fn->p != MODEL::P || fn->nx != MODEL::NX)               \
        {                                                           \
            *err = GSLNLS_EINVAL;                                   \
            return nullptr;                                         \
        }                                                           \
        auto *d = fn->x_on_device ? nullptr : DenseFit<MODEL>::acquire(); \
        if (!d)                                                     \
            d = new DenseFit<MODEL>();                              \
        rc = d->init(fn, y, n, swts);                               \
        b = d;                                                      \
    }
#ifndef GSLNLS_NO_EXPR
    if (fn->id == GSLNLS_MODEL_EXPR)
        return make_dense_expr(fn, y, n, swts, err);
#endif
    switch (fn->id)
    {
    case GSLNLS_MODEL_EXPDECAY:
        GSLNLS_MAKE(ModelExpDecay);
        break;
    case GSLNLS_MODEL_MISRA1A:
        GSLNLS_MAKE(ModelMisra1a);
        break;
    case GSLNLS_MODEL_GAUSSPK:
        GSLNLS_MAKE(ModelGaussPeak);
        break;
    case GSLNLS_MODEL_GAUSS1:
        GSLNLS_MAKE(ModelGauss1);
        break;
    default:
        *err = GSLNLS_E_UNSUPPORTED;
        return nullptr;
    }
#undef GSLNLS_MAKE
    if (rc != GSLNLS_SUCCESS)
    {
        delete b;
        *err = rc;
        return nullptr;
    }
    *err = GSLNLS_SUCCESS;
    return b;
}

struct gslnls_large
{
    DenseBase *dense = nullptr; // row models: data owner
    LargeOps *ops = nullptr;
    int n = 0, p = 0;
};

template <class M>
static LargeOps *make_row_ops(DenseBase *b)
{
    return new RowLargeOps<M>(*static_cast<DenseFit<M> *>(b));
}

struct gslnls_batch
{
    int model_id = 0, p = 0, nx = 0, n = 0, B = 0;
    double *d_x = nullptr, *d_y = nullptr, *d_usw = nullptr, *d_sw = nullptr, *d_par = nullptr, *d_scal = nullptr;
    unsigned long long *d_keys = nullptr;
    unsigned long long *d_passes = nullptr; // [2] device counters of the last call, h_passes their host copy
    unsigned long long h_passes[2] = {0, 0};
    int *d_ints = nullptr;
    hipStream_t st = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
};

template <class M>
static int batch_irls_run(gslnls_batch *h, int lo, int hi, int jac, int fvv, const double *start, const double *lupars,
                          const int *ci, const double *cd, int loss_rho, const double *loss_cc, float *kernel_ms)
{
    constexpr int P = M::P;
    constexpr int T = 256;
    if (fvv && !M::HAS_FVV)
        return GSLNLS_E_UNSUPPORTED;
    IrlsBatchArgs<P> a;
    a.x = h->d_x;
    a.y = h->d_y;
    a.usw = h->d_usw;
    a.sw = h->d_sw;
    a.keys = h->d_keys;
    a.n = h->n;
    a.lo = lo;
    a.hi = hi;
    for (int k = 0; k < P; ++k)
    {
        a.start[k] = start[k];
        a.lu[2 * k] = lupars ? lupars[2 * k] : -INFINITY;
        a.lu[2 * k + 1] = lupars ? lupars[2 * k + 1] : INFINITY;
    }
    a.has_lu = lupars != nullptr;
    a.prm = make_params(ci, cd, jac, fvv, lupars != nullptr, true);
    a.loss.rho = loss_rho;
    static const int ncc[9] = {0, 1, 2, 1, 1, 1, 1, 3, 3};
    for (int k = 0; k < 3; ++k)
        a.loss.cc[k] = (loss_rho >= 1 && loss_rho <= 8 && k < ncc[loss_rho]) ? loss_cc[k] : 0.0;
    a.irls_maxiter = ci[14];
    a.irls_xtol = cd[10];
    a.par = h->d_par;
    a.scal = h->d_scal;
    a.ints = h->d_ints;
    a.prof = nullptr;
    unsigned long long *d_prof = nullptr;
    if (getenv("GSLNLS_BATCH_PROF")) // developer diagnostic: per-phase shader cycles of every data set
    {
        if (hipMalloc(&d_prof, sizeof(unsigned long long) * 8 * (size_t)h->B) == hipSuccess)
        {
            (void)hipMemset(d_prof, 0, sizeof(unsigned long long) * 8 * (size_t)h->B);
            a.prof = d_prof;
        }
    }
    if (!h->d_passes)
        GSLNLS_HIP_OK(hipMalloc(&h->d_passes, 2 * sizeof(unsigned long long)));
    GSLNLS_HIP_OK(hipMemsetAsync(h->d_passes, 0, 2 * sizeof(unsigned long long), h->st));
    a.pass_total = h->d_passes;
    const int jacmode = jac ? JAC_ANALYTIC : (ci[5] ? JAC_CENTER : JAC_FORWARD);
    hipEventRecord(h->e0, h->st);
    switch (jacmode)
    {
    case JAC_ANALYTIC:
        hipLaunchKernelGGL((irls_batch_kernel<M, JAC_ANALYTIC, T>), dim3(hi - lo), dim3(T), 0, h->st, a);
        break;
    case JAC_FORWARD:
        hipLaunchKernelGGL((irls_batch_kernel<M, JAC_FORWARD, T>), dim3(hi - lo), dim3(T), 0, h->st, a);
        break;
    default:
        hipLaunchKernelGGL((irls_batch_kernel<M, JAC_CENTER, T>), dim3(hi - lo), dim3(T), 0, h->st, a);
        break;
    }
    hipEventRecord(h->e1, h->st);
    GSLNLS_HIP_OK(hipMemcpyAsync(h->h_passes, h->d_passes, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->st));
    GSLNLS_HIP_OK(hipStreamSynchronize(h->st));
    if (kernel_ms)
        hipEventElapsedTime(kernel_ms, h->e0, h->e1);
    if (d_prof)
    {
        std::vector<unsigned long long> hp(8 * (size_t)h->B);
        (void)hipMemcpy(hp.data(), d_prof, sizeof(unsigned long long) * hp.size(), hipMemcpyDeviceToHost);
        (void)hipFree(d_prof);
        double tot[8] = {0};
        for (int d = lo; d < hi; ++d)
            for (int k = 0; k < 8; ++k)
                tot[k] += (double)hp[(size_t)d * 8 + k];
        const double nd = hi - lo;
        fprintf(stderr, "[batch prof] per data set (kcycles): rows %.0f  reduce %.0f  advance %.0f  reweight %.0f  total %.0f "
                        "| passes %.1f | of reweight: radix selects %.0f, bracketed selects %.0f\n",
                tot[0] / nd / 1e3, tot[1] / nd / 1e3, tot[2] / nd / 1e3, tot[3] / nd / 1e3, tot[4] / nd / 1e3, tot[5] / nd,
                tot[6] / nd / 1e3, tot[7] / nd / 1e3);
    }
    return GSLNLS_SUCCESS;
}

extern "C" {

gslnls_dense *gslnls_dense_create(const gslnls_model *fn, const double *y, int n, const double *swts, int *err)
{
    int e = 0;
    DenseBase *b = make_dense(fn, y, n, swts, &e);
    if (err)
        *err = e;
    if (!b)
        return nullptr;
    gslnls_dense *h = new gslnls_dense;
    h->impl = b;
    return h;
}

void gslnls_dense_destroy(gslnls_dense *h)
{
    if (h)
    {
        release_dense(h->impl);
        delete h;
    }
}

int gslnls_dense_solve(gslnls_dense *h, int jac, int fvv, const double *start, const double *lupars,
                       const int *control_int, const double *control_dbl, int chunk, gslnls_result *out)
{
    if (!h || !h->impl)
        return GSLNLS_EINVAL;
    return h->impl->solve(jac, fvv, start, lupars, control_int, control_dbl, chunk, out);
}

float gslnls_dense_time_pass(gslnls_dense *h, int jac, const double *theta, int reps)
{
    if (!h || !h->impl)
        return -1.f;
    return h->impl->time_pass(jac, theta, reps);
}

int gslnls_dense_loop_event_stats(gslnls_dense *h, double *ms_total, long long *launches_total, int reset)
{
    if (!h || !h->impl)
        return GSLNLS_EINVAL;
    return h->impl->loop_event_stats(ms_total, launches_total, reset);
}

#ifdef GSLNLS_STAMPS
int gslnls_debug_stamps(gslnls_dense *h, int jac, const double *theta, int warm, unsigned long long *out, int *nrows)
{
    return h->impl->debug_stamps(jac, theta, warm, out, nrows);
}
int gslnls_debug_adv_stamps(unsigned long long *out8)
{
    return hipMemcpyFromSymbol(out8, HIP_SYMBOL(gslnls::g_adv_stamps), sizeof(unsigned long long) * 8) == hipSuccess ? 0 : -1;
}
#endif

gslnls_large *gslnls_large_create(const gslnls_model *fn, const double *y, int n, const double *weights, int *err)
{
    int e = GSLNLS_SUCCESS;
    gslnls_large *h = new gslnls_large;
    h->n = n;
    h->p = fn->p;
    std::vector<double> sw;
    const double *swp = nullptr;
    if (weights && !fn->x_on_device)
    {
        sw.resize(n);
        for (int i = 0; i < n; ++i)
            sw[i] = sqrt(weights[i]); // gsl_multilarge_nlinear_winit
        swp = sw.data();
    }
    else if (weights)
        swp = weights; // device data: caller passes sqrt(weights) already on device
    if (fn->id == GSLNLS_MODEL_GLMEXP)
    {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
            e = GSLNLS_E_NODEVICE;
        else if (fn->nx != fn->p)
            e = GSLNLS_EINVAL;
        else
        {
#define GSLNLS_GLM(PP)                                                  \
    {                                                                   \
        auto *o = new GlmLargeOps<PP>();                                \
        e = o->init(fn->x, y, swp, n, fn->x_on_device != 0);            \
        h->ops = o;                                                     \
    }
            switch (fn->p)
            {
            case 16:
                GSLNLS_GLM(16);
                break;
            case 32:
                GSLNLS_GLM(32);
                break;
            case 64:
                GSLNLS_GLM(64);
                break;
            default:
                e = GSLNLS_E_UNSUPPORTED;
            }
#undef GSLNLS_GLM
        }
    }
    else
    {
        h->dense = make_dense(fn, y, n, swp, &e);
        if (h->dense)
        {
            h->ops = h->dense->make_large_ops();
            if (!h->ops && e == GSLNLS_SUCCESS)
                e = GSLNLS_E_UNSUPPORTED; // (formulas with p > 9: the wide path has no matrix-free operators)
        }
    }
    if (err)
        *err = e;
    if (e != GSLNLS_SUCCESS || !h->ops)
    {
        delete h->ops;
        release_dense(h->dense);
        delete h;
        return nullptr;
    }
    return h;
}

// one destroyed sparse-callback problem is parked for the next create (sparse_large.hpp: allocation vs binding)
static SparseCbOps *g_parked_sparse = nullptr;

void gslnls_large_destroy(gslnls_large *h)
{
    if (h)
    {
        if (auto *sp = dynamic_cast<SparseCbOps *>(h->ops))
        {
            // (large ones give their memory back at once: 64 MB of n- and p-sized buffers is the limit)
            if ((size_t)sp->cap_n * 40 + (size_t)sp->cap_p * 100 + (size_t)sp->cap_nnz * 40 <= (size_t)64 << 20)
            {
                delete g_parked_sparse;
                g_parked_sparse = sp;
                h->ops = nullptr;
            }
        }
        delete h->ops;
        release_dense(h->dense);
        delete h;
    }
}

int gslnls_large_solve(gslnls_large *h, const double *start, const int *control_int, const double *control_dbl,
                       gslnls_large_result *out)
{
    if (!h || !h->ops)
        return GSLNLS_EINVAL;
    const int p = h->p, n = h->n;
    LargeOps &ops = *h->ops;
    if (h->dense && h->dense->prepare_device())
        return GSLNLS_E_NODEVICE;
    ops.nevalf = ops.nevaldfu = ops.nevaldf2 = 0;
    ops.npass = 0;
    LargeResult R;
    trace_begin(control_int[1] != 0);
    struct TraceOff
    {
        ~TraceOff() { g_trace_on = false; }
    } trace_off;
    const bool trace = control_int[1] != 0 && out->ssrtrace && out->partrace;
    if (trace)
    {
        const int mi = control_int[0];
        for (int i = 0; i <= mi; ++i)
            out->ssrtrace[i] = NAN;
        for (size_t i = 0; i < (size_t)(mi + 1) * p; ++i)
            out->partrace[i] = NAN;
    }
    const int rc = large_solve(ops, start, control_int, control_dbl, R, trace ? out->ssrtrace : nullptr,
                               trace ? out->partrace : nullptr);
    if (rc)
        return rc;
    const bool ok = (R.status == ST_SUCCESS || R.status == ST_EMAXITER);
    for (int k = 0; k < p; ++k)
        if (out->par)
            out->par[k] = ok ? R.x[k] : start[k];
    if (out->resid)
    {
        if (ok)
            ops.residual(R.x.data(), out->resid);
        else
            for (int i = 0; i < n; ++i)
                out->resid[i] = NAN;
    }
    if (out->covar)
    {
        bool good = ok;
        if (good)
        {
            // gsl_multilarge_nlinear_covar (src/nls_large.c:255): (J^T J)^-1 at the final point.  From p = 65 on the inverse
            // comes from the device (round 5: the host's factor-and-invert was 15 ms of every call at p = 500) -- from
            // where J^T J sits when the operator keeps it there, else uploaded; the host routine when the natural-order
            // factorisation refuses the matrix
            std::vector<double> A((size_t)p * p);
            bool done = false;
            const bool try_dev = p >= 65 && p <= 4096 && !getenv("GSLNLS_BD_HOST_EPILOGUE");
            if (try_dev && ops.can_keep_jtj_on_device())
            {
                ops.jtj_device_only = true;
                const int rj = ops.full_jtj(R.x.data(), A.data());
                ops.jtj_device_only = false;
                good = rj == 0;
                if (good)
                {
                    if (const double *jd = ops.jtj_device())
                        done = bd_spd_inverse(p, jd, nullptr, out->covar) == 0;
                    if (!done)
                        good = ops.jtj_download(A.data()) == GSLNLS_SUCCESS;
                }
            }
            else
            {
                good = ops.full_jtj(R.x.data(), A.data()) == 0;
                if (good && try_dev)
                    done = bd_spd_inverse(p, nullptr, A.data(), out->covar) == 0;
            }
            if (good && !done)
            {
                good = lg_chol(p, A);
                if (good)
                {
                    lg_chol_invert(p, A);
                    for (int i = 0; i < p; ++i)
                        for (int k = 0; k < p; ++k)
                            out->covar[i + (size_t)p * k] = A[(size_t)i * p + k];
                }
            }
        }
        if (!good)
            for (size_t i = 0; i < (size_t)p * p; ++i)
                out->covar[i] = NAN;
    }
    out->niter = R.niter;
    out->conv = R.status;
    out->info = R.info;
    out->ssr = R.chisq1;
    out->ssrtol = R.chisq0 - R.chisq1;
    out->chisq_init = R.chisq_init;
    out->neval[0] = (int)ops.nevalf;
    out->neval[1] = (int)ops.nevaldfu;
    out->neval[2] = (int)ops.nevaldf2;
    out->neval[3] = 0;
    out->n_passes = (int)ops.npass;
    out->last_pass_ms = ops.pass_ms;
    // the summary block of a verbose call (src/nls_large.c:259-273)
    trace_printf("*******************\nsummary from method 'multilarge/%s'\n", gslnls_algorithm_name(control_int[2]));
    trace_printf("number of iterations: %d\n", out->niter);
    // (gsl_strerror of the convergence test's info = 1 / 2, i.e. of GSL_EDOM / GSL_ERANGE: what the reference prints)
    trace_printf("reason for stopping: %s\n", gslnls_strerror(out->info));
    trace_printf("initial ssr = %g\n", out->chisq_init);
    trace_printf("final ssr = %g\n", out->ssr);
    trace_printf("ssr/dof = %g\n", out->ssr / (n - p));
    trace_printf("ssr achieved tolerance = %g\n", out->ssrtol);
    trace_printf("function evaluations: %d\n", out->neval[0]);
    trace_printf("jacobian-vector product evaluations: %d\n", out->neval[1]);
    trace_printf("jacobian-jacobian product evaluations: %d\n", out->neval[2]);
    trace_printf("fvv evaluations: %d\n", out->neval[3]);
    trace_printf("status = %s\n*******************\n", gslnls_strerror(out->conv));
    return R.status;
}

gslnls_large *gslnls_large_create_sparse(int n, int p, const double *y, const double *weights, gslnls_large_f_cb f,
                                         gslnls_large_jac_cb jac, void *user, int *err)
{
    int e = GSLNLS_SUCCESS;
    gslnls_large *h = nullptr;
    if (n < 1 || p < 1 || !y || !f || !jac)
        e = GSLNLS_EINVAL;
    else
    {
        std::vector<double> sw;
        if (weights)
        {
            sw.resize(n);
            for (int i = 0; i < n; ++i)
                sw[i] = sqrt(weights[i]);
        }
        SparseCbOps *ops = nullptr;
        if (g_parked_sparse && g_parked_sparse->fits(n, p))
        {
            ops = g_parked_sparse;
            g_parked_sparse = nullptr;
            e = ops->bind(n, p, y, weights ? sw.data() : nullptr, f, jac, user);
        }
        else
        {
            ops = new SparseCbOps();
            e = ops->init(n, p, y, weights ? sw.data() : nullptr, f, jac, user);
        }
        if (e == GSLNLS_SUCCESS)
        {
            h = new gslnls_large;
            h->ops = ops;
            h->n = n;
            h->p = p;
        }
        else
            delete ops;
    }
    if (err)
        *err = e;
    return h;
}

int gslnls_nls_large(const gslnls_model *fn, const double *y, int n, const double *start, const double *weights,
                     const int *control_int, const double *control_dbl, gslnls_large_result *out)
{
    int err = 0;
    gslnls_large *h = gslnls_large_create(fn, y, n, weights, &err);
    if (!h)
        return err;
    const int rc = gslnls_large_solve(h, start, control_int, control_dbl, out);
    gslnls_large_destroy(h);
    return rc;
}

float gslnls_large_time_pass(gslnls_large *h, int mode, const double *x, const double *u, int reps)
{
    if (!h || !h->ops || reps < 1)
        return -1.f;
    if (h->dense && h->dense->prepare_device())
        return -1.f;
    std::vector<double> g(h->p), d(h->p);
    double ssr, bad, nw2;
    if (h->ops->eval(x, &ssr, g.data(), d.data(), nullptr, &bad))
        return -1.f;
    if (h->ops->lazy_jac && h->ops->eval_jac(g.data(), d.data(), nullptr))
        return -1.f;
    h->ops->accept();
    double tot = 0.0;
    for (int r = 0; r < reps; ++r)
    {
        if (mode == 0)
        {
            if (h->ops->eval(x, &ssr, g.data(), d.data(), nullptr, &bad))
                return -1.f;
        }
        else if (mode == 2)
        {
            std::vector<double> J((size_t)h->p * h->p);
            const double t0 = now_s();
            if (h->ops->full_jtj(x, J.data()))
                return -1.f;
            h->ops->pass_ms = (float)(1e3 * (now_s() - t0)); // kernel + reduce + 32 KB readback
        }
        else if (h->ops->jtjv(x, u, &nw2, g.data()))
            return -1.f;
        tot += h->ops->pass_ms;
    }
    return (float)(tot / reps);
}

gslnls_batch *gslnls_batch_create(int model_id, int p, int nx, const double *x, const double *y, const double *swts,
                                  int n, int B, int *err)
{
    int ndev = 0, e = GSLNLS_SUCCESS;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        e = GSLNLS_E_NODEVICE;
    gslnls_batch *h = nullptr;
    if (e == GSLNLS_SUCCESS && (B < 0 || n < 1 || p < 1 || (B > 0 && (!x || !y))))
        e = GSLNLS_EINVAL;
    if (e == GSLNLS_SUCCESS && B == 0)
    {
        // an empty block (B_total = 9 over 8 ranks leaves ranks 5..7 without data sets): the handle only takes part
        // in the final all-gather of gslnls_batch_irls_gather
        h = new gslnls_batch;
        h->model_id = model_id;
        h->p = p;
        h->nx = nx;
        h->n = n;
        h->B = 0;
        if (hipStreamCreateWithFlags(&h->st, hipStreamNonBlocking) != hipSuccess)
            e = GSLNLS_E_NODEVICE;
    }
    else if (e == GSLNLS_SUCCESS)
    {
        h = new gslnls_batch;
        h->model_id = model_id;
        h->p = p;
        h->nx = nx;
        h->n = n;
        h->B = B;
        const size_t nb = sizeof(double) * (size_t)n * B;
        bool ok = hipStreamCreateWithFlags(&h->st, hipStreamNonBlocking) == hipSuccess &&
                  hipEventCreate(&h->e0) == hipSuccess && hipEventCreate(&h->e1) == hipSuccess &&
                  hipMalloc(&h->d_x, nb * nx) == hipSuccess && hipMalloc(&h->d_y, nb) == hipSuccess &&
                  hipMalloc(&h->d_sw, nb) == hipSuccess && hipMalloc(&h->d_keys, nb) == hipSuccess &&
                  hipMalloc(&h->d_par, sizeof(double) * (size_t)B * p) == hipSuccess &&
                  hipMalloc(&h->d_scal, sizeof(double) * (size_t)B * 4) == hipSuccess &&
                  hipMalloc(&h->d_ints, sizeof(int) * (size_t)B * 4) == hipSuccess &&
                  hipMemcpy(h->d_x, x, nb * nx, hipMemcpyHostToDevice) == hipSuccess &&
                  hipMemcpy(h->d_y, y, nb, hipMemcpyHostToDevice) == hipSuccess;
        if (ok && swts)
            ok = hipMalloc(&h->d_usw, nb) == hipSuccess && hipMemcpy(h->d_usw, swts, nb, hipMemcpyHostToDevice) == hipSuccess;
        if (!ok)
            e = GSLNLS_E_NODEVICE;
    }
    if (err)
        *err = e;
    if (e != GSLNLS_SUCCESS)
    {
        gslnls_batch_destroy(h);
        return nullptr;
    }
    return h;
}

void gslnls_batch_destroy(gslnls_batch *h)
{
    if (!h)
        return;
    hipFree(h->d_x);
    hipFree(h->d_y);
    hipFree(h->d_usw);
    hipFree(h->d_sw);
    hipFree(h->d_keys);
    hipFree(h->d_par);
    hipFree(h->d_scal);
    hipFree(h->d_ints);
    if (h->e0)
        hipEventDestroy(h->e0);
    if (h->e1)
        hipEventDestroy(h->e1);
    if (h->st)
        hipStreamDestroy(h->st);
    delete h;
}

int gslnls_batch_irls(gslnls_batch *h, int lo, int hi, int jac, int fvv, const double *start, const double *lupars,
                      const int *control_int, const double *control_dbl, int loss_rho, const double *loss_cc,
                      double *par, double *scal, int *ints, float *kernel_ms)
{
    if (!h || lo < 0 || hi > h->B || lo >= hi || loss_rho < 1 || loss_rho > 8 || control_int[2] > 1)
        return GSLNLS_EINVAL;
    int rc;
    switch (h->model_id)
    {
    case GSLNLS_MODEL_EXPDECAY:
        rc = batch_irls_run<ModelExpDecay>(h, lo, hi, jac, fvv, start, lupars, control_int, control_dbl, loss_rho, loss_cc, kernel_ms);
        break;
    case GSLNLS_MODEL_MISRA1A:
        rc = batch_irls_run<ModelMisra1a>(h, lo, hi, jac, fvv, start, lupars, control_int, control_dbl, loss_rho, loss_cc, kernel_ms);
        break;
    case GSLNLS_MODEL_GAUSSPK:
        rc = batch_irls_run<ModelGaussPeak>(h, lo, hi, jac, fvv, start, lupars, control_int, control_dbl, loss_rho, loss_cc, kernel_ms);
        break;
    case GSLNLS_MODEL_GAUSS1:
        rc = batch_irls_run<ModelGauss1>(h, lo, hi, jac, fvv, start, lupars, control_int, control_dbl, loss_rho, loss_cc, kernel_ms);
        break;
    default:
        return GSLNLS_E_UNSUPPORTED;
    }
    if (rc)
        return rc;
    const int cnt = hi - lo, p = h->p;
    if (par)
        GSLNLS_HIP_OK(hipMemcpy(par, h->d_par + (size_t)lo * p, sizeof(double) * (size_t)cnt * p, hipMemcpyDeviceToHost));
    if (scal)
        GSLNLS_HIP_OK(hipMemcpy(scal, h->d_scal + (size_t)lo * 4, sizeof(double) * (size_t)cnt * 4, hipMemcpyDeviceToHost));
    if (ints)
        GSLNLS_HIP_OK(hipMemcpy(ints, h->d_ints + (size_t)lo * 4, sizeof(int) * (size_t)cnt * 4, hipMemcpyDeviceToHost));
    return GSLNLS_SUCCESS;
}

// pack the per-data-set outputs of a batched robust fit into one record of p + 8 doubles: par[p], scal[4], ints[4]
__global__ void batch_pack_kernel(const double *par, const double *scal, const int *ints, int p, int cnt, double *out)
{
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= cnt)
        return;
    double *o = out + (size_t)d * (p + 8);
    for (int k = 0; k < p; ++k)
        o[k] = par[(size_t)d * p + k];
    for (int k = 0; k < 4; ++k)
    {
        o[p + k] = scal[(size_t)d * 4 + k];
        o[p + 4 + k] = (double)ints[(size_t)d * 4 + k];
    }
}

int gslnls_batch_last_passes(gslnls_batch *h, long long *lm_passes, long long *reweightings)
{
    if (!h)
        return GSLNLS_EINVAL;
    if (lm_passes)
        *lm_passes = (long long)h->h_passes[0];
    if (reweightings)
        *reweightings = (long long)h->h_passes[1];
    return GSLNLS_SUCCESS;
}

int gslnls_batch_irls_gather(gslnls_batch *h, int B_total, int jac, int fvv, const double *start, const double *lupars,
                             const int *control_int, const double *control_dbl, int loss_rho, const double *loss_cc,
                             double *par, double *scal, int *ints, float *kernel_ms)
{
    if (!h || B_total < 1)
        return GSLNLS_EINVAL;
    const int world = g_comm.world, rank = g_comm.rank;
    const int per = (B_total + world - 1) / world;
    const int lo = std::min(B_total, rank * per), hi = std::min(B_total, lo + per);
    const int p = h->p, K = p + 8;
    if (world == 1)
    {
        if (h->B != B_total)
            return GSLNLS_EINVAL;
        int rc1 = gslnls_batch_irls(h, 0, h->B, jac, fvv, start, lupars, control_int, control_dbl, loss_rho, loss_cc, nullptr,
                                    nullptr, nullptr, kernel_ms);
        if (rc1)
            return rc1;
        GSLNLS_HIP_OK(hipMemcpy(par, h->d_par, sizeof(double) * (size_t)h->B * p, hipMemcpyDeviceToHost));
        GSLNLS_HIP_OK(hipMemcpy(scal, h->d_scal, sizeof(double) * (size_t)h->B * 4, hipMemcpyDeviceToHost));
        GSLNLS_HIP_OK(hipMemcpy(ints, h->d_ints, sizeof(int) * (size_t)h->B * 4, hipMemcpyDeviceToHost));
        return GSLNLS_SUCCESS;
    }
    // one all-gather of per x (p + 8) doubles per rank: the final exchange of SURVEY.md 8(e), row "batched IRLS".
    // What the ranks can only disagree on (this rank's handle, its fit, its HIP calls) is folded into `rc` and carried
    // through the collective in the status slot of the shard's first record (MS_SHARD_FAILED): from here to the
    // all-gather there is no return that the other ranks would not take as well.
    const size_t shard_doubles = (size_t)per * K;
    const bool lib = g_comm.rccl != nullptr;
    if (lib)
    {
        const int e = g_rccl.ensure_together(shard_doubles, h->st); // all ranks or none
        if (e)
            return e;
    }
    else if (!g_comm.allgather || !g_comm.shard_buf || !g_comm.all_buf ||
             (long long)per * K * world > g_comm.cap_points * (long long)gslnls_mstart_record_size(p))
        return GSLNLS_EINVAL; // the registration (gslnls_set_comm) is the same on every rank
    int rc = 0;
    if (h->B != hi - lo)
        rc = GSLNLS_EINVAL; // the handle must hold exactly this rank's block (an empty one: a handle created with B = 0)
    else if (hi > lo)
        rc = gslnls_batch_irls(h, 0, h->B, jac, fvv, start, lupars, control_int, control_dbl, loss_rho, loss_cc, nullptr,
                               nullptr, nullptr, kernel_ms);
    else if (kernel_ms)
        *kernel_ms = 0.f;
    auto soft = [&rc](hipError_t e) {
        if (e != hipSuccess && rc == 0)
        {
            fprintf(stderr, "gslnls: HIP error %s in gslnls_batch_irls_gather (carried into the collective)\n", hipGetErrorString(e));
            (void)hipGetLastError();
            rc = GSLNLS_E_NODEVICE;
        }
    };
    std::vector<double> rec((size_t)world * per * K, 0.0);
    const bool dev_shard = lib || g_comm.buffers_on_device;
    double *d_shard = lib ? g_rccl.shard : (g_comm.buffers_on_device ? g_comm.shard_buf : nullptr);
    if (dev_shard)
    {
        soft(hipMemsetAsync(d_shard, 0, sizeof(double) * shard_doubles, h->st));
        if (hi > lo && rc == 0)
        {
            hipLaunchKernelGGL(batch_pack_kernel, dim3((h->B + 255) / 256), dim3(256), 0, h->st, h->d_par, h->d_scal, h->d_ints, p,
                               h->B, d_shard);
            soft(hipGetLastError());
        }
        if (rc)
        {
            static const double bad = MS_SHARD_FAILED;
            (void)hipMemcpyAsync(d_shard + p + 4, &bad, sizeof(double), hipMemcpyHostToDevice, h->st);
        }
    }
    else
    {
        // host buffers (gloo in the CPU-side tests): pack the shard on the host, nothing to allocate
        double *sh = g_comm.shard_buf;
        std::fill(sh, sh + shard_doubles, 0.0);
        if (hi > lo && rc == 0)
        {
            std::vector<double> hp((size_t)h->B * p), hs((size_t)h->B * 4);
            std::vector<int> hi4((size_t)h->B * 4);
            soft(hipMemcpy(hp.data(), h->d_par, sizeof(double) * hp.size(), hipMemcpyDeviceToHost));
            soft(hipMemcpy(hs.data(), h->d_scal, sizeof(double) * hs.size(), hipMemcpyDeviceToHost));
            soft(hipMemcpy(hi4.data(), h->d_ints, sizeof(int) * hi4.size(), hipMemcpyDeviceToHost));
            for (int d = 0; d < h->B && rc == 0; ++d)
            {
                double *o = sh + (size_t)d * K;
                for (int k = 0; k < p; ++k)
                    o[k] = hp[(size_t)d * p + k];
                for (int k = 0; k < 4; ++k)
                {
                    o[p + k] = hs[(size_t)d * 4 + k];
                    o[p + 4 + k] = (double)hi4[(size_t)d * 4 + k];
                }
            }
        }
        if (rc)
            sh[p + 4] = MS_SHARD_FAILED;
    }
    if (lib)
    {
        int e = g_rccl.allgather(shard_doubles, h->st);
        if (e)
            return e;
        GSLNLS_HIP_OK(hipMemcpyAsync(rec.data(), g_rccl.all, sizeof(double) * rec.size(), hipMemcpyDeviceToHost, h->st));
        GSLNLS_HIP_OK(hipStreamSynchronize(h->st));
    }
    else
    {
        if (dev_shard)
            soft(hipStreamSynchronize(h->st));
        // the callback form counts in records of the multi-start size; hand it the doubles as (per * K) x 1
        const int e = g_comm.allgather(g_comm.ctx, (int)shard_doubles, 1);
        if (e)
            return e;
        if (g_comm.buffers_on_device)
            GSLNLS_HIP_OK(hipMemcpy(rec.data(), g_comm.all_buf, sizeof(double) * rec.size(), hipMemcpyDeviceToHost));
        else
            memcpy(rec.data(), g_comm.all_buf, sizeof(double) * rec.size());
    }
    // (every shard was zero-filled before it was packed, so the status slot of a rank with an empty block is valid too:
    // such a rank can still fail -- a handle of the wrong size, a HIP error)
    for (int r = 0; r < world; ++r)
        if (rec[(size_t)r * per * K + p + 4] == MS_SHARD_FAILED)
            return rc ? rc : GSLNLS_FAILURE;
    if (rc)
        return rc;
    for (int d = 0; d < B_total; ++d)
    {
        const double *o = rec.data() + (size_t)d * K; // block r starts at r * per: contiguous in d
        for (int k = 0; k < p; ++k)
            par[(size_t)d * p + k] = o[k];
        for (int k = 0; k < 4; ++k)
        {
            scal[(size_t)d * 4 + k] = o[p + k];
            ints[(size_t)d * 4 + k] = (int)o[p + 4 + k];
        }
    }
    return GSLNLS_SUCCESS;
}

int gslnls_lower_formula(const char *rhs, int p, const char *const *parnames, int *par_order, char *varnames_out,
                          int varnames_cap)
{
    if (!rhs || p < 1 || !parnames || !par_order)
        return 0;
    std::vector<std::string> vars;
    const int id = lower_formula(rhs, p, parnames, par_order, vars);
    if (id > 0 && varnames_out && varnames_cap > 0)
    {
        std::string joined;
        for (size_t i = 0; i < vars.size(); ++i)
            joined += (i ? "," : "") + vars[i];
        strncpy(varnames_out, joined.c_str(), (size_t)varnames_cap - 1);
        varnames_out[varnames_cap - 1] = 0;
    }
    return id;
}

int gslnls_set_comm(int rank, int world, gslnls_allgather_fn fn, void *ctx, double *shard_buf, double *all_buf,
                    long long cap_points, int buffers_on_device)
{
    if (world < 1 || rank < 0 || rank >= world)
        return GSLNLS_EINVAL;
    g_comm = MsComm(); // the callback form replaces an in-library communicator, if one was bound
    g_comm.rank = rank;
    g_comm.world = world;
    g_comm.allgather = fn;
    g_comm.ctx = ctx;
    g_comm.shard_buf = shard_buf;
    g_comm.all_buf = all_buf;
    g_comm.cap_points = cap_points;
    g_comm.buffers_on_device = buffers_on_device;
    return GSLNLS_SUCCESS;
}

int gslnls_comm_get_unique_id(char *id128)
{
    if (!id128)
        return GSLNLS_EINVAL;
    return g_rccl.get_unique_id(id128);
}

static int comm_bind(int rc)
{
    if (rc == GSLNLS_SUCCESS)
    {
        g_comm = MsComm();
        g_comm.rank = g_rccl.rank;
        g_comm.world = g_rccl.world;
        g_comm.rccl = &g_rccl;
        g_comm.rccl_run = ms_rccl_run_batch;
        g_comm.buffers_on_device = 1;
        const char *f = getenv("GSLNLS_COMM_FORCE_COLLECTIVE");
        g_comm.force_collective = (f && f[0] == '1') ? 1 : 0;
    }
    return rc;
}

int gslnls_comm_init_rank(const char *id128, int rank, int world)
{
    if (!id128)
        return GSLNLS_EINVAL;
    return comm_bind(g_rccl.init_rank(id128, rank, world));
}

int gslnls_comm_init_file(const char *path, int rank, int world, int timeout_s)
{
    if (!path)
        return GSLNLS_EINVAL;
    return comm_bind(g_rccl.init_file(path, rank, world, timeout_s > 0 ? timeout_s : 60));
}

void gslnls_comm_destroy(void)
{
    g_rccl.destroy();
    g_comm = MsComm();
}

long long gslnls_comm_allgather_count(void) { return g_rccl.n_allgathers; }

void gslnls_comm_set_timing(int on) { g_rccl.set_timing(on); }

double gslnls_comm_allgather_ms(long long *timed)
{
    g_rccl.collect_timing();
    if (timed)
        *timed = g_rccl.allgather_timed;
    return g_rccl.allgather_ms_total;
}

const char *gslnls_comm_last_error(void) { return g_rccl.api.err; }

int gslnls_dense_mstart(gslnls_dense *h, int jac, int fvv, const double *start2p, const double *lupars,
                        const int *control_int, const double *control_dbl, const int *has_start, gslnls_result *out)
{
    if (!h || !h->impl)
        return GSLNLS_EINVAL;
    return h->impl->mstart(jac, fvv, start2p, lupars, control_int, control_dbl, has_start, g_comm, 0, nullptr, out);
}

int gslnls_mstart_batch(gslnls_dense *h, int jac, const double *ranges, const double *kd, long long first_draw,
                        int count, int lo, int hi, int maxiter, double dtol, const int *control_int,
                        const double *control_dbl, const double *lupars, double *records, int records_on_device,
                        float *kernel_ms)
{
    if (!h || !h->impl || count < 1)
        return GSLNLS_EINVAL;
    if (lo < 0)
    {
        if (records_on_device)
            return GSLNLS_EINVAL;
        h->impl->batch_comm = &g_comm;
    }
    else if (hi > count || lo > hi)
        return GSLNLS_EINVAL;
    return h->impl->mstart_batch(jac, ranges, kd, first_draw, count, lo, hi, maxiter, dtol, control_int, control_dbl,
                                 lupars, records, records_on_device, kernel_ms);
}

int gslnls_mstart_record_size(int p) { return 3 * p + 8; }

int gslnls_dense_set_swts(gslnls_dense *h, const double *swts)
{
    if (!h || !h->impl)
        return GSLNLS_EINVAL;
    return h->impl->set_swts(swts);
}

void gslnls_set_interrupt_hook(int (*check)(void))
{
    g_interrupt_hook = check;
    ms_interrupt_hook = check;
}

int gslnls_dense_diagnostics(gslnls_dense *h, int jac, const double *par, const int *control_int,
                             const double *control_dbl, double *hat, double *cooks)
{
    if (!h || !h->impl || !par || !control_int || !control_dbl)
        return GSLNLS_EINVAL;
    return h->impl->diagnostics(jac, par, control_int, control_dbl, hat, cooks);
}

// ---- trace = TRUE: the text of a verbose call (trace_log.hpp) ------------------------------------------------------------
// iteration lines of the final solve (callback, src/nls.c:980-995; none under a robust loss: callback_irls only records,
// src/nls_irls.c:364-374) + the summary block (src/nls.c:610-630), appended to `log`
static void format_nls_trace(const gslnls_result *res, int n, int p, const int *ci, int loss_rho, std::string &log)
{
    char buf[512];
    auto put = [&](const char *fmt, auto... a) {
        snprintf(buf, sizeof(buf), fmt, a...);
        log += buf;
    };
    const int maxiter = ci[0];
    if (!loss_rho && res->ssrtrace && res->partrace)
        for (int it = 1; it <= res->niter && it <= maxiter; ++it)
        {
            put("iter %3d: ssr = %g, par = (", it, res->ssrtrace[it]);
            for (int k = 0; k < p; ++k)
                put((k < p - 1) ? "%g, " : "%g)\n", res->partrace[it + (size_t)(maxiter + 1) * trace_index(k, p)]);
        }
    put("*******************\nsummary from method 'multifit/%s'\n", gslnls_algorithm_name(ci[2]));
    if (loss_rho)
    {
        put("IRLS number of iterations: %d\n", res->irls_niter);
        put("IRLS achieved tolerance: %g\n", res->irls_tol);
        put("IRLS convergence status: %s\n", gslnls_strerror(res->irls_status));
    }
    put("number of iterations: %d\n", res->niter);
    put("initial ssr: %g\n", res->chisq_init);
    put("final ssr: %g\n", res->ssr);
    put("ssr/dof: %g\n", res->ssr / (n - p));
    put("ssr achieved tolerance: %g\n", res->ssrtol);
    put("function evaluations: %d\n", res->neval[0]);
    put("jacobian evaluations: %d\n", res->neval[1]);
    put("fvv evaluations: %d\n", res->neval[2]);
    put("status: %s\n*******************\n", gslnls_strerror(res->conv));
}

static size_t copy_text(const std::string &t, char *buf, size_t cap)
{
    if (buf && cap)
    {
        const size_t k = t.size() < cap - 1 ? t.size() : cap - 1;
        memcpy(buf, t.data(), k);
        buf[k] = 0;
    }
    return t.size();
}

size_t gslnls_trace_text(char *buf, size_t cap) { return copy_text(g_trace_log, buf, cap); }

size_t gslnls_format_trace(const gslnls_result *res, int n, int p, const int *control_int, int loss_rho, char *buf, size_t cap)
{
    if (!res || !control_int || p < 1)
        return 0;
    std::string t;
    format_nls_trace(res, n, p, control_int, loss_rho, t);
    return copy_text(t, buf, cap);
}

int gslnls_trace_set_order(const int *par_order, int p)
{
    g_trace_inv.clear();
    if (!par_order || p < 1)
        return GSLNLS_SUCCESS;
    g_trace_inv.assign(p, -1);
    for (int j = 0; j < p; ++j)
        if (par_order[j] >= 0 && par_order[j] < p)
            g_trace_inv[par_order[j]] = j;
    for (int k = 0; k < p; ++k)
        if (g_trace_inv[k] < 0)
        {
            g_trace_inv.clear();
            return GSLNLS_EINVAL; // not a permutation
        }
    return GSLNLS_SUCCESS;
}

// closes the log of a verbose gslnls_nls* call: a fit that produced a result (any GSL status) gets its lines
static void trace_finish_nls(int rc, const gslnls_result *out, int n, int p, const int *ci, int loss_rho)
{
    if (g_trace_on && rc > GSLNLS_E_NODEVICE)
        format_nls_trace(out, n, p, ci, loss_rho, g_trace_log);
    g_trace_on = false;
    g_trace_inv.clear();
}

int gslnls_nls(const gslnls_model *fn, const double *y, int n, int jac, int fvv, const double *start,
               int start_is_matrix, const double *swts, int swts_is_matrix, const double *lupars,
               const int *control_int, const double *control_dbl, const int *has_start, int loss_rho,
               const double *loss_cc, gslnls_result *out)
{
    if (swts && swts_is_matrix)
        return GSLNLS_E_UNSUPPORTED; // GLS: n x n factor, not lowered
    g_call_prof = CallProfile();
    const double t0 = now_s();
    trace_begin(control_int && control_int[1] != 0);
#ifndef GSLNLS_NO_EXPR
    // (test switch GSLNLS_MATRIX_PATH_MIN_P: formulas from that many parameters on take the matrix path -- the wide path's
    // state machine and the matrix path's can then be run on the same problem, tests/test_gpu_wide.py)
    int matrix_min_p = 65;
    if (const char *e = getenv("GSLNLS_MATRIX_PATH_MIN_P"))
        matrix_min_p = atoi(e) > 1 ? atoi(e) : 65;
    if (fn && fn->id == GSLNLS_MODEL_EXPR && fn->p >= matrix_min_p) // (default: beyond WIDE_MAX_P = 64, vm_program.hpp)
    {
        // more than 64 parameters: the Jacobian is a matrix in HBM (csrc/bd_host.hpp)
        if (loss_rho < 0 || loss_rho > 8 || (loss_rho != 0 && !loss_cc) || (start_is_matrix && !has_start))
            return GSLNLS_EINVAL;
        const int rcb = bd_formula_nls(fn, y, n, jac, fvv, start, start_is_matrix, has_start, g_comm, swts, lupars, control_int,
                                       control_dbl, loss_rho, loss_cc, out);
        g_call_prof.total_ms = 1e3 * (now_s() - t0);
        trace_finish_nls(rcb, out, n, fn->p, control_int, loss_rho);
        return rcb;
    }
#endif
    // the result vectors are usually pages the process has never touched (a large Rf_allocVector is a fresh mmap): fault
    // them in on a helper thread while the data uploads and the fit runs, instead of under the device-to-host copy
    OutputPrefault pre;
    if (out)
        pre.start(out->resid, (size_t)n * sizeof(double), out->grad, fn ? (size_t)n * fn->p * sizeof(double) : 0);
    int err = 0;
    DenseBase *b = make_dense(fn, y, n, swts, &err);
    if (!b)
        return err;
    const double t1 = now_s();
    b->prefault = &pre;
    int rc;
    if (start_is_matrix)
        rc = b->mstart(jac, fvv, start, lupars, control_int, control_dbl, has_start, g_comm, loss_rho, loss_cc, out);
    else if (loss_rho != 0)
        rc = b->irls(jac, fvv, start, lupars, control_int, control_dbl, loss_rho, loss_cc, out);
    else
        rc = b->solve(jac, fvv, start, lupars, control_int, control_dbl, 0, out);
    b->prefault = nullptr;
    pre.join();
    const double t2 = now_s();
    release_dense(b);
    const double t3 = now_s();
    g_call_prof.create_ms = 1e3 * (t1 - t0) - g_call_prof.h2d_ms;
    g_call_prof.loop_ms = 1e3 * (t2 - t1) - g_call_prof.finalize_ms - g_call_prof.d2h_ms;
    g_call_prof.destroy_ms = 1e3 * (t3 - t2);
    g_call_prof.total_ms = 1e3 * (t3 - t0);
    trace_finish_nls(rc, out, n, fn->p, control_int, loss_rho);
    return rc;
}

int gslnls_nls_fn(int n, int p, const double *y, gslnls_fn_cb f, gslnls_jac_cb jac, gslnls_fvv_cb fvv, void *user,
                  const double *start, const double *swts, const double *lupars, const int *control_int,
                  const double *control_dbl, gslnls_result *out)
{
    if (!f || !y || !start || !control_int || !control_dbl || !out || n < 1 || p < 1)
        return GSLNLS_EINVAL;
    if (p > 4096)
        return GSLNLS_E_UNSUPPORTED;
    trace_begin(control_int[1] != 0);
    const int rc = bd_callback_nls(n, p, y, f, jac, fvv, user, start, 0, nullptr, g_comm, swts, lupars, control_int, control_dbl, 0,
                                   nullptr, out);
    trace_finish_nls(rc, out, n, p, control_int, 0);
    return rc;
}

int gslnls_nls_fn_loss(int n, int p, const double *y, gslnls_fn_cb f, gslnls_jac_cb jac, gslnls_fvv_cb fvv, void *user,
                       const double *start, const double *swts, const double *lupars, const int *control_int,
                       const double *control_dbl, int loss_rho, const double *loss_cc, gslnls_result *out)
{
    if (!f || !y || !start || !control_int || !control_dbl || !out || n < 1 || p < 1 || loss_rho < 0 || loss_rho > 8 ||
        (loss_rho != 0 && !loss_cc))
        return GSLNLS_EINVAL;
    if (p > 4096)
        return GSLNLS_E_UNSUPPORTED;
    trace_begin(control_int[1] != 0);
    const int rc = bd_callback_nls(n, p, y, f, jac, fvv, user, start, 0, nullptr, g_comm, swts, lupars, control_int, control_dbl,
                                   loss_rho, loss_cc, out);
    trace_finish_nls(rc, out, n, p, control_int, loss_rho);
    return rc;
}

int gslnls_nls_fn_mstart(int n, int p, const double *y, gslnls_fn_cb f, gslnls_jac_cb jac, gslnls_fvv_cb fvv, void *user,
                         const double *start2p, const int *has_start, const double *swts, const double *lupars,
                         const int *control_int, const double *control_dbl, int loss_rho, const double *loss_cc,
                         gslnls_result *out)
{
    if (!f || !y || !start2p || !has_start || !control_int || !control_dbl || !out || n < 1 || p < 1 || loss_rho < 0 ||
        loss_rho > 8 || (loss_rho != 0 && !loss_cc))
        return GSLNLS_EINVAL;
    if (p > 4096)
        return GSLNLS_E_UNSUPPORTED;
    trace_begin(control_int[1] != 0);
    const int rc = bd_callback_nls(n, p, y, f, jac, fvv, user, start2p, 1, has_start, g_comm, swts, lupars, control_int, control_dbl,
                                   loss_rho, loss_cc, out);
    trace_finish_nls(rc, out, n, p, control_int, loss_rho);
    return rc;
}

int gslnls_last_call_profile(double *ms, int cap)
{
    const double v[7] = {g_call_prof.create_ms, g_call_prof.h2d_ms,     g_call_prof.loop_ms, g_call_prof.finalize_ms,
                         g_call_prof.d2h_ms,    g_call_prof.destroy_ms, g_call_prof.total_ms};
    if (!ms)
        return GSLNLS_EINVAL;
    for (int k = 0; k < cap && k < 7; ++k)
        ms[k] = v[k];
    return 7;
}

int gslnls_last_matrix_path_profile(double *ms, int cap)
{
    if (!ms)
        return GSLNLS_EINVAL;
    return bd_last_profile(ms, cap);
}

double gslnls_debug_bd_syrk_ms(int n, int p, int reps) { return bd_syrk_ms(n, p, reps); }

int gslnls_solver_served(const int *control_int, const gslnls_result *res)
{
    if (!control_int || !res)
        return 0;
    if (control_int[4] == 1)
        return 1; // cholesky: exactly what the device path does
    return (res->jtj_cond <= GSLNLS_COND_LIMIT) ? 1 : 0; // qr / svd: NaN (failed fit) compares false
}

const char *gslnls_strerror(int code)
{
    switch (code)
    {
    case GSLNLS_SUCCESS:
        return "success";
    case GSLNLS_FAILURE:
        return "failure";
    case GSLNLS_CONTINUE:
        return "the iteration has not converged yet";
    case 1:
        return "input domain error"; // GSL_EDOM (only ever printed as "reason for stopping" of the large path, info = 1)
    case 2:
        return "output range error"; // GSL_ERANGE (the same, info = 2)
    case GSLNLS_EINVAL:
        return "invalid argument supplied by user";
    case GSLNLS_EBADFUNC:
        return "problem with user-supplied function";
    case GSLNLS_EMAXITER:
        return "exceeded max number of iterations";
    case GSLNLS_ENOPROG:
        return "iteration is not making progress towards solution";
    case GSLNLS_E_NODEVICE:
        return "no HIP device / HIP runtime failure";
    case GSLNLS_E_INTERRUPTED:
        return "interrupted by the caller";
    case GSLNLS_E_UNSUPPORTED:
        return "configuration not lowered to the device";
    default:
        return "unknown error code";
    }
}

const char *gslnls_algorithm_name(int trs)
{
    switch (trs)
    {
    case 1:
        return "levenberg-marquardt+accel";
    case 5:
        return "steihaug-toint";
    default:
        return "levenberg-marquardt";
    }
}

int gslnls_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

void gslnls_trim_cache(void)
{
    delete g_parked_sparse;
    g_parked_sparse = nullptr;
    bd_trim_pool();
    DenseFit<ModelExpDecay>::trim_pool();
    DenseFit<ModelMisra1a>::trim_pool();
    DenseFit<ModelGaussPeak>::trim_pool();
    DenseFit<ModelGauss1>::trim_pool();
#ifndef GSLNLS_NO_EXPR
    trim_dense_expr();
#endif
}

int gslnls_set_device(int ordinal)
{
    return hipSetDevice(ordinal) == hipSuccess ? GSLNLS_SUCCESS : GSLNLS_E_NODEVICE;
}

const char *gslnls_version(void) { return "gslnls-mi355x 0.1 (gfx950)"; }

} // extern "C"

#ifdef GSLNLS_NO_EXPR
// diagnostic single-TU builds (make stamps) carry no expression lowering
extern "C" int gslnls_expr_build(const gslnls_model *, char *, int) { return GSLNLS_E_UNSUPPORTED; }
extern "C" int gslnls_expr_native_state(const gslnls_model *, int) { return -1; }
extern "C" int gslnls_expr_prefetch(const gslnls_model *, int) { return -1; }
extern "C" void gslnls_shutdown(void) {}
#endif
