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
#include "dense_host.hpp"
#include "mstart_host.hpp"
#include "formula.hpp"
#include "irls_host.hpp"
#include "large_host.hpp"

using namespace gslnls;

struct gslnls_dense
{
    DenseBase *impl;
};

// process-wide communicator of the multi-start sharding (one process per GPU)
static MsComm g_comm;

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
        auto *d = new DenseFit<MODEL>();                            \
        rc = d->init(fn, y, n, swts);                               \
        b = d;                                                      \
    }
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
        delete h->impl;
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

#ifdef GSLNLS_STAMPS
int gslnls_debug_stamps(gslnls_dense *h, int jac, const double *theta, int warm, unsigned long long *out, int *nrows)
{
    return h->impl->debug_stamps(jac, theta, warm, out, nrows);
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
            switch (fn->id)
            {
            case GSLNLS_MODEL_EXPDECAY:
                h->ops = make_row_ops<ModelExpDecay>(h->dense);
                break;
            case GSLNLS_MODEL_MISRA1A:
                h->ops = make_row_ops<ModelMisra1a>(h->dense);
                break;
            case GSLNLS_MODEL_GAUSSPK:
                h->ops = make_row_ops<ModelGaussPeak>(h->dense);
                break;
            case GSLNLS_MODEL_GAUSS1:
                h->ops = make_row_ops<ModelGauss1>(h->dense);
                break;
            default:
                e = GSLNLS_E_UNSUPPORTED;
            }
        }
    }
    if (err)
        *err = e;
    if (e != GSLNLS_SUCCESS || !h->ops)
    {
        delete h->ops;
        delete h->dense;
        delete h;
        return nullptr;
    }
    return h;
}

void gslnls_large_destroy(gslnls_large *h)
{
    if (h)
    {
        delete h->ops;
        delete h->dense;
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
    ops.nevalf = ops.nevaldfu = ops.nevaldf2 = 0;
    ops.npass = 0;
    LargeResult R;
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
            // gsl_multilarge_nlinear_covar (src/nls_large.c:255): (J^T J)^-1 at the final point
            std::vector<double> A((size_t)p * p);
            good = ops.full_jtj(R.x.data(), A.data()) == 0 && lg_chol(p, A);
            if (good)
            {
                lg_chol_invert(p, A);
                for (int i = 0; i < p; ++i)
                    for (int k = 0; k < p; ++k)
                        out->covar[i + (size_t)p * k] = A[(size_t)i * p + k];
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
    return R.status;
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
    std::vector<double> g(h->p), d(h->p);
    double ssr, bad, nw2;
    if (h->ops->eval(x, &ssr, g.data(), d.data(), nullptr, &bad))
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
        else if (h->ops->jtjv(x, u, &nw2, g.data()))
            return -1.f;
        tot += h->ops->pass_ms;
    }
    return (float)(tot / reps);
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

int gslnls_dense_mstart(gslnls_dense *h, int jac, int fvv, const double *start2p, const double *lupars,
                        const int *control_int, const double *control_dbl, const int *has_start, gslnls_result *out)
{
    if (!h || !h->impl)
        return GSLNLS_EINVAL;
    return h->impl->mstart(jac, fvv, start2p, lupars, control_int, control_dbl, has_start, g_comm, out);
}

int gslnls_mstart_batch(gslnls_dense *h, int jac, const double *ranges, const double *kd, long long first_draw,
                        int count, int lo, int hi, int maxiter, double dtol, const int *control_int,
                        const double *control_dbl, const double *lupars, double *records, int records_on_device,
                        float *kernel_ms)
{
    if (!h || !h->impl || lo < 0 || hi > count || lo > hi)
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

int gslnls_nls(const gslnls_model *fn, const double *y, int n, int jac, int fvv, const double *start,
               int start_is_matrix, const double *swts, int swts_is_matrix, const double *lupars,
               const int *control_int, const double *control_dbl, const int *has_start, int loss_rho,
               const double *loss_cc, gslnls_result *out)
{
    if (swts && swts_is_matrix)
        return GSLNLS_E_UNSUPPORTED; // GLS: n x n factor, not lowered
    if (loss_rho != 0 && start_is_matrix)
        return GSLNLS_E_UNSUPPORTED; // robust multi-start second pass (src/nls.c:401-509): next round
    int err = 0;
    DenseBase *b = make_dense(fn, y, n, swts, &err);
    if (!b)
        return err;
    int rc;
    if (start_is_matrix)
        rc = b->mstart(jac, fvv, start, lupars, control_int, control_dbl, has_start, g_comm, out);
    else if (loss_rho != 0)
        rc = b->irls(jac, fvv, start, lupars, control_int, control_dbl, loss_rho, loss_cc, out);
    else
        rc = b->solve(jac, fvv, start, lupars, control_int, control_dbl, 0, out);
    delete b;
    return rc;
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

int gslnls_set_device(int ordinal)
{
    return hipSetDevice(ordinal) == hipSuccess ? GSLNLS_SUCCESS : GSLNLS_E_NODEVICE;
}

const char *gslnls_version(void) { return "gslnls-mi355x 0.1 (gfx950)"; }

} // extern "C"
