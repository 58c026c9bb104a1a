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
