// wide_models.hip -- the wide dense path (10 <= p <= 64): the statically compiled kernels of wide_host.hpp (reduce,
// advance, the solve test hook) and the factory the expression dispatcher (vm_models.hip) calls.
#include <hip/hip_runtime.h>
#include <memory>
#include <string>
#include <vector>
#include "../../include/gslnls_core.h"
#include "expr_compile.hpp"
#include "wide_host.hpp"

namespace gslnls
{

DenseBase *make_dense_wide(const gslnls_model *fn, const double *y, int n, const double *swts, int *err)
{
    std::vector<std::string> pn(fn->parnames, fn->parnames + fn->p), vn(fn->xnames, fn->xnames + fn->nx);
    auto *prog = new WideProgram;
    const std::string e = compile_expression_t(fn->expr, pn, vn, fn->nx > 0 ? fn->nx : 1, *prog);
    if (!e.empty())
    {
        fprintf(stderr, "gslnls: cannot lower model expression: %s\n", e.c_str());
        delete prog;
        *err = GSLNLS_E_UNSUPPORTED;
        return nullptr;
    }
    auto *d = new WideFit();
    *err = d->init(*prog, fn, y, n, swts);
    delete prog;
    if (*err != GSLNLS_SUCCESS)
    {
        delete d;
        return nullptr;
    }
    return d;
}

// ahead-of-time build of the wide kernels of an expression (gslnls_expr_build): analytic + forward-difference units
int wide_expr_build(const gslnls_model *fn, std::string &first_path)
{
    std::vector<std::string> pn(fn->parnames, fn->parnames + fn->p), vn(fn->xnames, fn->xnames + fn->nx);
    auto prog = std::make_unique<WideProgram>();
    const std::string e = compile_expression_t(fn->expr, pn, vn, fn->nx > 0 ? fn->nx : 1, *prog);
    if (!e.empty())
    {
        fprintf(stderr, "gslnls: cannot lower model expression: %s\n", e.c_str());
        return GSLNLS_E_UNSUPPORTED;
    }
    const std::string src = rtc_wide_source(*prog, fn->nx > 0 ? fn->nx : 1);
    const int PW = 16 * ((fn->p + 15) / 16);
    for (int jm = 0; jm < 2; ++jm)
    {
        auto ent = rtc_request(src, {rtc_wide_pass_expr(jm, PW), rtc_wide_finalize_expr(jm)}, true);
        if (ent->state.load() != RTC_READY)
        {
            fprintf(stderr, "gslnls: native lowering failed: %s\n", ent->log.c_str());
            return GSLNLS_E_UNSUPPORTED;
        }
        if (jm == 0)
            first_path = ent->cache_path;
        // the one-launch-per-step kernel (wide_kernels.hpp): the unit fits otherwise wait for in the background
        auto step = rtc_request(src, {rtc_wide_step_expr(jm, PW)}, true);
        if (step->state.load() != RTC_READY)
        {
            fprintf(stderr, "gslnls: native lowering (step kernel) failed: %s\n", step->log.c_str());
            return GSLNLS_E_UNSUPPORTED;
        }
        // the one-workgroup-per-fit kernel of the multi-start evaluator
        auto fitk = rtc_request(src, {rtc_wide_fit_expr(jm, PW)}, true);
        if (fitk->state.load() != RTC_READY)
        {
            fprintf(stderr, "gslnls: native lowering (fit kernel) failed: %s\n", fitk->log.c_str());
            return GSLNLS_E_UNSUPPORTED;
        }
    }
    return GSLNLS_SUCCESS;
}

} // namespace gslnls

extern "C" int gslnls_debug_wide_sums(gslnls_dense *h, int jac, int fdtype, const double *theta, double *totals)
{
    // (gslnls_dense is { DenseBase *impl; }: capi.hip)
    struct Handle
    {
        gslnls::DenseBase *impl;
    };
    auto *w = h ? dynamic_cast<gslnls::WideFit *>(reinterpret_cast<Handle *>(h)->impl) : nullptr;
    if (!w || !theta || !totals)
        return GSLNLS_EINVAL;
    return w->sums_at(jac, fdtype, theta, totals);
}

extern "C" int gslnls_debug_wide_solve(int p, const double *Ap, const double *diag, double mu, const double *rhs, double *sol)
{
    using namespace gslnls;
    if (p < 1 || p > WP || !Ap || !diag || !rhs || !sol)
        return GSLNLS_EINVAL;
    const int na = p * (p + 1) / 2;
    // (read at every call: the tests exercise both orders of elimination)
    const int pivoted = getenv("GSLNLS_WIDE_PIVOTED") && atoi(getenv("GSLNLS_WIDE_PIVOTED")) != 0;
    double *d = nullptr;
    GSLNLS_HIP_OK(hipMalloc(&d, sizeof(double) * (size_t)(na + 3 * p)));
    hipError_t he = hipMemcpy(d, Ap, sizeof(double) * na, hipMemcpyHostToDevice);
    if (he == hipSuccess)
        he = hipMemcpy(d + na, diag, sizeof(double) * p, hipMemcpyHostToDevice);
    if (he == hipSuccess)
        he = hipMemcpy(d + na + p, rhs, sizeof(double) * p, hipMemcpyHostToDevice);
    if (he == hipSuccess)
    {
        hipLaunchKernelGGL(wide_solve_debug_kernel, dim3(1), dim3(64), 0, 0, p, d, d + na, mu, d + na + p, d + na + 2 * p, pivoted);
        he = hipMemcpy(sol, d + na + 2 * p, sizeof(double) * p, hipMemcpyDeviceToHost);
    }
    if (he != hipSuccess)
    {
        (void)hipFree(d);
        return GSLNLS_E_NODEVICE;
    }
    if (const char *e = getenv("GSLNLS_WIDE_SOLVE_REPS"))
    {
        // developer timing of the solve alone (HIP events around `reps` back-to-back launches)
        const int reps = atoi(e) > 0 ? atoi(e) : 1;
        hipEvent_t e0, e1;
        GSLNLS_HIP_OK(hipEventCreate(&e0));
        GSLNLS_HIP_OK(hipEventCreate(&e1));
        (void)hipEventRecord(e0, 0);
        for (int r = 0; r < reps; ++r)
            hipLaunchKernelGGL(wide_solve_debug_kernel, dim3(1), dim3(64), 0, 0, p, d, d + na, mu, d + na + p, d + na + 2 * p, pivoted);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        fprintf(stderr, "[wide solve] p = %d: %.2f us per launch (incl. ~2 us launch boundary)\n", p, 1e3 * ms / reps);
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
    }
    hipFree(d);
    return GSLNLS_SUCCESS;
}
