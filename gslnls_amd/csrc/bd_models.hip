// bd_models.hip -- the translation unit of the "Jacobian as a matrix in HBM" path (bd_host.hpp, bd_kernels.hpp): gsl_nls() on
// R functions (host closures) and on formulas with more than 64 parameters (rows by a kernel compiled in process).
#include <hip/hip_runtime.h>
#include <memory>
#include <string>
#include <vector>
#include "../../include/gslnls_core.h"
#include "expr_compile.hpp"
#include "rtc_host.hpp"
#include "bd_host.hpp"

namespace gslnls
{

// gslnls_nls_fn (capi.hip): the closures of a function model
// start_is_matrix: 2 x p ranges + has_start -> the multi-start procedure (bd_mstart), through `comm` when one is bound
int bd_callback_nls(int n, int p, const double *y, gslnls_fn_cb f, gslnls_jac_cb jac, gslnls_fvv_cb fvv, void *user,
                    const double *start, int start_is_matrix, const int *has_start, const MsComm &comm, const double *swts,
                    const double *lupars, const int *ci, const double *cd, int loss_rho, const double *loss_cc, gslnls_result *out)
{
    BdCallbackModel m;
    m.n = n;
    m.p = p;
    m.f = f;
    m.jac = jac;
    m.fv = fvv;
    m.user = user;
    m.has_jac = jac != nullptr;
    m.has_fvv = fvv != nullptr;
    BdFit fit;
    int rc = fit.init(n, p, y, swts, &m);
    if (rc)
        return rc;
    if (start_is_matrix)
        return bd_mstart(fit, jac != nullptr, fvv != nullptr, start, lupars, ci, cd, has_start, comm, loss_rho, loss_cc, out);
    if (loss_rho != 0)
        return fit.irls(jac != nullptr, fvv != nullptr, start, lupars, ci, cd, loss_rho, loss_cc, out);
    return fit.solve(jac != nullptr, fvv != nullptr, start, lupars, ci, cd, out);
}

// ---- formulas with more than 64 parameters: the Jacobian as a matrix in HBM (bd_host.hpp) --------------------------------
inline std::string rtc_bd_expr(int mode) { return "&gslnls::bd_model_kernel<gslnls::ModelJit, " + std::to_string(mode) + ">"; }

struct BdFormulaModel : BdModel
{
    int n = 0, p = 0, nx = 1;
    double *d_x = nullptr, *d_theta = nullptr;
    std::shared_ptr<RtcEntry> rtc;
    hipFunction_t fn_mode[3] = {nullptr, nullptr, nullptr};
    ~BdFormulaModel() override
    {
        if (d_x)
            (void)hipFree(d_x);
        if (d_theta)
            (void)hipFree(d_theta);
    }
    int launch(int mode, const double *theta, const double *v, double *d_fval, double *d_J, hipStream_t st)
    {
        if (hipMemcpyAsync(d_theta, theta, sizeof(double) * p, hipMemcpyHostToDevice, st) != hipSuccess)
            return 1;
        if (v && hipMemcpyAsync(d_theta + p, v, sizeof(double) * p, hipMemcpyHostToDevice, st) != hipSuccess)
            return 1;
        // (theta / v are the caller's vectors: the copies have to be over before it changes them)
        if (hipStreamSynchronize(st) != hipSuccess)
            return 1;
        const double *th = d_theta, *dir = d_theta + p, *xx = d_x;
        long long nn = n;
        int g = (int)((nn + 255) / 256);
        g = g > 2048 ? 2048 : (g < 1 ? 1 : g);
        void *args[] = {(void *)&th, (void *)&dir, (void *)&xx, (void *)&nn, (void *)&d_fval, (void *)&d_J};
        return hipModuleLaunchKernel(fn_mode[mode], g, 1, 1, 256, 1, 1, 0, st, args, nullptr) == hipSuccess ? 0 : 1;
    }
    int values(const double *theta, double *d_fval, hipStream_t st) override { return launch(0, theta, nullptr, d_fval, nullptr, st); }
    int jacobian(const double *theta, double *d_J, hipStream_t st) override
    {
        // (the values written beside the Jacobian go to the theta buffer's tail: nobody reads them)
        return launch(1, theta, nullptr, d_scratch, d_J, st);
    }
    int fvv(const double *theta, const double *v, double *d_out, hipStream_t st) override { return launch(2, theta, v, d_out, nullptr, st); }
    double *d_scratch = nullptr;
};

// gsl_nls() on a formula with 64 < p <= 512 parameters; robust losses through BdFit::irls, start ranges through bd_mstart
int bd_formula_nls(const gslnls_model *fn, const double *y, int n, int jac, int fvv, const double *start, int start_is_matrix,
                   const int *has_start, const MsComm &comm, const double *swts, const double *lupars, const int *ci,
                   const double *cd, int loss_rho, const double *loss_cc, gslnls_result *out)
{
    if (!fn->expr || !fn->parnames || (fn->nx > 0 && !fn->xnames) || fn->nx > WIDE_NX || fn->p > BIG_MAX_P || fn->x_on_device)
        return fn->p > BIG_MAX_P || fn->nx > WIDE_NX || fn->x_on_device ? GSLNLS_E_UNSUPPORTED : GSLNLS_EINVAL;
    std::vector<std::string> pn(fn->parnames, fn->parnames + fn->p), vn(fn->xnames, fn->xnames + fn->nx);
    auto prog = std::make_unique<BigProgram>();
    const int nxm = fn->nx > 0 ? fn->nx : 1;
    const std::string e = compile_expression_t(fn->expr, pn, vn, nxm, *prog);
    if (!e.empty())
    {
        fprintf(stderr, "gslnls: cannot lower model expression: %s\n", e.c_str());
        return GSLNLS_E_UNSUPPORTED;
    }
    if (fvv && prog->nfvv == 0)
        return GSLNLS_E_UNSUPPORTED;
    BdFormulaModel m;
    m.n = n;
    m.p = fn->p;
    m.nx = nxm;
    m.has_jac = true;
    m.has_fvv = prog->nfvv > 0;
    std::string src = "// generated by gslnls wide_models.hip (bd)\n#include \"bd_model_kernels.hpp\"\n";
    src += rtc_emit_model(*prog, nxm);
    std::vector<std::string> exprs = {rtc_bd_expr(0), rtc_bd_expr(1)};
    if (m.has_fvv)
        exprs.push_back(rtc_bd_expr(2));
    m.rtc = rtc_request(src, exprs, true);
    std::string msg;
    if (m.rtc->state.load() == RTC_READY)
        for (size_t k = 0; k < exprs.size(); ++k)
            m.fn_mode[k] = rtc_function(*m.rtc, exprs[k], msg);
    else
        msg = m.rtc->log;
    if (!m.fn_mode[0] || !m.fn_mode[1] || (m.has_fvv && !m.fn_mode[2]))
    {
        fprintf(stderr, "gslnls: cannot build the kernels of a p = %d model (needs the in-process compiler): %s\n", fn->p, msg.c_str());
        return GSLNLS_E_UNSUPPORTED;
    }
    const size_t nb = sizeof(double) * (size_t)n;
    GSLNLS_HIP_OK(hipMalloc(&m.d_x, nb * nxm));
    GSLNLS_HIP_OK(hipMalloc(&m.d_theta, sizeof(double) * (size_t)2 * fn->p + nb));
    m.d_scratch = m.d_theta + 2 * fn->p;
    if (fn->nx > 0)
        GSLNLS_HIP_OK(hipMemcpy(m.d_x, fn->x, nb * nxm, hipMemcpyHostToDevice));
    else
        GSLNLS_HIP_OK(hipMemset(m.d_x, 0, nb));
    BdFit fit;
    int rc = fit.init(n, fn->p, y, swts, &m);
    if (rc)
        return rc;
    if (start_is_matrix)
        return bd_mstart(fit, jac, fvv, start, lupars, ci, cd, has_start, comm, loss_rho, loss_cc, out);
    if (loss_rho != 0)
        return fit.irls(jac, fvv, start, lupars, ci, cd, loss_rho, loss_cc, out);
    return fit.solve(jac, fvv, start, lupars, ci, cd, out);
}


} // namespace gslnls
