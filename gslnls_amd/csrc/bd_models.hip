// bd_models.hip -- the translation unit of the "Jacobian as a matrix in HBM" path (bd_host.hpp, bd_kernels.hpp): gsl_nls() on
// R functions (host closures) and on formulas with more than 64 parameters (rows by a kernel compiled in process).
#include <hip/hip_runtime.h>
#include <map>
#include <memory>
#include <mutex>
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
    double *d_x = nullptr, *d_theta = nullptr, *d_ring = nullptr;
    std::shared_ptr<RtcEntry> rtc;
    hipFunction_t fn_mode[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}; // values | + J | D^2 m[v, v] | + J weighted and flagged | weighted residual + partial sums
    ~BdFormulaModel() override
    {
        if (d_x)
            bd_pool_free(d_x);
        if (d_theta)
            bd_pool_free(d_theta);
        if (d_ring)
            bd_pool_free(d_ring);
        if (h_ring)
            bd_pool_free(h_ring);
    }
    // theta / v are the caller's vectors and may change as soon as this returns: they go through a ring of pinned slots
    // (one asynchronous copy each, no synchronisation -- round 5; the ring is longer than the evaluations a Jacobian by
    // central differences enqueues between two synchronisations of the fit: 2 p + 2 <= 1026)
    static constexpr int RING = 1100;
    double *h_ring = nullptr;
    int ring_at = 0;
    int launch(int mode, const double *theta, const double *v, double *d_fval, double *d_J, hipStream_t st)
    {
        double *slot = h_ring + (size_t)ring_at * 2 * p;
        ring_at = (ring_at + 1) % RING;
        memcpy(slot, theta, sizeof(double) * p);
        if (v)
            memcpy(slot + p, v, sizeof(double) * p);
        // (the device copy of theta is per slot as well: a kernel still running must not see the next evaluation's theta)
        double *dth = d_ring + (size_t)((ring_at + RING - 1) % RING) * 2 * p;
        if (hipMemcpyAsync(dth, slot, sizeof(double) * (size_t)(v ? 2 * p : p), hipMemcpyHostToDevice, st) != hipSuccess)
            return 1;
        return launch_dev(mode, dth, dth + p, d_fval, d_J, st);
    }
    int launch_dev(int mode, const double *th, const double *dir, double *d_fval, double *d_J, hipStream_t st, const double *sw = nullptr,
                   double *part = nullptr, int *nparts = nullptr)
    {
        const double *xx = d_x;
        long long nn = n;
        int g = (int)((nn + 255) / 256);
        g = g > 2048 ? 2048 : (g < 1 ? 1 : g);
        if (mode == 3)
            g = g > BD_MAXG ? BD_MAXG : g; // (one flag per workgroup travels home with g and the diagonal)
        if (mode == 4)
            g = *nparts; // (bd_resid_kernel's grid: the partial sums are added on the host in that order)
        if (nparts)
            *nparts = g;
        void *args[] = {(void *)&th, (void *)&dir, (void *)&xx, (void *)&nn, (void *)&d_fval, (void *)&d_J, (void *)&sw, (void *)&part};
        return hipModuleLaunchKernel(fn_mode[mode], g, 1, 1, 256, 1, 1, 0, st, args, nullptr) == hipSuccess ? 0 : 1;
    }
    // model value, weighted residual and the partial sums of its squares in one kernel (theta on the device)
    int resid_dev(const double *d_theta, const double *d_y, const double *d_sw, double *d_f, double *d_parts, int g, hipStream_t st) override
    {
        if (!fn_mode[4])
            return -1;
        int gg = g;
        return launch_dev(4, d_theta, d_theta, d_f, const_cast<double *>(d_y), st, d_sw, d_parts, &gg);
    }
    // the Jacobian with the rows' sqrt(w) and the non-finite flags in one kernel; theta from the device when the caller has
    // it there (the accepted trial point of a fused trial step: no upload)
    int jacobian_flagged(const double *theta, const double *d_theta, double *d_J, const double *d_sw, double *d_part, int *nparts,
                         hipStream_t st) override
    {
        if (!fn_mode[3])
            return -1;
        if (d_theta)
            return launch_dev(3, d_theta, d_theta, d_scratch, d_J, st, d_sw, d_part, nparts);
        double *slot = h_ring + (size_t)ring_at * 2 * p;
        ring_at = (ring_at + 1) % RING;
        memcpy(slot, theta, sizeof(double) * p);
        double *dth = d_ring + (size_t)((ring_at + RING - 1) % RING) * 2 * p;
        if (hipMemcpyAsync(dth, slot, sizeof(double) * (size_t)p, hipMemcpyHostToDevice, st) != hipSuccess)
            return 1;
        return launch_dev(3, dth, dth, d_scratch, d_J, st, d_sw, d_part, nparts);
    }
    bool theta_on_device() const override { return true; }
    int values_dev(const double *d_th, double *d_fval, hipStream_t st) override { return launch_dev(0, d_th, d_th, d_fval, nullptr, st); }
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
    const double t_a = now_s();
    std::vector<std::string> pn(fn->parnames, fn->parnames + fn->p), vn(fn->xnames, fn->xnames + fn->nx);
    const int nxm = fn->nx > 0 ? fn->nx : 1;
    // The lowering of the expression (parse, symbolic gradient and second directional derivative, the row model as C++
    // text) is kept per formula: at p = 500 it is 280 ms of a call whose whole fit is 7 -- measured, round 5 -- and a
    // multi-start or a series of fits repeats it unchanged.  Key: expression + parameter and column names in order.
    struct Lowered
    {
        std::string src;
        int nfvv = 0;
    };
    static std::mutex cache_mu;
    static std::map<std::string, std::shared_ptr<Lowered>> cache;
    std::string key = std::string(fn->expr) + "\x01" + std::to_string(nxm);
    for (const std::string &q : pn)
        key += "\x02" + q;
    for (const std::string &q : vn)
        key += "\x03" + q;
    std::shared_ptr<Lowered> low;
    {
        std::lock_guard<std::mutex> lock(cache_mu);
        auto it = cache.find(key);
        if (it != cache.end())
            low = it->second;
    }
    if (!low)
    {
        auto prog = std::make_unique<BigProgram>();
        const std::string e = compile_expression_t(fn->expr, pn, vn, nxm, *prog);
        if (!e.empty())
        {
            fprintf(stderr, "gslnls: cannot lower model expression: %s\n", e.c_str());
            return GSLNLS_E_UNSUPPORTED;
        }
        low = std::make_shared<Lowered>();
        low->nfvv = prog->nfvv;
        low->src = "// generated by gslnls wide_models.hip (bd)\n#include \"bd_model_kernels.hpp\"\n";
        low->src += rtc_emit_model(*prog, nxm);
        std::lock_guard<std::mutex> lock(cache_mu);
        if (cache.size() >= 8)
            cache.erase(cache.begin());
        cache[key] = low;
    }
    if (fvv && low->nfvv == 0)
        return GSLNLS_E_UNSUPPORTED;
    const double t_b = now_s();
    BdFormulaModel m;
    m.n = n;
    m.p = fn->p;
    m.nx = nxm;
    m.has_jac = true;
    m.has_fvv = low->nfvv > 0;
    const std::string &src = low->src;
    std::vector<std::string> exprs = {rtc_bd_expr(0), rtc_bd_expr(1), rtc_bd_expr(3), rtc_bd_expr(4)};
    if (m.has_fvv)
        exprs.push_back(rtc_bd_expr(2));
    m.rtc = rtc_request(src, exprs, true);
    std::string msg;
    if (m.rtc->state.load() == RTC_READY)
        for (size_t k = 0; k < exprs.size(); ++k)
        {
            const int mode = k == 0 ? 0 : (k == 1 ? 1 : (k == 2 ? 3 : (k == 3 ? 4 : 2)));
            m.fn_mode[mode] = rtc_function(*m.rtc, exprs[k], msg);
        }
    else
        msg = m.rtc->log;
    if (!m.fn_mode[0] || !m.fn_mode[1] || !m.fn_mode[3] || !m.fn_mode[4] || (m.has_fvv && !m.fn_mode[2]))
    {
        fprintf(stderr, "gslnls: cannot build the kernels of a p = %d model (needs the in-process compiler): %s\n", fn->p, msg.c_str());
        return GSLNLS_E_UNSUPPORTED;
    }
    const double t_c = now_s();
    const size_t nb = sizeof(double) * (size_t)n;
    GSLNLS_HIP_OK(bd_dev_alloc(&m.d_x, nb * nxm));
    GSLNLS_HIP_OK(bd_dev_alloc(&m.d_theta, sizeof(double) * (size_t)2 * fn->p + nb));
    m.d_scratch = m.d_theta + 2 * fn->p;
    GSLNLS_HIP_OK(bd_dev_alloc(&m.d_ring, sizeof(double) * (size_t)BdFormulaModel::RING * 2 * fn->p));
    GSLNLS_HIP_OK(bd_host_alloc(&m.h_ring, sizeof(double) * (size_t)BdFormulaModel::RING * 2 * fn->p, hipHostMallocDefault));
    if (fn->nx > 0)
        GSLNLS_HIP_OK(hipMemcpy(m.d_x, fn->x, nb * nxm, hipMemcpyHostToDevice));
    else
        GSLNLS_HIP_OK(hipMemset(m.d_x, 0, nb));
    BdFit fit;
    int rc = fit.init(n, fn->p, y, swts, &m);
    if (rc)
        return rc;
    g_bd_prof.setup_ms = 1e3 * (now_s() - t_a);
    if (getenv("GSLNLS_LARGE_PROF"))
        fprintf(stderr, "[bd] p = %d: expression -> program %.2f ms, kernels (cache / in-process compiler) %.2f ms, buffers + upload %.2f ms\n", fn->p,
                1e3 * (t_b - t_a), 1e3 * (t_c - t_b), 1e3 * (now_s() - t_c));
    if (start_is_matrix)
        return bd_mstart(fit, jac, fvv, start, lupars, ci, cd, has_start, comm, loss_rho, loss_cc, out);
    if (loss_rho != 0)
        return fit.irls(jac, fvv, start, lupars, ci, cd, loss_rho, loss_cc, out);
    return fit.solve(jac, fvv, start, lupars, ci, cd, out);
}


// J^T J of a dense n x p column-major block for the large path's dense operator (sparse_large.hpp): the matrix path's
// SYRK on the matrix cores.  The scratch for the partial blocks belongs to the caller (grown here when too small).
int bd_dense_jtj(const double *d_J, int n, int p, double *d_C, hipStream_t st, double **cpart, size_t *cpart_bytes)
{
    const BdSyrkGeom geom = bd_syrk_geom(n, p);
    const size_t need = sizeof(double) * geom.scratch;
    if (*cpart_bytes < need)
    {
        if (*cpart)
            (void)hipFree(*cpart);
        *cpart = nullptr;
        *cpart_bytes = 0;
        GSLNLS_HIP_OK(hipMalloc(cpart, need));
        *cpart_bytes = need;
    }
    bd_syrk_launch(d_J, (long long)n, p, d_C, *cpart, geom, st);
    return GSLNLS_SUCCESS;
}

// gslnls_last_matrix_path_profile / gslnls_debug_bd_syrk_ms (include/gslnls_core.h)
int bd_last_profile(double *v, int cap)
{
    const double a[12] = {g_bd_prof.setup_ms, g_bd_prof.loop_ms, g_bd_prof.solve_ms, g_bd_prof.jac_ms, g_bd_prof.resid_ms, g_bd_prof.covar_ms,
                          g_bd_prof.down_ms,  g_bd_prof.cond_ms, (double)g_bd_prof.trial_steps, (double)g_bd_prof.jacobians,
                          (double)g_bd_prof.fused, (double)g_bd_prof.p};
    for (int k = 0; k < cap && k < 12; ++k)
        v[k] = a[k];
    return 12;
}
double bd_syrk_ms(int n, int p, int reps) { return bd_time_syrk(n, p, reps); }
void bd_trim_pool() { bd_pool().drain(); }

// (J^T J)^-1 of a p x p SPD matrix by the device routine above, for the large path's covariance (capi.hip): from where the
// matrix sits on the device, or uploaded from the host first.  Non-zero: the caller keeps its host routine.
int bd_spd_inverse(int p, const double *d_A, const double *A_host, double *covar_host)
{
    if (d_A)
        return bd_device_inverse(p, d_A, covar_host, nullptr);
    if (!A_host)
        return GSLNLS_EINVAL;
    double *d_tmp = nullptr;
    if (bd_dev_alloc(&d_tmp, sizeof(double) * (size_t)p * p) != hipSuccess)
    {
        (void)hipGetLastError();
        return GSLNLS_E_NODEVICE;
    }
    int rc = hipMemcpy(d_tmp, A_host, sizeof(double) * (size_t)p * p, hipMemcpyHostToDevice) == hipSuccess ? 0 : GSLNLS_E_NODEVICE;
    if (!rc)
        rc = bd_device_inverse(p, d_tmp, covar_host, nullptr);
    bd_pool_free(d_tmp);
    return rc;
}

} // namespace gslnls
