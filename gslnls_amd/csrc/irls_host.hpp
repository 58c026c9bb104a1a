// irls_host.hpp -- gsl_multifit_nlinear_rho_driver (src/nls_irls.c:412-546) around the device solve.
//
// Per IRLS iteration the host enqueues: cold re-start of the LM loop from the ORIGINAL start with the
// current weights (App. D quirk, src/nls_irls.c:447-456), then the re-weighting chain of
// irls_kernels.hpp.  It reads back only the p-sized state to apply the stopping rule
// test_delta_irls (src/nls_irls.c:343-362); residuals, weights, psi, psi' stay in HBM until the end.
#pragma once
#include "trace_log.hpp"
#include "dense_host.hpp"
#include "irls_kernels.hpp"

namespace gslnls
{

template <class M>
int DenseFit<M>::irls(int jac, int fvv, const double *start, const double *lupars, const int *ci, const double *cd,
                      int loss_rho, const double *loss_cc, gslnls_result *out)
{
    const bool trace = ci[1] != 0 && out->ssrtrace && out->partrace;
    int rc = prepare(jac, fvv, lupars, ci, cd, trace);
    if (rc)
        return rc;
    const int jacmode = jac ? JAC_ANALYTIC : (ci[5] ? JAC_CENTER : JAC_FORWARD);
    const int irls_maxiter = ci[14];
    const double irls_xtol = cd[10];
    LossCfg L;
    L.rho = loss_rho;
    {
        static const int ncc[9] = {0, 1, 2, 1, 1, 1, 1, 3, 3};
        for (int k = 0; k < 3; ++k)
            L.cc[k] = (loss_rho >= 1 && loss_rho <= 8 && k < ncc[loss_rho]) ? loss_cc[k] : 0.0;
    }
    const size_t nb = sizeof(double) * (size_t)n;
    const double *user_sw = ctx.sw; // sqrt of the user weights (or nullptr)
    double *d_r = nullptr, *d_wt = nullptr, *d_psi = nullptr, *d_psip = nullptr, *d_swA = nullptr, *d_swB = nullptr,
           *d_part = nullptr;
    unsigned long long *d_keys = nullptr;
    SelectState *d_sel = nullptr;
    IrlsScalars *d_sc = nullptr;
    constexpr int TW = 256;
    int nblk = (int)((n + TW - 1) / TW);
    if (nblk > 1024)
        nblk = 1024;
    // one arena kept by the problem (and, through the pool of parked problems, by the next one-shot call): ten
    // hipMalloc / hipFree pairs per robust fit cost ~3 ms, more than the fit of a small data set
    {
        auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
        const size_t need = 6 * up(nb) + up(sizeof(unsigned long long) * (size_t)n) + up(sizeof(double) * nblk) +
                            up(sizeof(SelectState) * 2) + up(sizeof(IrlsScalars));
        if (irls_arena_bytes < need)
        {
            hipFree(irls_arena);
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
        d_r = reinterpret_cast<double *>(take(nb));
        d_wt = reinterpret_cast<double *>(take(nb));
        d_psi = reinterpret_cast<double *>(take(nb));
        d_psip = reinterpret_cast<double *>(take(nb));
        d_swA = reinterpret_cast<double *>(take(nb));
        d_swB = reinterpret_cast<double *>(take(nb));
        d_keys = reinterpret_cast<unsigned long long *>(take(sizeof(unsigned long long) * (size_t)n));
        d_part = reinterpret_cast<double *>(take(sizeof(double) * nblk));
        d_sel = reinterpret_cast<SelectState *>(take(sizeof(SelectState) * 2));
        d_sc = reinterpret_cast<IrlsScalars *>(take(sizeof(IrlsScalars)));
    }
    auto cleanup = [&]() { ctx.sw = user_sw; };

    // the solver always runs weighted under a robust loss (gsl_multifit_nlinear_winit, src/nls.c:542-543):
    // first solve with the user weights (or ones)
    double *cur_sw = d_swA, *next_sw = d_swB;
    if (user_sw)
        GSLNLS_HIP_OK(hipMemcpyAsync(cur_sw, user_sw, nb, hipMemcpyDeviceToDevice, stream));
    else
    {
        std::vector<double> ones((size_t)n, 1.0);
        GSLNLS_HIP_OK(hipMemcpy(cur_sw, ones.data(), nb, hipMemcpyHostToDevice));
    }

    std::vector<double> workp(P), xprev(P);
    int irls_iter = 0, irls_status = ST_FAILURE, status = ST_CONTINUE;
    double chisq_init = NAN, chisq_carry = NAN, sigma = 1.0;
    long long total_launches = 0;
    float total_ms = 0.f;
    do
    {
        irls_iter += 1;
        if (irls_iter > 1)
        {
            std::swap(cur_sw, next_sw); // the weights computed at the end of the previous iteration
            for (int k = 0; k < P; ++k)
                workp[k] = h_state[0].x[k];
        }
        else
            for (int k = 0; k < P; ++k)
                workp[k] = start[k];
        ctx.sw = cur_sw;
        ctx.prm.has_weights = 1;
        ctx.prm.chisq_in = (irls_iter > 1) ? chisq_carry : NAN;
        rc = run_loop(jacmode, start, lupars, 0);
        if (rc)
        {
            cleanup();
            return rc;
        }
        total_launches += last_launches;
        total_ms += last_ms;
        const LmState<P> &s = h_state[0];
        status = s.status;
        if (irls_iter == 1)
            chisq_init = s.chisq_init;
        chisq_carry = s.chisq1;
        trace_printf("IRLS iter: %3d, weighted ssr: %g, par: (", irls_iter, s.chisq1); // (src/nls_irls.c:466-472)
        trace_vector(s.x, P);
        if (status == ST_EBADFUNC || (status == ST_ENOPROG && irls_iter == 1))
            break;

        // ---- re-weighting chain, all on device ----
        const int Gf = std::min(2048, (int)((n + T - 1) / T));
        hipLaunchKernelGGL((irls_resid_kernel<M, T>), dim3(Gf), dim3(T), 0, stream, ctx, last_parity, d_r, d_keys);
        const unsigned long long k_lo = (unsigned long long)((n - 1) / 2), k_hi = (unsigned long long)(n / 2);
        const int nsel = (k_lo == k_hi) ? 1 : 2;
        for (int which = 0; which < nsel; ++which)
        {
            SelectState *st = d_sel + which;
            hipLaunchKernelGGL(select_init_kernel, dim3(1), dim3(1), 0, stream, st, which == 0 ? k_lo : k_hi);
            for (int pass = 7; pass >= 0; --pass)
            {
                hipLaunchKernelGGL(select_hist_kernel, dim3(std::min(1024, (int)((n + 255) / 256))), dim3(256), 0, stream,
                                   d_keys, (long long)n, pass, st);
                hipLaunchKernelGGL(select_pick_kernel, dim3(1), dim3(1), 0, stream, pass, st);
            }
        }
        hipLaunchKernelGGL(irls_sigma_kernel, dim3(1), dim3(1), 0, stream, d_sel, d_sel + (nsel - 1), d_sc);
        hipLaunchKernelGGL((irls_weight_kernel<TW>), dim3(nblk), dim3(TW), 0, stream, d_r, (long long)n, L, d_sc, d_wt,
                           d_psi, d_psip, d_part);
        hipLaunchKernelGGL(irls_scale_kernel, dim3(1), dim3(1), 0, stream, d_part, nblk, (long long)n, d_sc);
        hipLaunchKernelGGL((irls_apply_kernel<TW>), dim3(nblk), dim3(TW), 0, stream, d_wt, (long long)n, d_sc, user_sw,
                           next_sw);
        IrlsScalars hsc;
        GSLNLS_HIP_OK(hipMemcpyAsync(&hsc, d_sc, sizeof(hsc), hipMemcpyDeviceToHost, stream));
        GSLNLS_HIP_OK(hipStreamSynchronize(stream));
        sigma = hsc.sigma;

        // test_delta_irls (src/nls_irls.c:343-362)
        irls_status = ST_CONTINUE;
        for (int k = 0; k < P; ++k)
        {
            const double xi = s.x[k], dxi = fabs(workp[k] - xi);
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

    // after a converged / exhausted loop cur_sw are the weights of the LAST SOLVE (what resid / grad are
    // reported with, src/nls.c:695-737) and d_wt holds the NEW irls weights (src/nls.c:587)
    if (!(status == ST_EBADFUNC || (status == ST_ENOPROG && irls_iter == 1)))
    {
        if (irls_iter >= irls_maxiter && irls_status != ST_SUCCESS)
        {
            irls_status = ST_EMAXITER;
            h_state[0].status = ST_EMAXITER;
            h_state[0].info = ST_EMAXITER;
        }
    }
    ctx.sw = cur_sw;
    h_state[0].chisq_init = chisq_init;
    last_launches = total_launches;
    last_ms = total_ms;
    const LmState<P> fin = h_state[0];
    rc = pack(jacmode, start, out, trace);
    const bool ok = (fin.status == ST_SUCCESS || fin.status == ST_EMAXITER);
    double irls_delta = 0.0;
    for (int k = 0; k < P; ++k)
        irls_delta = fmax(irls_delta, fabs(workp[k] - fin.x[k]));
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
    cleanup();
    return rc;
}

} // namespace gslnls
