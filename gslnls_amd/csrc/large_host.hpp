// large_host.hpp -- what runs behind .Call(C_nls_large) (src/nls_large.c:77-424): the
// gsl_multilarge_nlinear trust-region iteration on the normal equations with the Steihaug-Toint
// CG subproblem (control_int[2] == 5) or LM (== 0), iterated by gsl_multilarge_nlinear_driver2
// (src/nls_fit.c:153-224).
//
// The p-sized control logic (CG recurrences, tau on the trust-region boundary, rho, delta, mu,
// scaling, convergence test; GSL multilarge trust.c / cgst.c / lm.c / scaling.c restated from
// SURVEY.md App. A.6) is host code as in the reference; every n-sized operation is one of the
// two device passes of large_kernels.hpp.  Residual, m = exp(A x) and the row data never leave
// HBM during the iteration.
#pragma once
#include "trace_log.hpp"
#include <hip/hip_runtime.h>
#include <math.h>
#include <float.h>
#include <vector>
#include <chrono>
#include "dense_host.hpp"
#include "large_kernels.hpp"

namespace gslnls
{

struct LargeOps
{
    int n = 0, p = 0;
    long nevalf = 0, nevaldfu = 0, nevaldf2 = 0;
    float pass_ms = 0.f; // HIP-event time of the last pass kernel
    long npass = 0;
    virtual ~LargeOps() {}
    // EVAL pass at x (trial buffers): ssr, g = J^T f, diag(J^T J) [and full J^T J when jtj != nullptr]
    virtual int eval(const double *x, double *ssr, double *g, double *diag, double *jtj, double *bad) = 0;
    virtual void accept() = 0; // the last EVAL point becomes the current point
    // Operators whose Jacobian is expensive to obtain (host callbacks) set lazy_jac: eval() then returns ssr only
    // and eval_jac() completes g, diag [, J^T J] for the same point once the step is accepted -- the reference's
    // own order (f at every trial, df only after acceptance).
    bool lazy_jac = false;
    virtual int eval_jac(double *, double *, double *) { return GSLNLS_E_UNSUPPORTED; }
    // JTJV pass at the current point xcur: ||J u||^2 and J^T J u
    virtual int jtjv(const double *xcur, const double *u, double *normw2, double *out) = 0;
    // optional: a whole Steihaug-Toint CG step with the p-sized recurrences on the device (sparse_cg.hpp): dx and the
    // step's status (ST_SUCCESS / ST_EMAXITER); the backend adds the products it made to nevaldfu.  cached_njdx2:
    // ||J dx||^2 of that step when the backend already computed it behind the step
    bool device_cg = false;
    virtual int cgst_device(const double *, const double *, double, long, double *, int *) { return GSLNLS_E_UNSUPPORTED; }
    virtual bool cached_njdx2(double *) { return false; }
    virtual int full_jtj(const double *xcur, double *jtj) = 0;          // p x p row-major (symmetric)
    // J^T J of the CURRENT point as it sits on the device (complete: no work pending on it), or nullptr
    virtual const double *jtj_device() { return nullptr; }
    // Operators that can form J^T J on the device and leave it there: with jtj_device_only set, eval_jac / full_jtj compute
    // it but do not copy the p x p matrix to the host (2 MB per accepted point at p = 500); jtj_download fetches the
    // current one when the host needs it after all (the host factorisation as a fallback)
    virtual bool can_keep_jtj_on_device() const { return false; }
    bool jtj_device_only = false;
    virtual int jtj_download(double *) { return GSLNLS_E_UNSUPPORTED; }
    virtual int residual(const double *xcur, double *resid_host) = 0;  // weighted residual at the current point
};

inline double lg_nrm2(int n, const double *x)
{
    double scale = 0.0, ssq = 1.0;
    for (int i = 0; i < n; ++i)
        if (x[i] != 0.0)
        {
            const double a = fabs(x[i]);
            if (isinf(a))
                return INFINITY;
            if (scale < a)
            {
                ssq = 1.0 + ssq * (scale / a) * (scale / a);
                scale = a;
            }
            else
                ssq += (a / scale) * (a / scale);
        }
    return scale * sqrt(ssq);
}

// plain Cholesky helpers on a p x p row-major matrix (host, p-sized)
inline bool lg_chol(int p, std::vector<double> &A)
{
    for (int j = 0; j < p; ++j)
    {
        double ajj = A[j * p + j];
        for (int k = 0; k < j; ++k)
            ajj -= A[j * p + k] * A[j * p + k];
        if (!(ajj > 0.0))
            return false;
        ajj = sqrt(ajj);
        A[j * p + j] = ajj;
        for (int i = j + 1; i < p; ++i)
        {
            double s = A[i * p + j];
            for (int k = 0; k < j; ++k)
                s -= A[i * p + k] * A[j * p + k];
            A[i * p + j] = s / ajj;
        }
    }
    return true;
}

inline void lg_chol_invert(int p, std::vector<double> &A)
{
    // L^-1 is kept transposed (LiT[j][k] = (L^-1)[k][j]) so that both inner loops run over contiguous memory; the sums
    // are taken in the same order as in the textbook form (p = 500: 20 ms -> a few)
    std::vector<double> LiT((size_t)p * p, 0.0);
    for (int j = 0; j < p; ++j)
    {
        double *col = &LiT[(size_t)j * p];
        col[j] = 1.0 / A[(size_t)j * p + j];
        for (int i = j + 1; i < p; ++i)
        {
            const double *row = &A[(size_t)i * p];
            double s = 0.0;
            for (int k = j; k < i; ++k)
                s -= row[k] * col[k];
            col[i] = s / row[i];
        }
    }
    for (int i = 0; i < p; ++i)
        for (int j = 0; j <= i; ++j)
        {
            const double *a = &LiT[(size_t)i * p], *b = &LiT[(size_t)j * p];
            double s = 0.0;
            for (int k = i; k < p; ++k)
                s += a[k] * b[k];
            A[(size_t)i * p + j] = A[(size_t)j * p + i] = s;
        }
}

// GSL's modified Cholesky (Gill-Murray-Wright, pivoted) for the multilarge LM step; same algorithm as
// lm_solve<P> in lm_core.hpp with run-time p
// (wider vectors where the host has them -- the inner loops are contiguous; contraction is off so that every clone rounds
// like the baseline one: avx512f would bring fused multiply-adds with it.  Raw pointers only in here: a clone cannot
// count on the out-of-line copies of inline library templates)
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
__attribute__((target_clones("default", "avx2", "avx512f")))
#endif
inline void lg_mchol_core(int p, double *M, double *b, double *cj, int *perm)
{
#pragma clang fp contract(off)
    // Only the lower triangle is kept (row-major): the symmetric interchange touches each stored element once, and
    // the rank-one update of step j reads column j from a contiguous copy and walks rows -- no mirrored stores down
    // columns, so the inner loop is contiguous and vectorises (p = 500: ~10 ms per solve before).  Same operations on
    // the same values in the same order as the full-matrix form this replaces.
    auto swp = [](double &x, double &y) {
        const double t = x;
        x = y;
        y = t;
    };
    for (int i = 0; i < p; ++i)
        perm[i] = i;
    double gamma = 0.0, xi = 0.0;
    for (int i = 0; i < p; ++i)
    {
        gamma = fmax(gamma, fabs(M[(size_t)i * p + i]));
        for (int j = 0; j < i; ++j)
            xi = fmax(xi, fabs(M[(size_t)i * p + j]));
    }
    double beta = (p == 1) ? fmax(fmax(gamma, xi), DBL_EPSILON)
                           : fmax(fmax(gamma, xi / sqrt((double)p * p - 1.0)), DBL_EPSILON);
    beta = sqrt(beta);
    for (int j = 0; j < p; ++j)
    {
        int q = j;
        double maxd = fabs(M[(size_t)j * p + j]);
        for (int i = j + 1; i < p; ++i)
            if (fabs(M[(size_t)i * p + i]) > maxd)
            {
                maxd = fabs(M[(size_t)i * p + i]);
                q = i;
            }
        if (q != j)
        {
            // rows / columns j and q (q > j) of the symmetric matrix, in the lower triangle:
            // (j,j) <-> (q,q);  (j,k) <-> (q,k) for k < j;  (k,j) <-> (q,k) for j < k < q;  (k,j) <-> (k,q) for k > q
            swp(M[(size_t)j * p + j], M[(size_t)q * p + q]);
            for (int k = 0; k < j; ++k)
                swp(M[(size_t)j * p + k], M[(size_t)q * p + k]);
            for (int k = j + 1; k < q; ++k)
                swp(M[(size_t)k * p + j], M[(size_t)q * p + k]);
            for (int k = q + 1; k < p; ++k)
                swp(M[(size_t)k * p + j], M[(size_t)k * p + q]);
            swp(b[j], b[q]);
            const int tp = perm[j];
            perm[j] = perm[q];
            perm[q] = tp;
        }
        double theta = 0.0;
        for (int i = j + 1; i < p; ++i)
        {
            cj[i] = M[(size_t)i * p + j];
            theta = fmax(theta, fabs(cj[i]));
        }
        const double u = theta / beta;
        const double alpha = fmax(fmax(DBL_EPSILON, fabs(M[(size_t)j * p + j])), u * u);
        const double ainv = 1.0 / alpha;
        for (int i = j + 1; i < p; ++i)
        {
            const double vi = cj[i];
            double *row = &M[(size_t)i * p];
            for (int k = j + 1; k <= i; ++k)
                row[k] -= ainv * vi * cj[k];
        }
        for (int i = j + 1; i < p; ++i)
            M[(size_t)i * p + j] = cj[i] * ainv;
        M[(size_t)j * p + j] = alpha;
    }
    for (int i = 0; i < p; ++i)
    {
        const double *row = &M[(size_t)i * p];
        for (int j = 0; j < i; ++j)
            b[i] -= row[j] * b[j];
    }
    for (int i = 0; i < p; ++i)
        b[i] /= M[(size_t)i * p + i];
    for (int i = p - 1; i >= 0; --i)
        for (int j = i + 1; j < p; ++j)
            b[i] -= M[(size_t)j * p + i] * b[j];
}

inline void lg_mchol_solve(int p, const std::vector<double> &Ain, const std::vector<double> &rhs, std::vector<double> &sol)
{
    std::vector<double> M(Ain), b(rhs), cj(p);
    std::vector<int> perm(p);
    lg_mchol_core(p, M.data(), b.data(), cj.data(), perm.data());
    sol.assign(p, 0.0);
    for (int i = 0; i < p; ++i)
        sol[perm[i]] = b[i];
}

// the same solve on the device (mchol_device.hip); GSLNLS_SUCCESS, or an error code when the device cannot take it
int mchol_device_solve(int p, const double *A_host, const double *rhs_host, double *sol_host);
// ... with J^T J where the operators left it on the device (A = J^T J + mu D^2 is formed there)
int mchol_device_solve_resident(int p, const double *jtj_dev, const double *diag_host, double mu, const double *rhs_host,
                                double *sol_host);
// ... and with more work enqueued BEHIND the back substitution on the solve's own stream, before the host is told: the
// matrix path's trial step (bd_host.hpp, round 5) evaluates the model at x + dx, its residual and the rows of dx^T J^T J dx
// there, and the solve's one host synchronisation brings all of it back -- `extra_n` doubles from `extra_dev` travel with
// the solution into `extra_host`.  *tail_valid = 0 when the natural-order factorisation was refused (the pivoted routine
// produced the solution after the tail had run on a discarded one): the caller repeats the tail's work its own way.
struct MCholTail
{
    void (*enqueue)(void *ctx, void *stream, const double *d_sol) = nullptr;
    // (round 5) the same, with the factor the solve leaves on the device: L (p x p row-major, lower triangle) and 1 / L_jj
    // -- the covariance at a fit's end is built from it (bd_host.hpp); only meaningful when *tail_valid comes back 1
    void (*enqueue_factor)(void *ctx, void *stream, const double *d_L, const double *d_dinv, int p) = nullptr;
    void *ctx = nullptr;
    const double *extra_dev = nullptr;
    int extra_n = 0;
    double *extra_host = nullptr;
};
int mchol_device_solve_resident_tail(int p, const double *jtj_dev, const double *diag_host, double mu, const double *rhs_host,
                                     double *sol_host, const MCholTail *tail, int *tail_valid);
// s = (J^T J) v with J^T J where it sits on the device (p doubles up, p doubles down): the row sums of the predicted
// reduction v^T J^T J v, each in the order of the host loop (j ascending, product and sum rounded separately)
int mchol_device_symv(int p, const double *jtj_dev, const double *v_host, double *s_host);
// the damped solve with those row sums for v = its own solution behind it, in one submission (round 5)
int mchol_device_solve_resident_symv(int p, const double *jtj_dev, const double *diag_host, double mu, const double *rhs_host,
                                     double *sol_host, double *rows_host, int *rows_valid);

struct LargeResult
{
    std::vector<double> x;
    int niter = 0, status = ST_CONTINUE, info = 0;
    double chisq0 = 0, chisq1 = 0, chisq_init = 0;
};

// trust_init + driver2 + trust_iterate of gsl_multilarge_nlinear
inline int large_solve(LargeOps &ops, const double *start, const int *ci, const double *cd, LargeResult &R,
                       double *ssrtrace, double *partrace)
{
    const int p = ops.p, n = ops.n;
    const int maxiter = ci[0], trs = ci[2], scale = ci[3];
    const double factor_up = cd[0], factor_down = cd[1], xtol = cd[5], gtol = cd[7];
    if (!(trs == 0 || trs == 5))
        return GSLNLS_E_UNSUPPORTED; // lmaccel / dogleg / ddogleg / subspace2D of multilarge are not lowered
    const bool need_jtj = (trs == 0);
    // beyond one wavefront's 64 parameters the factorisation runs on the device (mchol_device.hip); GSLNLS_LARGE_CHOL_DEVICE_MIN
    // moves the threshold, 0 = host always.  Operators that form J^T J on the device then keep it there: the damped solve
    // and the row sums of the predicted reduction read it in place, the host copy is fetched only if the device refuses.
    static const int dev_min = [] {
        const char *e = getenv("GSLNLS_LARGE_CHOL_DEVICE_MIN");
        return e ? atoi(e) : 65; // (round 5: was 400 -- measured against the round-3 kernels; see bd_host.hpp)
    }();
    const bool jtj_stays = need_jtj && dev_min > 0 && p >= dev_min && p <= 4096 && ops.can_keep_jtj_on_device() &&
                           !getenv("GSLNLS_LARGE_JTJ_HOST");
    ops.jtj_device_only = jtj_stays;
    struct ClearMode
    {
        LargeOps &o;
        ~ClearMode() { o.jtj_device_only = false; }
    } clear_mode{ops};
    bool host_jtj_valid = !jtj_stays;
    std::vector<double> x(start, start + p), g(p), dJ(p), JTJ(need_jtj ? (size_t)p * p : 0), diag(p, 1.0), dx(p, 0.0),
        xt(p), gt(p), dJt(p), JTJt(need_jtj ? (size_t)p * p : 0), vel(p, 0.0), z(p), r(p), d(p), wp(p), Bd(p);
    double fnorm2 = 0.0, bad = 0.0, delta, mu, nu = 2.0;

    auto do_scale = [&](bool init) {
        for (int j = 0; j < p; ++j)
        {
            if (scale == 1)
            {
                if (init)
                    diag[j] = 1.0;
                continue;
            }
            double norm = sqrt(dJ[j]);
            if (norm == 0.0)
                norm = 1.0;
            if (init || scale == 2)
                diag[j] = norm;
            else
                diag[j] = fmax(diag[j], norm);
        }
    };

    // GSLNLS_LARGE_PROF=1: wall time per phase of the driver (host clock; every phase ends in a stream synchronisation)
    static const bool prof = getenv("GSLNLS_LARGE_PROF") != nullptr;
    double t_phase[5] = {0, 0, 0, 0, 0}; // eval, eval_jac, step (cgst / lm), predicted reduction, total
    long n_phase[4] = {0, 0, 0, 0};
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto since = [](std::chrono::steady_clock::time_point t0) {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    };
    const auto t_begin = now();

    // trust_init
    int rc = ops.eval(x.data(), &fnorm2, g.data(), dJ.data(), need_jtj ? JTJ.data() : nullptr, &bad);
    if (rc)
        return rc;
    if (ops.lazy_jac && (rc = ops.eval_jac(g.data(), dJ.data(), need_jtj ? JTJ.data() : nullptr)))
        return rc;
    ops.accept();
    ops.nevalf += 1;
    ops.nevaldfu += 1;
    ops.nevaldf2 += 1;
    do_scale(true);
    {
        double Dx = 0.0, mx = -1.0;
        for (int j = 0; j < p; ++j)
        {
            Dx += (diag[j] * x[j]) * (diag[j] * x[j]);
            mx = fmax(mx, dJ[j] / (diag[j] * diag[j]));
        }
        delta = 0.3 * fmax(1.0, sqrt(Dx));
        mu = 1.0e-3 * mx;
    }
    R.chisq_init = R.chisq0 = R.chisq1 = fnorm2;
    if (ssrtrace)
        ssrtrace[0] = fnorm2;
    if (partrace)
        for (int k = 0; k < p; ++k)
            partrace[(size_t)(maxiter + 1) * k] = x[k];

    auto cgst_tau = [&](double dl) {
        const double norm_p = lg_nrm2(p, z.data()), norm_d = lg_nrm2(p, d.data());
        double u = 0.0;
        for (int i = 0; i < p; ++i)
            u += z[i] * d[i];
        const double t1 = u / (norm_d * norm_d);
        const double t2 = t1 * u + (dl + norm_p) * (dl - norm_p);
        return -t1 + sqrt(t2) / norm_d;
    };

    // GSL cgst.c cgst_step; every CG iteration is ONE fused pass (||J u||^2 and J^T J u together)
    auto cgst_step = [&]() -> int {
        const long cgmaxit = n; // cgst_alloc: max_iter == 0 -> n
        if (ops.device_cg)
        {
            int cgst = ST_SUCCESS;
            const int rcd = ops.cgst_device(g.data(), diag.data(), delta, cgmaxit, dx.data(), &cgst);
            if (rcd != GSLNLS_E_UNSUPPORTED)
                return rcd ? rcd : cgst;
        }
        for (int i = 0; i < p; ++i)
        {
            z[i] = 0.0;
            r[i] = d[i] = -g[i] / diag[i];
            wp[i] = g[i] / diag[i];
        }
        const double norm_g = lg_nrm2(p, wp.data());
        for (long it = 0; it < cgmaxit; ++it)
        {
            for (int i = 0; i < p; ++i)
                wp[i] = d[i] / diag[i];
            double nw2 = 0.0;
            int s = ops.jtjv(x.data(), wp.data(), &nw2, Bd.data());
            if (s)
                return s;
            ops.nevaldfu += 1; // NoTrans product
            const double norm_Jd = sqrt(nw2);
            if (norm_Jd == 0.0)
            {
                const double tau = cgst_tau(delta);
                for (int i = 0; i < p; ++i)
                    dx[i] = (z[i] + tau * d[i]) / diag[i];
                return ST_SUCCESS;
            }
            const double norm_r = lg_nrm2(p, r.data());
            double u = norm_r / norm_Jd;
            const double alpha = u * u;
            for (int i = 0; i < p; ++i)
                wp[i] = z[i] + alpha * d[i];
            u = lg_nrm2(p, wp.data());
            if (u >= delta)
            {
                const double tau = cgst_tau(delta);
                for (int i = 0; i < p; ++i)
                    dx[i] = (z[i] + tau * d[i]) / diag[i];
                return ST_SUCCESS;
            }
            z = wp;
            ops.nevaldfu += 1; // Trans product (already delivered by the fused pass)
            for (int i = 0; i < p; ++i)
                r[i] -= alpha * (Bd[i] / diag[i]);
            const double norm_rp1 = lg_nrm2(p, r.data());
            if (norm_rp1 / norm_g < 1.0e-6)
            {
                for (int i = 0; i < p; ++i)
                    dx[i] = z[i] / diag[i];
                return ST_SUCCESS;
            }
            u = norm_rp1 / norm_r;
            const double beta = u * u;
            for (int i = 0; i < p; ++i)
                d[i] = r[i] + beta * d[i];
        }
        for (int i = 0; i < p; ++i)
            dx[i] = z[i] / diag[i];
        return ST_EMAXITER;
    };

    std::vector<double> symv_rows; // rows of (J^T J) dx that came home with the lm step's solution, when symv_valid
    bool symv_valid = false;
    auto lm_step = [&]() -> int {
        symv_valid = false;
        std::vector<double> A, rhs(p);
        for (int i = 0; i < p; ++i)
            rhs[i] = -g[i];
        // A = J^T J + mu D^2 on the host (not needed when J^T J is taken where it sits on the device); a host copy that
        // cannot be fetched is an error of the fit, never a reason to go on with a stale or empty matrix
        auto damped = [&]() -> int {
            if (!host_jtj_valid)
            {
                if (ops.jtj_download(JTJ.data()) != GSLNLS_SUCCESS)
                    return GSLNLS_E_NODEVICE;
                host_jtj_valid = true;
            }
            A = JTJ;
            for (int i = 0; i < p; ++i)
                A[(size_t)i * p + i] += mu * diag[i] * diag[i];
            return GSLNLS_SUCCESS;
        };
        vel.assign(p, 0.0);
        int drc = GSLNLS_E_UNSUPPORTED; // (below the threshold, or a size the device routine does not take: p > 4096)
        if (dev_min > 0 && p >= dev_min)
        {
            if (const double *jd = ops.jtj_device())
            {
                // (with J^T J kept on the device the rows of the predicted reduction ride behind the solve)
                int rv = 0;
                symv_rows.resize(p);
                drc = jtj_stays ? mchol_device_solve_resident_symv(p, jd, diag.data(), mu, rhs.data(), vel.data(), symv_rows.data(), &rv)
                                : mchol_device_solve_resident(p, jd, diag.data(), mu, rhs.data(), vel.data());
                symv_valid = jtj_stays && drc == GSLNLS_SUCCESS && rv != 0;
            }
            else
            {
                if (const int e = damped())
                    return e;
                drc = mchol_device_solve(p, A.data(), rhs.data(), vel.data());
            }
        }
        if (drc == GSLNLS_E_UNSUPPORTED)
        {
            if (A.empty())
                if (const int e = damped())
                    return e;
            lg_mchol_solve(p, A, rhs, vel);
        }
        else if (drc != GSLNLS_SUCCESS)
            return drc; // a device failure is an error of the fit, not a reason to continue on the host
        dx = vel;
        return ST_SUCCESS;
    };

    int status = ST_CONTINUE, iter = 0, info = 0;
    do
    {
        if (g_interrupt_hook && g_interrupt_hook())
            return GSLNLS_E_INTERRUPTED;
        R.chisq0 = R.chisq1;
        // ---- trust_iterate ----
        int itstatus = ST_SUCCESS, bad_steps = 0;
        bool found = false;
        while (!found)
        {
            auto tp = now();
            int st = (trs == 5) ? cgst_step() : lm_step();
            t_phase[2] += since(tp);
            n_phase[2] += 1;
            if (st <= GSLNLS_E_NODEVICE)
                return st; // (the library's own error codes: the device failed under the step)
            double rho = -1.0, ssr_t = 0.0;
            if (st == ST_SUCCESS)
            {
                for (int i = 0; i < p; ++i)
                    xt[i] = x[i] + dx[i];
                tp = now();
                rc = ops.eval(xt.data(), &ssr_t, gt.data(), dJt.data(), need_jtj ? JTJt.data() : nullptr, &bad);
                t_phase[0] += since(tp);
                n_phase[0] += 1;
                if (rc)
                    return rc;
                ops.nevalf += 1;
                if (ssr_t < fnorm2)
                {
                    const double ared = 1.0 - ssr_t / fnorm2;
                    double pred;
                    if (trs == 5)
                    {
                        // quadratic model: -2 g.dx/||f||^2 - (||J dx||/||f||)^2, one more product with J
                        double nJdx2 = 0.0, gTdx = 0.0;
                        tp = now();
                        if (!ops.cached_njdx2(&nJdx2) && (rc = ops.jtjv(x.data(), dx.data(), &nJdx2, Bd.data())))
                            return rc;
                        t_phase[3] += since(tp);
                        n_phase[3] += 1;
                        ops.nevaldfu += 1;
                        for (int i = 0; i < p; ++i)
                            gTdx += g[i] * dx[i];
                        pred = -nJdx2 / fnorm2 - 2.0 * gTdx / fnorm2;
                    }
                    else
                    {
                        double vJv = 0.0, Dv2 = 0.0;
                        // the row sums s_i = sum_j (J^T J)_ij v_j where J^T J sits (same products, same order, rounded the same
                        // way as the loop below), or on the host
                        bool rows_done = false;
                        if (jtj_stays)
                        {
                            if (symv_valid)
                            {
                                wp = symv_rows; // (they came home with the step's solution)
                                rows_done = true;
                            }
                            else if (const double *jd = ops.jtj_device())
                                rows_done = mchol_device_symv(p, jd, vel.data(), wp.data()) == GSLNLS_SUCCESS;
                            if (!rows_done && !host_jtj_valid)
                            {
                                if (ops.jtj_download(JTJ.data()) != GSLNLS_SUCCESS)
                                    return GSLNLS_E_NODEVICE;
                                host_jtj_valid = true;
                            }
                        }
                        for (int i = 0; i < p; ++i)
                        {
                            double s = 0.0;
                            if (rows_done)
                                s = wp[i];
                            else
                                for (int j = 0; j < p; ++j)
                                    s += JTJ[i * p + j] * vel[j];
                            vJv += s * vel[i];
                            Dv2 += (diag[i] * vel[i]) * (diag[i] * vel[i]);
                        }
                        pred = vJv / fnorm2 + 2.0 * mu * Dv2 / fnorm2;
                    }
                    rho = pred > 0.0 ? ared / pred : -1.0;
                }
                if (rho > 0.0)
                    found = true;
            }
            else if (st == ST_EBADFUNC)
                return st;
            lmd_radius(rho, factor_up, factor_down, delta);
            if (found)
            {
                // accepted: g, J^T J (diag) at x_trial came with the same EVAL pass (or are completed now)
                tp = now();
                if (ops.lazy_jac && (rc = ops.eval_jac(gt.data(), dJt.data(), need_jtj ? JTJt.data() : nullptr)))
                    return rc;
                t_phase[1] += since(tp);
                n_phase[1] += 1;
                ops.accept();
                ops.nevaldfu += 1;
                ops.nevaldf2 += 1;
                x = xt;
                g = gt;
                dJ = dJt;
                if (need_jtj)
                    JTJ.swap(JTJt); // (the trial buffer is rewritten by the next evaluation: no 8 p^2-byte copy per iteration)
                host_jtj_valid = !jtj_stays;
                fnorm2 = ssr_t;
                do_scale(false);
                lmd_nielsen_accept(rho, mu, nu);
                bad_steps = 0;
            }
            else
            {
                lmd_nielsen_reject(mu, nu);
                if (++bad_steps > LMD_MAX_REJECTS)
                {
                    itstatus = ST_ENOPROG;
                    break;
                }
            }
        }
        R.niter += 1;
        R.chisq1 = fnorm2;
        if (itstatus == ST_EBADFUNC || (itstatus == ST_ENOPROG && iter == 0))
        {
            info = itstatus;
            status = itstatus;
            break;
        }
        ++iter;
        if (ssrtrace)
            ssrtrace[iter] = fnorm2;
        if (partrace)
            for (int k = 0; k < p; ++k)
                partrace[iter + (size_t)(maxiter + 1) * k] = x[k];
        if (g_trace_on)
        {
            // callback_large (src/nls_large.c:715-739): |x|^2 and cond(J) = 1 / gsl_multilarge_nlinear_rcond.  GSL's
            // Steihaug-Toint solver reports rcond = 0 (cgst.c: not implemented) -> inf; its lm solver the square root of
            // the 1-norm reciprocal condition of J^T J from the Cholesky factor (cholesky.c: gsl_linalg_cholesky_rcond, an
            // estimator) -- here the 1-norm condition itself, from the explicit inverse (trace runs only)
            double xsq = 0.0, cond = INFINITY;
            for (int k = 0; k < p; ++k)
                xsq += x[k] * x[k];
            if (need_jtj)
            {
                if (!host_jtj_valid)
                {
                    if (ops.jtj_download(JTJ.data()) != GSLNLS_SUCCESS)
                        return GSLNLS_E_NODEVICE;
                    host_jtj_valid = true;
                }
                std::vector<double> Ai(JTJ);
                double n1 = 0.0, n1i = 0.0;
                for (int j = 0; j < p; ++j)
                {
                    double c = 0.0;
                    for (int i = 0; i < p; ++i)
                        c += fabs(JTJ[(size_t)i * p + j]);
                    n1 = fmax(n1, c);
                }
                if (lg_chol(p, Ai))
                {
                    lg_chol_invert(p, Ai);
                    for (int j = 0; j < p; ++j)
                    {
                        double c = 0.0;
                        for (int i = 0; i < p; ++i)
                            c += fabs(Ai[(size_t)i * p + j]);
                        n1i = fmax(n1i, c);
                    }
                    cond = sqrt(n1 * n1i);
                }
            }
            trace_printf("iter %3d: ssr = %g, |x|^2 = %g, cond(J) = %g\n", iter, fnorm2, xsq, cond);
        }
        // gsl_multilarge_nlinear_test
        bool ok = true;
        for (int i = 0; i < p && ok; ++i)
            if (!(fabs(dx[i]) < xtol * xtol + xtol * fabs(x[i])))
                ok = false;
        if (ok)
        {
            info = 1;
            status = ST_SUCCESS;
        }
        else
        {
            double gnorm = 0.0;
            for (int i = 0; i < p; ++i)
                gnorm = fmax(gnorm, fabs(fmax(x[i], 1.0) * g[i]));
            if (gnorm <= gtol * fmax(0.5 * fnorm2, 1.0))
            {
                info = 2;
                status = ST_SUCCESS;
            }
            else
                status = ST_CONTINUE;
        }
    } while (status == ST_CONTINUE && iter < maxiter);
    if (iter >= maxiter && status != ST_SUCCESS && status != ST_EBADFUNC && status != ST_ENOPROG)
        status = ST_EMAXITER;
    if (prof)
    {
        t_phase[4] = since(t_begin);
        fprintf(stderr,
                "[large prof] total %.2f ms | f at trial points %.2f (%ld) | Jacobian at accepted points %.2f (%ld) | steps %.2f "
                "(%ld) | predicted reduction %.2f (%ld) | host algebra and the rest %.2f\n",
                t_phase[4], t_phase[0], n_phase[0], t_phase[1], n_phase[1], t_phase[2], n_phase[2], t_phase[3], n_phase[3],
                t_phase[4] - t_phase[0] - t_phase[1] - t_phase[2] - t_phase[3]);
    }
    R.x = x;
    R.status = status;
    R.info = info;
    return 0;
}

// ------------------------------------------------------------------------------------------------
template <class M>
struct RowLargeOps : LargeOps
{
    static constexpr int P = M::P;
    static constexpr int NV = PassSums<P>::NV;
    static constexpr int T = 256;
    DenseFit<M> &fit;
    int G;
    double *d_part = nullptr, *d_tot = nullptr, *d_x = nullptr, *d_u = nullptr;
    double h_tot[NV];
    explicit RowLargeOps(DenseFit<M> &f) : fit(f)
    {
        n = f.n;
        p = P;
        G = (int)std::min<long long>(1024, ((long long)n + T - 1) / T);
        hipMalloc(&d_part, sizeof(double) * NV * G);
        hipMalloc(&d_tot, sizeof(double) * NV);
        hipMalloc(&d_x, sizeof(double) * P);
        hipMalloc(&d_u, sizeof(double) * P);
    }
    ~RowLargeOps() override
    {
        hipFree(d_part);
        hipFree(d_tot);
        hipFree(d_x);
        hipFree(d_u);
    }
    int pass(int mode, const double *x, const double *u)
    {
        hipStream_t st = fit.stream;
        GSLNLS_HIP_OK(hipMemcpyAsync(d_x, x, sizeof(double) * P, hipMemcpyHostToDevice, st));
        if (u)
            GSLNLS_HIP_OK(hipMemcpyAsync(d_u, u, sizeof(double) * P, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL((large_row_kernel<M, T>), dim3(G), dim3(T), 0, st, fit.ctx, mode, d_x, u ? d_u : nullptr,
                           d_part);
        hipLaunchKernelGGL(large_reduce_kernel, dim3(NV), dim3(64), 0, st, d_part, NV, G, d_tot);
        GSLNLS_HIP_OK(hipMemcpyAsync(h_tot, d_tot, sizeof(double) * NV, hipMemcpyDeviceToHost, st));
        GSLNLS_HIP_OK(hipStreamSynchronize(st));
        ++npass;
        return 0;
    }
    int eval(const double *x, double *ssr, double *g, double *diag, double *jtj, double *bad) override
    {
        int rc = pass(LG_EVAL, x, nullptr);
        if (rc)
            return rc;
        const PassSums<P> *s = reinterpret_cast<const PassSums<P> *>(h_tot);
        *ssr = s->ssr;
        *bad = s->badj;
        for (int i = 0; i < P; ++i)
        {
            g[i] = s->g[i];
            diag[i] = s->A[tri(i, i)];
            if (jtj)
                for (int j = 0; j <= i; ++j)
                    jtj[i * P + j] = jtj[j * P + i] = s->A[tri(i, j)];
        }
        return 0;
    }
    void accept() override {}
    int jtjv(const double *xcur, const double *u, double *normw2, double *out) override
    {
        int rc = pass(LG_JTJV, xcur, u);
        if (rc)
            return rc;
        const PassSums<P> *s = reinterpret_cast<const PassSums<P> *>(h_tot);
        *normw2 = s->ssr;
        for (int i = 0; i < P; ++i)
            out[i] = s->g[i];
        return 0;
    }
    int full_jtj(const double *xcur, double *jtj) override
    {
        double ssr, bad, g[P], dg[P];
        return eval(xcur, &ssr, g, dg, jtj, &bad);
    }
    int residual(const double *xcur, double *resid_host) override
    {
        // weighted residual through the dense finalize kernel at xcur
        LmState<P> s;
        lm_state_reset<P>(s, xcur, nullptr);
        GSLNLS_HIP_OK(hipMemcpy(fit.ctx.state[0], &s, sizeof(s), hipMemcpyHostToDevice));
        if (!fit.d_resid)
            GSLNLS_HIP_OK(hipMalloc(&fit.d_resid, sizeof(double) * (size_t)n));
        fit.ctx.prm.h_df = 1e-8;
        fit.launch_finalize(JAC_ANALYTIC, 0, fit.d_resid, nullptr, nullptr);
        GSLNLS_HIP_OK(hipMemcpyAsync(resid_host, fit.d_resid, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost,
                                     fit.stream));
        GSLNLS_HIP_OK(hipStreamSynchronize(fit.stream));
        return 0;
    }
};

// dense GLM family with A resident in HBM
template <class M>
LargeOps *DenseFit<M>::make_large_ops()
{
    return new RowLargeOps<M>(*this);
}

template <int P>
struct GlmLargeOps : LargeOps
{
    static constexpr int T = 256;
    const double *d_A = nullptr, *d_y = nullptr, *d_sw = nullptr;
    bool owns = false;
    double *d_m[2] = {nullptr, nullptr}, *d_f[2] = {nullptr, nullptr}; // [cur, trial]
    int cur = 0;
    double *d_part = nullptr, *d_tot = nullptr, *d_x = nullptr, *d_u = nullptr, *d_jtjpart = nullptr;
    std::vector<double> h_tot;
    int G;
    hipStream_t st = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;

    int init(const double *A, const double *y, const double *sw, int n_, bool on_device)
    {
        n = n_;
        p = P;
        GSLNLS_HIP_OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        GSLNLS_HIP_OK(hipEventCreate(&e0));
        GSLNLS_HIP_OK(hipEventCreate(&e1));
        const size_t nb = sizeof(double) * (size_t)n;
        if (on_device)
        {
            d_A = A;
            d_y = y;
            d_sw = sw;
        }
        else
        {
            owns = true;
            double *a = nullptr, *yy = nullptr, *ww = nullptr;
            GSLNLS_HIP_OK(hipMalloc(&a, nb * P));
            GSLNLS_HIP_OK(hipMemcpy(a, A, nb * P, hipMemcpyHostToDevice));
            GSLNLS_HIP_OK(hipMalloc(&yy, nb));
            GSLNLS_HIP_OK(hipMemcpy(yy, y, nb, hipMemcpyHostToDevice));
            if (sw)
            {
                GSLNLS_HIP_OK(hipMalloc(&ww, nb));
                GSLNLS_HIP_OK(hipMemcpy(ww, sw, nb, hipMemcpyHostToDevice));
            }
            d_A = a;
            d_y = yy;
            d_sw = ww;
        }
        for (int k = 0; k < 2; ++k)
        {
            GSLNLS_HIP_OK(hipMalloc(&d_m[k], nb));
            GSLNLS_HIP_OK(hipMalloc(&d_f[k], nb));
        }
        G = 1024;
        h_tot.resize(2 * P + 2);
        GSLNLS_HIP_OK(hipMalloc(&d_part, sizeof(double) * (2 * P + 2) * G));
        GSLNLS_HIP_OK(hipMalloc(&d_tot, sizeof(double) * (2 * P + 2)));
        GSLNLS_HIP_OK(hipMalloc(&d_x, sizeof(double) * P));
        GSLNLS_HIP_OK(hipMalloc(&d_u, sizeof(double) * P));
        return 0;
    }
    ~GlmLargeOps() override
    {
        if (owns)
        {
            hipFree(const_cast<double *>(d_A));
            hipFree(const_cast<double *>(d_y));
            hipFree(const_cast<double *>(d_sw));
        }
        for (int k = 0; k < 2; ++k)
        {
            hipFree(d_m[k]);
            hipFree(d_f[k]);
        }
        hipFree(d_part);
        hipFree(d_tot);
        hipFree(d_x);
        hipFree(d_u);
        hipFree(d_jtjpart);
        if (e0)
            hipEventDestroy(e0);
        if (e1)
            hipEventDestroy(e1);
        if (st)
            hipStreamDestroy(st);
    }
    int pass(int mode, const double *x, const double *u)
    {
        GlmArgs a;
        a.A = d_A;
        a.y = d_y;
        a.sw = d_sw;
        a.n = n;
        a.partials = d_part;
        a.mode = mode;
        if (mode == LG_EVAL)
        {
            GSLNLS_HIP_OK(hipMemcpyAsync(d_x, x, sizeof(double) * P, hipMemcpyHostToDevice, st));
            a.m = d_m[cur ^ 1];
            a.f = d_f[cur ^ 1];
        }
        else
        {
            GSLNLS_HIP_OK(hipMemcpyAsync(d_u, u, sizeof(double) * P, hipMemcpyHostToDevice, st));
            a.m = d_m[cur];
            a.f = d_f[cur];
        }
        a.xpt = d_x;
        a.u = d_u;
        hipEventRecord(e0, st);
        hipLaunchKernelGGL((glm_pass_kernel<P, T>), dim3(G), dim3(T), 0, st, a);
        hipEventRecord(e1, st);
        hipLaunchKernelGGL(large_reduce_kernel, dim3(2 * P + 2), dim3(64), 0, st, d_part, 2 * P + 2, G, d_tot);
        GSLNLS_HIP_OK(hipMemcpyAsync(h_tot.data(), d_tot, sizeof(double) * (2 * P + 2), hipMemcpyDeviceToHost, st));
        GSLNLS_HIP_OK(hipStreamSynchronize(st));
        hipEventElapsedTime(&pass_ms, e0, e1);
        ++npass;
        return 0;
    }
    int eval(const double *x, double *ssr, double *g, double *diag, double *jtj, double *bad) override
    {
        int rc = pass(LG_EVAL, x, nullptr);
        if (rc)
            return rc;
        *ssr = h_tot[0];
        *bad = 0.0;
        for (int i = 0; i < P; ++i)
        {
            g[i] = h_tot[2 + i];
            diag[i] = h_tot[2 + P + i];
        }
        if (jtj)
        {
            // LM on the normal equations needs the whole J^T J at the trial point
            const int was = cur;
            cur ^= 1; // trial buffers hold m at x
            rc = full_jtj(x, jtj);
            cur = was;
        }
        return rc;
    }
    void accept() override { cur ^= 1; }
    int jtjv(const double *, const double *u, double *normw2, double *out) override
    {
        int rc = pass(LG_JTJV, nullptr, u);
        if (rc)
            return rc;
        *normw2 = h_tot[0];
        for (int i = 0; i < P; ++i)
            out[i] = h_tot[2 + i];
        return 0;
    }
    int full_jtj(const double *, double *jtj) override
    {
        constexpr int Gj = 1024; // four 256-thread workgroups per CU
        if (!d_jtjpart)
            GSLNLS_HIP_OK(hipMalloc(&d_jtjpart, sizeof(double) * (size_t)P * P * Gj + sizeof(double) * P * P));
        double *d_out = d_jtjpart + (size_t)P * P * Gj;
        // matrix cores for every p the family is built for (16, 32, 48, 64)
        hipLaunchKernelGGL((glm_jtj_mfma_kernel<P, 256>), dim3(Gj), dim3(256), 0, st, d_A, d_m[cur], (long long)n, d_jtjpart);
        hipLaunchKernelGGL(large_reduce_kernel, dim3(P * P), dim3(64), 0, st, d_jtjpart, P * P, Gj, d_out);
        GSLNLS_HIP_OK(hipMemcpyAsync(jtj, d_out, sizeof(double) * P * P, hipMemcpyDeviceToHost, st));
        GSLNLS_HIP_OK(hipStreamSynchronize(st));
        return 0;
    }
    int residual(const double *, double *resid_host) override
    {
        GSLNLS_HIP_OK(hipMemcpy(resid_host, d_f[cur], sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
        return 0;
    }
};

} // namespace gslnls
