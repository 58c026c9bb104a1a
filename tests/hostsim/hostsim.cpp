// hostsim.cpp -- TEST-ONLY harness (never linked into libgslnls_hip.so).
//
// Compiles the SAME headers the device kernels use (lm_core.hpp, rowops.hpp, models.hpp)
// with g++ and replaces the GPU's parallel pass by a serial loop, so the device state
// machine and row functions can be checked against the oracle on a CPU-only box.  This
// exercises "host logic" only; GPU parity proper is tests/test_gpu_*.py through the C ABI.
#include <cstdlib>
#include <cstdio>
#include <math.h>
#include <string.h>
#include "lm_core.hpp"
#include "models.hpp"
#include "rowops.hpp"
#include "batch_core.hpp"
#include "mstart_driver.hpp"
#include "irls_core.hpp"
#include "expr_compile.hpp"

using namespace gslnls;

template <class M, int JAC>
static void pass(const LmState<M::P> &s, const LmParams &prm, int n, const double *x, const double *y,
                 const double *sw, PassSums<M::P> &acc)
{
    constexpr int P = M::P;
    double th[P], delta[P];
    for (int k = 0; k < P; ++k)
        th[k] = (s.phase == PH_FVV) ? s.x[k] : s.xt[k];
    fd_deltas<P>(th, prm.h_df, delta);
    pass_zero<P>(acc);
    for (int i = 0; i < n; ++i)
    {
        double xr[M::NX];
        for (int c = 0; c < M::NX; ++c)
            xr[c] = x[i + (size_t)n * c];
        double Jrow[P];
        const double w = sw ? sw[i] : 1.0;
        if (s.phase == PH_FVV)
        {
            const double fv = row_fvv<M, JAC>(th, s.vel, delta, prm.h_fvv, prm.fvv_analytic != 0, xr, y[i], w, Jrow,
                                              &acc.badj);
            for (int k = 0; k < P; ++k)
                acc.g[k] += Jrow[k] * fv;
        }
        else
        {
            const double f = row_fj<M, JAC>(th, delta, xr, y[i], w, Jrow, &acc.badj);
            acc_fj<P>(acc, f, Jrow);
        }
    }
}

template <class M>
static int fit(int n, const double *x, const double *y, const double *sw, const double *start, const double *lupars,
               const int *ci, const double *cd, int jac, int fvv, double *par, int *ints, double *dbls, double *covar,
               double *ssrtrace, double *partrace)
{
    constexpr int P = M::P;
    LmParams prm;
    prm.maxiter = ci[0];
    prm.trs = (ci[2] == 1) ? 1 : 0;
    prm.scale = ci[3];
    prm.fdtype = ci[5] ? 1 : 0;
    prm.jac_analytic = jac;
    prm.fvv_analytic = fvv;
    prm.has_bounds = lupars != nullptr;
    prm.has_weights = sw != nullptr;
    prm.bench_hold = 0;
    prm.chisq_in = NAN;
    prm.factor_up = cd[0];
    prm.factor_down = cd[1];
    prm.avmax = cd[2];
    prm.h_df = cd[3];
    prm.h_fvv = cd[4];
    prm.xtol = cd[5];
    prm.ftol = cd[6];
    prm.gtol = cd[7];
    LmState<P> s;
    lm_state_reset<P>(s, start, lupars);
    const int jacmode = jac ? JAC_ANALYTIC : (ci[5] ? JAC_CENTER : JAC_FORWARD);
    long launches = 0;
    while (s.phase != PH_DONE && launches < 1000000)
    {
        PassSums<P> acc;
        if (jacmode == JAC_ANALYTIC)
            pass<M, JAC_ANALYTIC>(s, prm, n, x, y, sw, acc);
        else if (jacmode == JAC_FORWARD)
            pass<M, JAC_FORWARD>(s, prm, n, x, y, sw, acc);
        else
            pass<M, JAC_CENTER>(s, prm, n, x, y, sw, acc);
        const int nb = s.niter, pb = s.phase;
        lm_advance<P>(s, acc, prm);
        if (getenv("GSLNLS_HOSTSIM_TRACE"))
            fprintf(stderr, "pass %ld: phase %d -> %d niter %d -> %d mu %.3e bad_steps %d\n", launches, pb, s.phase, nb,
                    s.niter, s.mu, s.bad_steps);
        if (ssrtrace)
        {
            if (pb == PH_INIT)
            {
                ssrtrace[0] = s.chisq_init;
                for (int k = 0; k < P; ++k)
                    partrace[(size_t)(prm.maxiter + 1) * k] = s.x[k];
            }
            else if (s.niter != nb && s.status != ST_EBADFUNC && !(s.status == ST_ENOPROG && nb == 0))
            {
                ssrtrace[s.niter] = s.chisq1;
                for (int k = 0; k < P; ++k)
                    partrace[s.niter + (size_t)(prm.maxiter + 1) * k] = s.x[k];
            }
        }
        ++launches;
    }
    for (int k = 0; k < P; ++k)
        par[k] = s.x[k];
    ints[0] = s.niter;
    ints[1] = s.status;
    ints[2] = s.info;
    ints[3] = s.nevalf;
    ints[4] = s.nevaldf;
    ints[5] = s.nevalfvv;
    ints[6] = (int)launches;
    dbls[0] = s.chisq1;
    dbls[1] = s.chisq0 - s.chisq1;
    dbls[2] = s.chisq_init;
    dbls[3] = s.mu;
    dbls[4] = s.delta;
    dbls[5] = det_cholesky<P>(s.A);
    if (covar)
        covar_from_jtj<P>(s.A, covar);
    return s.status;
}

extern "C" int hostsim_fit(int model, int n, const double *x, const double *y, const double *sw, const double *start,
                           const double *lupars, const int *ci, const double *cd, int jac, int fvv, double *par,
                           int *ints, double *dbls, double *covar, double *ssrtrace, double *partrace)
{
    switch (model)
    {
    case 1:
        return fit<ModelExpDecay>(n, x, y, sw, start, lupars, ci, cd, jac, fvv, par, ints, dbls, covar, ssrtrace, partrace);
    case 2:
        return fit<ModelMisra1a>(n, x, y, sw, start, lupars, ci, cd, jac, fvv, par, ints, dbls, covar, ssrtrace, partrace);
    case 3:
        return fit<ModelGaussPeak>(n, x, y, sw, start, lupars, ci, cd, jac, fvv, par, ints, dbls, covar, ssrtrace, partrace);
    case 4:
        return fit<ModelGauss1>(n, x, y, sw, start, lupars, ci, cd, jac, fvv, par, ints, dbls, covar, ssrtrace, partrace);
    default:
        return -101;
    }
}

// direct access to the device linear algebra for unit tests
extern "C" void hostsim_lm_solve3(const double *Ap, const double *diag, double mu, const double *rhs, double *sol)
{
    lm_solve<3>(Ap, diag, mu, rhs, sol);
}
extern "C" void hostsim_lm_solve8(const double *Ap, const double *diag, double mu, const double *rhs, double *sol)
{
    lm_solve<8>(Ap, diag, mu, rhs, sol);
}

// ---------------------------------------------------------------------------------------------
// multi-start: the product's host driver (mstart_driver.hpp) around a CPU evaluator that runs the
// device per-point routine (batch_core.hpp) serially.  Test-only.
template <int NX>
struct RowsHost
{
    static constexpr int STATIC_N = 0, LPF = 1;
    const double *x, *y, *sw;
    int n;
    void operator()(int i, double *xr, double &yy, double &w) const
    {
        for (int c = 0; c < NX; ++c)
            xr[c] = x[i + (size_t)n * c];
        yy = y[i];
        w = sw ? sw[i] : 1.0;
    }
};

template <class M>
struct CpuEvaluator : MsEvaluator
{
    static constexpr int P = M::P;
    RowsHost<M::NX> rows;
    LmParams prm;
    int jacmode;
    const double *lupars;
    SobolTable tab;
    int run(MsBatch &b, int lo, int hi, double *out, bool) override
    {
        MsParams mp;
        mp.prm = prm;
        mp.prm.maxiter = b.maxiter;
        mp.prm.gtol = 1e-3;
        mp.dtol = b.dtol;
        mp.n = rows.n;
        mp.always_fit = b.always_fit;
        for (int idx = lo; idx < hi; ++idx)
        {
            double st[P];
            for (int k = 0; k < P; ++k)
                st[k] = b.draw[idx] >= 0 ? sobol_to_range(sobol_coord(tab, (unsigned int)b.draw[idx], k), b.range[2 * k],
                                                           b.range[2 * k + 1], b.kd[k])
                                         : b.start[(size_t)idx * P + k];
            MsRecord<P> rec;
            if (jacmode == JAC_ANALYTIC)
                ms_fit_point<M, JAC_ANALYTIC>(mp, rows, st, lupars, rec);
            else if (jacmode == JAC_FORWARD)
                ms_fit_point<M, JAC_FORWARD>(mp, rows, st, lupars, rec);
            else
                ms_fit_point<M, JAC_CENTER>(mp, rows, st, lupars, rec);
            memcpy(out + (size_t)(idx - lo) * MsRecord<P>::K, &rec, sizeof(rec));
        }
        return 0;
    }
    int fetch(const double *src, bool, double *dst, size_t nd) override
    {
        memcpy(dst, src, sizeof(double) * nd);
        return 0;
    }
};

template <class M>
static int mstart(int n, const double *x, const double *y, const double *sw, const double *start2p,
                  const double *lupars, const int *ci, const double *cd, const int *has_start, int jac, int fvv,
                  const MsComm &comm, double *mpopt, int *ints, double *dbls)
{
    constexpr int P = M::P;
    CpuEvaluator<M> ev;
    ev.rows = RowsHost<M::NX>{x, y, sw, n};
    ev.prm.maxiter = ci[0];
    ev.prm.trs = (ci[2] == 1) ? 1 : 0;
    ev.prm.scale = ci[3];
    ev.prm.fdtype = ci[5] ? 1 : 0;
    ev.prm.jac_analytic = jac;
    ev.prm.fvv_analytic = fvv;
    ev.prm.has_bounds = lupars != nullptr;
    ev.prm.has_weights = sw != nullptr;
    ev.prm.bench_hold = 0;
    ev.prm.chisq_in = NAN;
    ev.prm.chisq_in = NAN;
    ev.prm.factor_up = cd[0];
    ev.prm.factor_down = cd[1];
    ev.prm.avmax = cd[2];
    ev.prm.h_df = cd[3];
    ev.prm.h_fvv = cd[4];
    ev.prm.xtol = cd[5];
    ev.prm.ftol = cd[6];
    ev.prm.gtol = cd[7];
    ev.jacmode = jac ? JAC_ANALYTIC : (ci[5] ? JAC_CENTER : JAC_FORWARD);
    ev.lupars = lupars;
    sobol_build(ev.tab, P);
    MsState m;
    ms_init(m, P, ci, cd, start2p, has_start, lupars);
    int rc = ms_major_loop(m, ev, comm, start2p);
    if (rc)
        return rc;
    if (m.mssropt[1] < m.mssropt[0])
    {
        m.mssropt[0] = m.mssropt[1];
        m.ssrconv[0] = m.ssrconv[1];
        m.mpopt = m.mpopt1;
    }
    if (m.mssropt[0] < cd[6] || m.ssrconv[0] < cd[6])
        m.mpopt[0] = lupars ? fmin(m.mpopt[0] + 1.0e-4, lupars[1]) : m.mpopt[0] + 1.0e-4;
    for (int k = 0; k < P; ++k)
        mpopt[k] = m.mpopt[k];
    ints[0] = m.nsp;
    ints[1] = m.nwsp;
    ints[2] = m.mstarts;
    ints[3] = m.mstop;
    ints[4] = (int)m.total_fits;
    ints[5] = (int)m.next_draw;
    dbls[0] = m.mssropt[0];
    dbls[1] = m.ssrconv[0];
    return 0;
}

typedef int (*allgather_fn)(void *, int, int);

extern "C" int hostsim_mstart(int model, int n, const double *x, const double *y, const double *sw,
                              const double *start2p, const double *lupars, const int *ci, const double *cd,
                              const int *has_start, int jac, int fvv, int rank, int world, allgather_fn fn,
                              double *shard_buf, double *all_buf, long long cap_points, double *mpopt, int *ints,
                              double *dbls)
{
    MsComm comm;
    comm.rank = rank;
    comm.world = world;
    comm.allgather = fn;
    comm.shard_buf = shard_buf;
    comm.all_buf = all_buf;
    comm.cap_points = cap_points;
    comm.buffers_on_device = 0;
    switch (model)
    {
    case 1:
        return mstart<ModelExpDecay>(n, x, y, sw, start2p, lupars, ci, cd, has_start, jac, fvv, comm, mpopt, ints, dbls);
    case 2:
        return mstart<ModelMisra1a>(n, x, y, sw, start2p, lupars, ci, cd, has_start, jac, fvv, comm, mpopt, ints, dbls);
    case 3:
        return mstart<ModelGaussPeak>(n, x, y, sw, start2p, lupars, ci, cd, has_start, jac, fvv, comm, mpopt, ints, dbls);
    default:
        return -101;
    }
}

extern "C" void hostsim_sobol(int dim, int first, int count, double *out)
{
    SobolTable t;
    sobol_build(t, dim);
    for (int i = 0; i < count; ++i)
        for (int d = 0; d < dim; ++d)
            out[(size_t)i * dim + d] = sobol_coord(t, (unsigned int)(first + i), d);
}

extern "C" int hostsim_mstart_batch(int n, const double *x, const double *y, const double *sw, const double *ranges,
                                    const double *kd, long long first_draw, int count, int maxiter, double dtol,
                                    const int *ci, const double *cd, int jac, double *records)
{
    using M = ModelMisra1a;
    constexpr int P = M::P;
    CpuEvaluator<M> ev;
    ev.rows = RowsHost<M::NX>{x, y, sw, n};
    ev.prm.maxiter = ci[0];
    ev.prm.trs = 0;
    ev.prm.scale = ci[3];
    ev.prm.fdtype = ci[5] ? 1 : 0;
    ev.prm.jac_analytic = jac;
    ev.prm.fvv_analytic = 0;
    ev.prm.has_bounds = 0;
    ev.prm.has_weights = sw != nullptr;
    ev.prm.bench_hold = 0;
    ev.prm.chisq_in = NAN;
    ev.prm.chisq_in = NAN;
    ev.prm.factor_up = cd[0];
    ev.prm.factor_down = cd[1];
    ev.prm.avmax = cd[2];
    ev.prm.h_df = cd[3];
    ev.prm.h_fvv = cd[4];
    ev.prm.xtol = cd[5];
    ev.prm.ftol = cd[6];
    ev.prm.gtol = cd[7];
    ev.jacmode = jac ? JAC_ANALYTIC : (ci[5] ? JAC_CENTER : JAC_FORWARD);
    ev.lupars = nullptr;
    sobol_build(ev.tab, P);
    MsBatch b;
    b.count = count;
    b.p = P;
    b.K = MsRecord<P>::K;
    b.draw.resize(count);
    for (int i = 0; i < count; ++i)
        b.draw[i] = first_draw + i;
    b.start.assign((size_t)count * P, 0.0);
    b.range.assign(ranges, ranges + 2 * P);
    b.kd.assign(kd, kd + P);
    b.maxiter = maxiter;
    b.dtol = dtol;
    b.always_fit = 0;
    return ev.run(b, 0, count, records, false);
}

// one concentration batch through ms_run_batch with a callback communicator, records wanted or not (host_records):
// the multi-rank branch of gslnls_mstart_batch(lo < 0, records = NULL) without a device
extern "C" int hostsim_run_batch_comm(int n, const double *x, const double *y, int count, int rank, int world,
                                      allgather_fn fn, double *shard_buf, double *all_buf, long long cap_points,
                                      int want_records, const int *ci, const double *cd, double *records_out)
{
    using M = ModelMisra1a;
    constexpr int P = M::P;
    CpuEvaluator<M> ev;
    ev.rows = RowsHost<M::NX>{x, y, nullptr, n};
    ev.prm.maxiter = ci[0];
    ev.prm.trs = 0;
    ev.prm.scale = ci[3];
    ev.prm.fdtype = 0;
    ev.prm.jac_analytic = 1;
    ev.prm.fvv_analytic = 0;
    ev.prm.has_bounds = 0;
    ev.prm.has_weights = 0;
    ev.prm.bench_hold = 0;
    ev.prm.chisq_in = NAN;
    ev.prm.factor_up = cd[0];
    ev.prm.factor_down = cd[1];
    ev.prm.avmax = cd[2];
    ev.prm.h_df = cd[3];
    ev.prm.h_fvv = cd[4];
    ev.prm.xtol = cd[5];
    ev.prm.ftol = cd[6];
    ev.prm.gtol = cd[7];
    ev.jacmode = JAC_ANALYTIC;
    ev.lupars = nullptr;
    sobol_build(ev.tab, P);
    MsComm comm;
    comm.rank = rank;
    comm.world = world;
    comm.allgather = fn;
    comm.shard_buf = shard_buf;
    comm.all_buf = all_buf;
    comm.cap_points = cap_points;
    comm.buffers_on_device = 0;
    MsBatch b;
    b.count = count;
    b.p = P;
    b.K = MsRecord<P>::K;
    b.draw.resize(count);
    for (int i = 0; i < count; ++i)
        b.draw[i] = i;
    b.start.assign((size_t)count * P, 0.0);
    const double rg[4] = {1.0, 500.0, 0.01, 5.0};
    b.range.assign(rg, rg + 4);
    b.kd.assign(P, 0.75);
    b.maxiter = 5;
    b.dtol = 1e-6;
    b.always_fit = 0;
    b.host_records = want_records != 0;
    const int rc = ms_run_batch(ev, comm, b);
    if (rc == 0 && want_records && records_out)
        memcpy(records_out, b.rec, sizeof(double) * (size_t)count * b.K);
    return rc;
}

extern "C" void hostsim_psi(int rho, const double *cc, int n, const double *x, double *psi, double *psip)
{
    LossCfg L;
    L.rho = rho;
    for (int k = 0; k < 3; ++k)
        L.cc[k] = cc[k];
    for (int i = 0; i < n; ++i)
    {
        psi[i] = irls_psi(x[i], L);
        psip[i] = irls_psip(x[i], L);
    }
}

// expression models: compile (expr_compile.hpp) and interpret (vm_program.hpp) on the host
extern "C" int hostsim_expr_eval(const char *rhs, int p, const char *const *parnames, int nvars,
                                 const char *const *varnames, const double *theta, int n, const double *X,
                                 double *value, double *grad, int *stats, const double *dir, double *fvv)
{
    std::vector<std::string> pn(parnames, parnames + p), vn(varnames, varnames + nvars);
    VmProgram prog;
    const std::string err = compile_expression(rhs, pn, vn, prog);
    if (!err.empty())
        return -1;
    std::vector<double> slot(VM_MAX_SLOTS);
    for (int i = 0; i < n; ++i)
    {
        double xr[VM_NX] = {0, 0, 0};
        for (int c = 0; c < nvars; ++c)
            xr[c] = X[i + (size_t)n * c];
        const bool want_fvv = dir && fvv && prog.nfvv > 0;
        vm_run(prog, theta, xr, dir, want_fvv ? prog.nfvv : prog.nops, slot.data());
        value[i] = slot[prog.value_slot];
        for (int k = 0; k < p; ++k)
            grad[i + (size_t)n * k] = slot[prog.grad_slot[k]];
        if (fvv)
            fvv[i] = want_fvv ? slot[prog.fvv_slot] : NAN;
    }
    stats[3] = prog.nfvv;
    stats[0] = prog.nops;
    stats[1] = prog.nvalue;
    stats[2] = prog.nconst;
    return 0;
}
