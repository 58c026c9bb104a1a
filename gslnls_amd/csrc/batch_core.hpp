// batch_core.hpp -- one complete small fit executed by ONE lane (multi-start) .
//
// Device twin of what the reference does for a single sampled point inside
// gsl_multistart_driver (src/nls_mstart.c:72-95 and :245-255):
//     det_eval_jtj            src/nls_utils.c:23-53   f, J at the point, det(J^T J)
//     if det > dtol:
//         gsl_multifit_nlinear_(w)init + gsl_multifit_nlinear_driver2(maxiter, xtol, 1e-3, ftol)
//         det_cholesky_jtj    src/nls_utils.c:55-73   at the point the fit ended on
// The sequential bookkeeping around it (running best with its 0.99 hysteresis, top-q
// retention, dynamic ranges, acceptance of stationary points) stays on the host
// (mstart_driver.hpp) and is replayed in the reference's order from these per-point
// records, so thousands of points can be fitted at once without changing the result.
#pragma once
#include "lm_core.hpp"
#include "rowops.hpp"
#include "sobol.hpp"

namespace gslnls
{

struct MsParams
{
    LmParams prm;   // maxiter = mstart_p (concentration) or mstart_maxiter (local search); gtol = 1e-3
    double dtol;    // determinant tolerance of the det filter
    int n;          // residual rows
    int always_fit; // local search stage: no det filter in front of the fit
};

// per-point record, all doubles so one all-gather moves it: K = 3P + 8
template <int P>
struct MsRecord
{
    static constexpr int K = 3 * P + 8;
    double x[P];     // where the fit ended (or the sampled point itself when not fitted)
    double diag[P];  // trust_state->diag at the end (src/nls_mstart.c:324-326)
    double x0[P];    // the point the fit started from (the freshly sampled point for new draws)
    double chisq0;   // ssr before the last iteration
    double chisq1;   // ssr at the end
    double det0;     // det(J^T J) at the sampled point
    double det1;     // det(J^T J) where the fit ended
    double ssr_start;
    double niter;
    double status;   // GSL status of driver2, or -2 (CONTINUE) when the point was not fitted
    double nevalf;
};

// RowSrc: void operator()(int i, double *xr, double &y, double &sw) const
template <class M, int JAC, class RowSrc>
GSLNLS_HD void ms_pass(const LmState<M::P> &s, const MsParams &mp, const RowSrc &rows, PassSums<M::P> &acc)
{
    constexpr int P = M::P;
    double th[P], delta[P];
#pragma unroll
    for (int k = 0; k < P; ++k)
        th[k] = (s.phase == PH_FVV) ? s.x[k] : s.xt[k];
    fd_deltas<P>(th, mp.prm.h_df, delta);
    pass_zero<P>(acc);
    auto do_row = [&](const double *xr, double y, double sw) {
        double Jrow[P];
        if (s.phase == PH_FVV)
        {
            const double fv = row_fvv<M, JAC>(th, s.vel, delta, mp.prm.h_fvv, mp.prm.fvv_analytic != 0, xr, y, sw,
                                              Jrow, &acc.badj);
#pragma unroll
            for (int k = 0; k < P; ++k)
                acc.g[k] += Jrow[k] * fv;
        }
        else
        {
            const double f = row_fj<M, JAC>(th, delta, xr, y, sw, Jrow, &acc.badj);
            acc_fj<P>(acc, f, Jrow);
        }
    };
    if constexpr (RowSrc::STATIC_N > 0 && RowSrc::LPF > 1)
    {
        // tiny data set in registers, LPF lanes per fit: lane `sub` of a group takes rows sub, sub + LPF, ... (every
        // lane runs the same instructions on its own row of the step), then the groups add their sums up (lane
        // exchange inside quads: every lane of a group ends with the same bits) and all of them run lm_advance
        constexpr int LPF = RowSrc::LPF;
#pragma unroll
        for (int k = 0; k < RowSrc::STATIC_N / LPF; ++k)
        {
            if (k * LPF < mp.n)
            {
                double xr[M::NX], y = rows.y[k * LPF], sw = rows.sw[k * LPF];
#pragma unroll
                for (int c = 0; c < M::NX; ++c)
                    xr[c] = rows.x[k * LPF][c];
#pragma unroll
                for (int t = 1; t < LPF; ++t)
                {
                    const bool mine = rows.sub == t;
#pragma unroll
                    for (int c = 0; c < M::NX; ++c)
                        xr[c] = mine ? rows.x[k * LPF + t][c] : xr[c];
                    y = mine ? rows.y[k * LPF + t] : y;
                    sw = mine ? rows.sw[k * LPF + t] : sw;
                }
                // a row beyond n (odd n) is switched off through its weight, as in the grid-per-fit pass
                const bool live = k * LPF + rows.sub < mp.n;
                double Jrow[P];
                if (s.phase == PH_FVV)
                {
                    double fv = row_fvv<M, JAC>(th, s.vel, delta, mp.prm.h_fvv, mp.prm.fvv_analytic != 0, xr, y,
                                                live ? sw : 0.0, Jrow, &acc.badj);
                    fv = live ? fv : 0.0;
#pragma unroll
                    for (int q = 0; q < P; ++q)
                        acc.g[q] += Jrow[q] * fv;
                }
                else
                {
                    double f = row_fj<M, JAC>(th, delta, xr, y, live ? sw : 0.0, Jrow, &acc.badj);
                    f = live ? f : 0.0;
                    acc_fj<P>(acc, f, Jrow);
                }
            }
        }
        rows.combine(acc);
    }
    else if constexpr (RowSrc::STATIC_N > 0)
    {
        // tiny data sets held in registers: the loop is unrolled so that every row is a fixed register, and
        // the (uniform) row count only switches whole rows off
#pragma unroll
        for (int i = 0; i < RowSrc::STATIC_N; ++i)
        {
            if (i < mp.n)
                do_row(rows.x[i], rows.y[i], rows.sw[i]);
        }
    }
    else
    {
        for (int i = 0; i < mp.n; ++i)
        {
            double xr[M::NX], y, sw;
            rows(i, xr, y, sw);
            do_row(xr, y, sw);
        }
    }
}

// The fit of one point in three pieces, so that a lane can be handed its next point as soon as this one is done
// (batch_kernels.hpp, ms_fit_refill_kernel); ms_fit_point below strings them together for one point per lane.
// One call site for the pass and one for the state machine (the first trip doubles as det_eval_jtj at the sampled
// point): the lanes of a wavefront hold different fits, and every extra inlined copy is both code the instruction
// cache has to hold and a place where lanes in different states wait for each other.
template <int P>
struct MsPointState
{
    LmState<P> s;
    double start[P];
    double det0, ssr_start;
    bool first, fitted;
};

template <int P>
GSLNLS_HD void ms_point_begin(MsPointState<P> &q, const double *start, const double *lupars)
{
    lm_state_reset<P>(q.s, start, lupars);
#pragma unroll
    for (int k = 0; k < P; ++k)
        q.start[k] = start[k];
    q.det0 = 0.0;
    q.ssr_start = 0.0;
    q.first = true;
    q.fitted = false;
}

// consume the sums of the pass that q.s.phase asked for; returns true when the point is finished
template <int P>
GSLNLS_HD bool ms_point_step(MsPointState<P> &q, const MsParams &mp, const PassSums<P> &acc)
{
    if (q.first)
    {
        q.first = false;
        double det0 = det_cholesky<P>(acc.A);
        if (mp.prm.jac_analytic && !(acc.badj == 0.0))
            det0 = 0.0; // eval_df failed (src/nls_utils.c:47-48)
        q.det0 = det0;
        q.ssr_start = acc.ssr;
        if (!(mp.always_fit || det0 > mp.dtol))
            return true;
        q.fitted = true;
    }
    lm_advance<P>(q.s, acc, mp.prm);
    return q.s.phase == PH_DONE;
}

template <int P>
GSLNLS_HD void ms_point_record(const MsPointState<P> &q, const PassSums<P> &last, MsRecord<P> &rec)
{
    const LmState<P> &s = q.s;
    rec.det0 = q.det0;
    rec.ssr_start = q.ssr_start;
#pragma unroll
    for (int k = 0; k < P; ++k)
        rec.x0[k] = q.start[k];
    if (q.fitted)
    {
        rec.det1 = det_cholesky<P>(s.A);
        rec.chisq0 = s.chisq0;
        rec.chisq1 = s.chisq1;
        rec.niter = (double)s.niter;
        rec.status = (double)s.status;
        rec.nevalf = (double)s.nevalf;
#pragma unroll
        for (int k = 0; k < P; ++k)
        {
            rec.x[k] = s.x[k];
            rec.diag[k] = s.diag[k];
        }
    }
    else
    {
        rec.det1 = 0.0;
        rec.chisq0 = INFINITY;
        rec.chisq1 = last.ssr;
        rec.niter = 0.0;
        rec.status = (double)ST_CONTINUE;
        rec.nevalf = 1.0;
#pragma unroll
        for (int k = 0; k < P; ++k)
        {
            rec.x[k] = q.start[k];
            rec.diag[k] = 1.0;
        }
    }
}

template <class M, int JAC, class RowSrc>
GSLNLS_HD void ms_fit_point(const MsParams &mp, const RowSrc &rows, const double *start, const double *lupars,
                            MsRecord<M::P> &rec)
{
    constexpr int P = M::P;
    MsPointState<P> q;
    ms_point_begin<P>(q, start, lupars);
    PassSums<P> acc;
    for (int guard = 0; guard < 100000; ++guard)
    {
        ms_pass<M, JAC>(q.s, mp, rows, acc);
        if (ms_point_step<P>(q, mp, acc))
            break;
    }
    ms_point_record<P>(q, acc, rec);
}

} // namespace gslnls
