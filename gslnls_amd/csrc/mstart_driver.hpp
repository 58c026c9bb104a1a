// mstart_driver.hpp -- host side of multi-start: the modified Hickernell-Yuan bookkeeping.
//
// Replaces gsl_multistart_driver (src/nls_mstart.c:24-350) and the major loop with its stopping
// rule in C_nls_internal (src/nls.c:274-399, :518-531).  The reference evaluates the N sample
// points one after another; here every numerically heavy part of a major iteration is handed
// to a BatchEvaluator in two batches,
//     (1) concentration: all N slots at once (det filter + mstart_p LM iterations + det),
//     (2) local searches: every slot that reached ntix >= s, speculatively,
// and the order-dependent scalar bookkeeping (running best with 0.99 hysteresis, stale
// mchisq0/mchisq1 carry, top-q retention, dynamic ranges, NSP/NWSP counters, rejection
// limits, sampling exponents) is replayed sequentially from the per-point records exactly in
// the reference's loop order.  Each fit depends only on its own start, so the replay gives
// the reference's result.
//
// Multi-GPU: the evaluator computes only the shard [lo, hi) of a batch; `Comm::allgather`
// (RCCL over xGMI through torch.distributed in the Python host) completes the
// record array on every rank; every rank then replays the same commit, so all ranks hold the
// same state without further traffic.
//
// Header-only and free of HIP so that tests/hostsim can compile the same logic around a CPU
// evaluator built from the device headers.
#pragma once
#include <math.h>
#include <float.h>
#include <algorithm>
#include <numeric>
#include <vector>
#include "lm_core.hpp"
#include "sobol.hpp"
#include "trace_log.hpp"

namespace gslnls
{

// set through gslnls_set_interrupt_hook (capi.hip); polled once per major iteration
inline int (*ms_interrupt_hook)(void) = nullptr;

struct MsBatch
{
    int count = 0;
    int p = 0;
    int K = 0;                       // doubles per record (3p + 8): x[p], diag[p], x0[p], 8 scalars
    std::vector<long long> draw;     // >= 0: global draw index of a fresh Sobol point; -1: use `start`
    std::vector<double> start;       // count x p explicit start points (row-major)
    std::vector<double> range;       // 2p: current sampling ranges [l0, l1] pairs
    std::vector<double> kd;          // p: sampling exponents (pars->diag)
    int maxiter = 0;
    double dtol = 0.0;
    int always_fit = 0;
    int consecutive = -1;            // 1: draw[i] == draw[0] + i is known to hold, 0: known not to, -1: look
    std::vector<double> records;     // count x K, filled by the evaluator (+ all-gather) when it has no buffer of its own
    const double *rec = nullptr;     // the finished batch's records: `records`, or the evaluator's pinned staging buffer
                                     // (valid until the evaluator runs its next batch)
    bool host_records = true;        // false: leave the (gathered) records in device memory (throughput measurements)
};

struct MsComm
{
    int rank = 0, world = 1;
    // all-gather of equal-sized shards: gathers shard_buf[0 : per*K) of every rank into
    // all_buf[0 : world*per*K) on every rank (one RCCL all-gather over xGMI; gloo in CPU tests)
    int (*allgather)(void *ctx, int per_points, int K) = nullptr;
    void *ctx = nullptr;
    double *shard_buf = nullptr; // caller-owned (torch tensor), capacity cap_points/world... per*K doubles
    double *all_buf = nullptr;   // caller-owned, world*per*K doubles
    long long cap_points = 0;    // capacity of all_buf in points (>= world*per)
    int buffers_on_device = 0;
    // in-library collective (rccl_comm.hpp): when set, the batch kernel writes its shard straight into the
    // communicator's device buffer and the all-gather is enqueued behind it on the SAME stream -- nothing between
    // the kernel and the collective runs on the host.  Takes precedence over the callback form above.
    struct RcclComm *rccl = nullptr;
    int (*rccl_run)(struct MsEvaluator &ev, const MsComm &comm, struct MsBatch &b, int per, int lo, int hi) = nullptr;
    // test hook: take the collective path even with one rank (the all-gather of a 1-rank communicator)
    int force_collective = 0;
};

// value the status slot of a shard's first record carries when the rank could not compute its shard: every rank
// still enters the collective (a rank that returned early would leave the others blocked in it) and all of them fail
// together afterwards
constexpr double MS_SHARD_FAILED = -424242.0;

struct MsEvaluator
{
    virtual ~MsEvaluator() {}
    // compute the records of points [lo, hi) of the batch into out[0 : (hi-lo)*K)
    virtual int run(MsBatch &b, int lo, int hi, double *out, bool out_on_device) = 0;
    // copy ndoubles from src (device or host) to host dst
    virtual int fetch(const double *src, bool src_on_device, double *dst, size_t ndoubles) = 0;
    // stream-ordered form for the in-library collective: run() without the final synchronisation, the stream the
    // kernel went to, and a fetch enqueued on that stream (synchronises it)
    virtual int run_async(MsBatch &b, int lo, int hi, double *dev_out) { return run(b, lo, hi, dev_out, true); }
    virtual void *stream() { return nullptr; }
    virtual int fetch_stream(const double *dev_src, double *dst, size_t ndoubles) { return fetch(dev_src, true, dst, ndoubles); }
    // whole batch on this rank with the records left where the evaluator's copy engine put them (pinned host
    // memory): 8192 records are 0.9 MB, and zero-filling a vector plus one more memcpy of them cost more than the
    // batch kernel
    virtual int run_view(MsBatch &b, const double **view)
    {
        b.records.resize((size_t)b.count * b.K);
        const int rc = run(b, 0, b.count, b.records.data(), false);
        *view = b.records.data();
        return rc;
    }
    virtual int fetch_stream_view(const double *dev_src, size_t ndoubles, MsBatch &b, const double **view)
    {
        b.records.resize(ndoubles);
        const int rc = fetch_stream(dev_src, b.records.data(), ndoubles);
        *view = b.records.data();
        return rc;
    }
    // mark a device shard as failed (one double)
    virtual int poke(double *dev_dst, double value) { (void)dev_dst; (void)value; return -1; }
};


// stable ascending order with NaN last == R_orderVector1(.., nalast = TRUE, decreasing = FALSE)
inline void ms_order(const std::vector<double> &v, std::vector<int> &ord)
{
    ord.resize(v.size());
    std::iota(ord.begin(), ord.end(), 0);
    std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) {
        const bool na = std::isnan(v[a]), nb = std::isnan(v[b]);
        if (na || nb)
            return !na && nb;
        return v[a] < v[b];
    });
}

struct MsState
{
    // mdata + the pdata fields the driver touches (src/gsl_nls.h:20-50, :78-107)
    int N = 0, p = 0, mp = 0, q = 0, s = 0, niter = 0, maxstart = 0, minsp = 0;
    bool all_start = true;
    std::vector<int> has_start; // 2p
    double r = 0, tol = 0, dtol = 1e-6;
    std::vector<int> ntix, luchange, order;
    int mstop = ST_CONTINUE, mstarts = 0, nsp = 0, nwsp = 0;
    double rejectscl = 1.25;
    double mssropt[2] = {INFINITY, INFINITY}, ssrconv[2] = {1.0, 1.0};
    std::vector<double> start, maxlims, mssr, mx, diag, mpopt, mpopt1, lu; // lu: 2p [lower row; upper row] or empty
    long long next_draw = 0; // global Sobol draw counter (gsl_qrng state)
    double xtol = 0, ftol = 0;
    long long total_fits = 0; // concentration + local-search fits actually run (for the benchmark)
    bool second_pass = false; // the robust second pass (use_weights of gsl_multistart_driver: only the trace line differs)
};

inline int ms_run_batch(MsEvaluator &ev, const MsComm &comm, MsBatch &b)
{
    b.rec = nullptr;
    if (comm.world <= 1 && !(comm.rccl && comm.force_collective))
        return b.host_records ? ev.run_view(b, &b.rec) : ev.run(b, 0, b.count, nullptr, true);
    // contiguous blocks of ceil(count / world) points per rank; the last block may be short
    const int per = (b.count + comm.world - 1) / comm.world;
    const int lo = std::min(b.count, comm.rank * per), hi = std::min(b.count, lo + per);
    if (comm.rccl && comm.rccl_run)
        return comm.rccl_run(ev, comm, b, per, lo, hi);
    if (!comm.allgather || !comm.shard_buf || !comm.all_buf || (long long)per * comm.world > comm.cap_points)
        return -1; // (the registration is the same on every rank: all of them return here, none enters the collective)
    // the callback form always completes the batch in host memory -- also when the caller asked for no records
    // (host_records == false): the status words of the shards are read from there
    b.records.assign((size_t)b.count * b.K, 0.0);
    int rc = 0;
    if (hi > lo)
        rc = ev.run(b, lo, hi, comm.shard_buf, comm.buffers_on_device != 0);
    const int status_slot = 3 * b.p + 6; // K = 3p + 8: ..., niter, status, nevalf
    if (rc && hi > lo)
    {
        // this rank failed: say so in its shard and enter the collective anyway
        if (comm.buffers_on_device)
            (void)ev.poke(comm.shard_buf + status_slot, MS_SHARD_FAILED);
        else
            comm.shard_buf[status_slot] = MS_SHARD_FAILED;
    }
    const int rc2 = comm.allgather(comm.ctx, per, b.K);
    if (rc2)
        return rc2;
    const int rc3 = ev.fetch(comm.all_buf, comm.buffers_on_device != 0, b.records.data(), (size_t)b.count * b.K);
    if (rc3)
        return rc3;
    for (int r = 0; r < comm.world; ++r)
        if ((long long)r * per < b.count && b.records[((size_t)r * per) * b.K + status_slot] == MS_SHARD_FAILED)
            return rc ? rc : -1;
    b.rec = b.records.data();
    return 0;
}

// one major iteration == one call of gsl_multistart_driver (src/nls_mstart.c:24-350)
inline int ms_major_iteration(MsState &m, MsEvaluator &ev, const MsComm &comm)
{
    const int p = m.p, N = m.N, K = 3 * p + 8;
    double mchisq0 = INFINITY, mchisq1 = INFINITY; // locals of the reference function (:38)

    // ---- (1) sample + concentrate, all N slots in one batch (:42-128) ----
    MsBatch b;
    b.count = N;
    b.p = p;
    b.K = K;
    b.draw.assign(N, -1);
    b.start = m.mx;
    b.range = m.start;
    b.kd = m.diag;
    b.maxiter = m.mp;
    b.dtol = m.dtol;
    b.always_fit = 0;
    for (int nn = 0; nn < N; ++nn)
        if (m.ntix[nn] == 0)
            b.draw[nn] = m.next_draw++; // gsl_qrng_get in loop order (:48)
    int rc = ms_run_batch(ev, comm, b);
    if (rc)
        return rc;

    for (int nn = 0; nn < N; ++nn)
    {
        const double *rec = b.rec + (size_t)nn * K;
        const double *rx = rec, *rdiag = rec + p;
        (void)rdiag;
        const double *rx0 = rec + 2 * p, *sc = rec + 3 * p;
        const double chisq0 = sc[0], chisq1 = sc[1], det0 = sc[2], det1 = sc[3];
        const double ssr_start = sc[4], rniter = sc[5];
        m.mssr[nn] = NAN; // NA_REAL (:44)
        if (b.draw[nn] >= 0)
        {
            // the freshly sampled point is what the evaluator fitted from; keep it like gsl_matrix_set (:58-68)
            for (int k = 0; k < p; ++k)
                m.mx[(size_t)nn * p + k] = rx0[k];
        }
        if (det0 > m.dtol)
        {
            m.total_fits++;
            // driver2 sets chisq0 <- chisq1(in) at the top of every iteration; with a single iteration the
            // reported mchisq0 is therefore the value carried over from the previous point (:91-92)
            mchisq0 = (rniter <= 1.0) ? mchisq1 : chisq0;
            mchisq1 = chisq1;
            if (mchisq1 < INFINITY)
            {
                if (det1 > m.dtol)
                {
                    for (int k = 0; k < p; ++k)
                        m.mx[(size_t)nn * p + k] = rx[k];
                    m.mssr[nn] = mchisq1;
                    if (mchisq1 < 0.99 * fmin(m.mssropt[0], m.mssropt[1]))
                    {
                        m.mssropt[0] = mchisq1;
                        m.ssrconv[0] = mchisq0 - mchisq1;
                        for (int k = 0; k < p; ++k)
                            m.mpopt[k] = rx[k];
                    }
                }
                else if (mchisq1 < 0.99 * fmin(m.mssropt[0], m.mssropt[1]))
                {
                    m.mssropt[1] = mchisq1;
                    m.ssrconv[1] = mchisq0 - mchisq1;
                    for (int k = 0; k < p; ++k)
                        m.mpopt1[k] = rx[k];
                }
            }
        }
        else if (!(m.mssropt[0] < INFINITY) && det0 > DBL_EPSILON)
        {
            // back-up in case no stationary points are found (:117-127)
            mchisq1 = ssr_start;
            if (mchisq1 < 0.99 * m.mssropt[1])
            {
                m.mssropt[1] = mchisq1;
                m.ssrconv[1] = mchisq0 - mchisq1;
                for (int k = 0; k < p; ++k)
                    m.mpopt1[k] = m.mx[(size_t)nn * p + k];
            }
        }
    }

    // ---- reduce sample points (:131-138) ----
    ms_order(m.mssr, m.order);
    for (int nn = 0; nn < N; ++nn)
    {
        const int o = m.order[nn];
        if (nn < m.q && !std::isnan(m.mssr[o]))
            m.ntix[o] += 1;
        else
            m.ntix[o] = 0;
    }

    // ---- dynamic lower/upper limits (:141-233) ----
    if (!m.all_start)
    {
        double pk, pmin = 0.0, pmax = 1.0;
        double mssr_diff = m.mssr[m.order[0]];
        if (!std::isnan(mssr_diff))
        {
            for (int nn = N - 1; nn > 0; --nn)
                if (!std::isnan(m.mssr[m.order[nn]]))
                {
                    mssr_diff -= m.mssr[m.order[nn]];
                    break;
                }
        }
        if (std::isnan(mssr_diff) || fabs(mssr_diff) < 1e-5)
            for (int k = 0; k < p; ++k)
                m.luchange[k] += 1;
        const bool has_lu = !m.lu.empty();
        for (int k = 0; k < p; ++k)
        {
            int luchange_add = 0;
            if (m.mssropt[0] < INFINITY)
            {
                const std::vector<double> &best = (m.mssropt[1] < m.mssropt[0]) ? m.mpopt1 : m.mpopt;
                pmin = best[k];
                pmax = best[k];
            }
            for (int nn = 0; nn < m.q; ++nn)
            {
                const int o = m.order[nn];
                if (m.ntix[o] > 0 && m.mssr[o] < 1.25 * m.mssropt[0])
                {
                    pk = m.mx[(size_t)o * p + k];
                    pmin = (pk < pmin) ? pk : pmin;
                    pmax = (pk > pmax) ? pk : pmax;
                }
            }
            const double l0 = m.start[2 * k], l1 = m.start[2 * k + 1];
            if (!m.has_start[2 * k])
            {
                if (pmin < 0.9 * l0 || m.luchange[k] > 4)
                {
                    m.start[2 * k] = l0 < 0 ? fmax(l0 / pow(-1e-5 * (l0 - 1.0), 0.1) - 1.0, -1.0E5) : -0.1;
                    if (has_lu)
                        m.start[2 * k] = fmax(m.start[2 * k], m.lu[k]);
                    m.maxlims[2 * k] = fmin(m.start[2 * k], m.maxlims[2 * k]);
                    luchange_add = -1;
                }
                else if (pmin > 0.2 * l0)
                {
                    m.start[2 * k] = fmin(l0 / pow(-0.05 * (l0 - 1.0), 0.05), -0.01);
                    if (has_lu)
                        m.start[2 * k] = fmax(m.start[2 * k], m.lu[k]);
                    luchange_add = (m.mssropt[0] < INFINITY) ? -1 : 1;
                }
                else
                    luchange_add = 1;
            }
            if (!m.has_start[2 * k + 1])
            {
                if (pmax > 0.9 * l1 || m.luchange[k] > 4)
                {
                    m.start[2 * k + 1] = fmin(l1 / pow(1e-5 * (l1 + 1.0), 0.1) + 1.0, 1.0E5);
                    if (has_lu)
                        m.start[2 * k + 1] = fmin(m.start[2 * k + 1], m.lu[p + k]);
                    m.maxlims[2 * k + 1] = fmax(m.start[2 * k + 1], m.maxlims[2 * k + 1]);
                    luchange_add = -1;
                }
                else if (pmax < 0.2 * l1)
                {
                    m.start[2 * k + 1] = fmax(l1 / pow(0.05 * (l1 + 1.0), 0.05), 0.1);
                    if (has_lu)
                        m.start[2 * k + 1] = fmin(m.start[2 * k + 1], m.lu[p + k]);
                    luchange_add = (m.mssropt[0] < INFINITY) ? -1 : 1;
                }
                else
                    luchange_add = 1;
            }
            if (luchange_add)
                m.luchange[k] = (luchange_add > 0) ? m.luchange[k] + 1 : 0;
        }
    }

    // ---- (2) local optimisation stage (:236-349): speculative batch, sequential commit ----
    std::vector<int> cand;
    for (int nn = 0; nn < N; ++nn)
        if (m.ntix[nn] >= m.s)
            cand.push_back(nn);
    if (!cand.empty())
    {
        MsBatch lb;
        lb.count = (int)cand.size();
        lb.p = p;
        lb.K = K;
        lb.draw.assign(lb.count, -1);
        lb.start.resize((size_t)lb.count * p);
        for (int c = 0; c < lb.count; ++c)
            for (int k = 0; k < p; ++k)
                lb.start[(size_t)c * p + k] = m.mx[(size_t)cand[c] * p + k];
        lb.range = m.start;
        lb.kd = m.diag;
        lb.maxiter = m.niter;
        lb.dtol = m.dtol;
        lb.always_fit = 1;
        MsComm solo; // the handful of local searches is replicated on every rank (identical results)
        rc = ms_run_batch(ev, solo, lb);
        if (rc)
            return rc;
        for (int c = 0; c < lb.count; ++c)
        {
            const int nn = cand[c];
            const double *rec = lb.rec + (size_t)c * K;
            const double *rx = rec, *rdiag = rec + p;
            m.ntix[nn] = 0;
            m.nwsp += 1;
            if (m.nsp == 0 || m.mssr[nn] < (1 + m.tol) * m.mssropt[0])
            {
                m.total_fits++;
                const double *sc = rec + 3 * p;
                const double rniter = sc[5], det1 = sc[3];
                // mchisq1 = mssr[nn] before driver2 (:253): one iteration -> mchisq0 is that value
                mchisq0 = (rniter <= 1.0) ? m.mssr[nn] : sc[0];
                mchisq1 = sc[1];
                if (mchisq1 < INFINITY && (m.nsp == 0 || mchisq1 < 0.99 * m.mssropt[0]) &&
                    (det1 > m.dtol || mchisq1 < (2 * m.ftol)))
                {
                    int reject = 0;
                    if (m.rejectscl > 0)
                    {
                        for (int k = 0; k < p; ++k)
                        {
                            const double xk = rx[k];
                            if (m.all_start)
                                reject += (xk > fmax(m.maxlims[2 * k + 1], 1.0) || xk < fmin(m.maxlims[2 * k], -1.0));
                            else
                                reject += (xk > fmax(pow(m.maxlims[2 * k + 1], m.rejectscl), 1.0) ||
                                           xk < fmin(-pow(-m.maxlims[2 * k], m.rejectscl), -1.0));
                            if (reject > 0)
                                break;
                        }
                        if (!m.all_start)
                            m.rejectscl += 0.05;
                    }
                    if (!reject)
                    {
                        m.mssropt[0] = mchisq1;
                        m.ssrconv[0] = mchisq0 - mchisq1;
                        for (int k = 0; k < p; ++k)
                            m.mpopt[k] = rx[k];
                        m.nsp += 1;
                        m.nwsp = 0;
                        if (m.rejectscl > 0)
                            m.rejectscl = 1.25;
                        if (m.all_start)
                        {
                            double diagmin = rdiag[0];
                            for (int k = 1; k < p; ++k)
                                diagmin = fmin(diagmin, rdiag[k]);
                            for (int k = 0; k < p; ++k)
                                m.diag[k] = pow(diagmin / rdiag[k], 0.25);
                        }
                        // (src/nls_mstart.c:331-337)
                        trace_printf("mstart%s ssr* = %g, det(JTJ) = %g, NSP = %d, NWSP = %d, par = (", m.second_pass ? " (second pass)" : "",
                                     m.mssropt[0], det1, m.nsp, m.nwsp);
                        trace_vector(rx, p);
                    }
                }
                else if (mchisq1 < 0.99 * fmin(m.mssropt[0], m.mssropt[1]))
                {
                    m.mssropt[1] = mchisq1;
                    m.ssrconv[1] = mchisq0 - mchisq1;
                    for (int k = 0; k < p; ++k)
                        m.mpopt1[k] = rx[k];
                }
            }
        }
    }
    return 0;
}

// major loop + stopping rule (src/nls.c:372-399)
inline int ms_major_loop(MsState &m, MsEvaluator &ev, const MsComm &comm, const double *startptr)
{
    do
    {
        if (ms_interrupt_hook && ms_interrupt_hook())
            return -102; // GSLNLS_E_INTERRUPTED
        const int rc = ms_major_iteration(m, ev, comm);
        if (rc)
            return rc;
        m.mstarts += 1;
        if (m.mstarts > m.maxstart)
            m.mstop = ST_EMAXITER;
        if (m.nsp >= m.minsp && m.nwsp > (m.r + sqrt(m.r) * m.nsp))
            m.mstop = ST_SUCCESS;
        if (!(m.mstarts % 10) && !(m.mssropt[0] < INFINITY))
        {
            m.dtol = fmax(0.5 * m.dtol, DBL_EPSILON);
            if (!(m.mstarts % 100))
                for (int k = 0; k < m.p; ++k)
                {
                    m.start[2 * k] = startptr[2 * k];
                    m.start[2 * k + 1] = startptr[2 * k + 1];
                }
        }
    } while (m.mstop == ST_CONTINUE);
    return 0;
}

// the two lines that close the multi-start stage of a verbose call (src/nls.c:510-517)
inline void ms_trace_finished(const MsState &m)
{
    if (m.mstop == ST_SUCCESS)
        trace_printf("multi-start algorithm finished successfully (NSP = %d, NWSP = %d, # iterations = %d)\n", m.nsp, m.nwsp, m.mstarts);
    if (m.mstop == ST_EMAXITER)
        trace_printf("multi-start algorithm reached max. number of global iterations (NSP = %d, NWSP = %d, # iterations = %d)\n", m.nsp,
                     m.nwsp, m.mstarts);
    trace_printf("*******************\n");
}

// state set-up of src/nls.c:297-369
inline void ms_init(MsState &m, int p, const int *ci, const double *cd, const double *startptr, const int *has_start,
                    const double *lupars)
{
    m.p = p;
    m.N = ci[6];
    m.mp = ci[7];
    m.q = ci[8];
    m.s = ci[9];
    m.niter = ci[10];
    m.maxstart = ci[11];
    m.minsp = ci[12];
    m.r = cd[8];
    m.tol = cd[9];
    m.xtol = cd[5];
    m.ftol = cd[6];
    m.dtol = 1.0e-6;
    m.all_start = true;
    m.has_start.assign(has_start, has_start + 2 * p);
    m.ntix.assign(m.N, 0);
    m.luchange.assign(p, 0);
    m.mstop = ST_CONTINUE;
    m.mstarts = m.nsp = m.nwsp = 0;
    m.rejectscl = 1.25;
    m.mssropt[0] = m.mssropt[1] = INFINITY;
    m.ssrconv[0] = m.ssrconv[1] = 1.0;
    m.start.assign(startptr, startptr + 2 * p);
    m.maxlims.assign(startptr, startptr + 2 * p);
    m.mssr.assign(m.N, NAN);
    m.mx.assign((size_t)m.N * p, 0.0);
    m.diag.assign(p, 0.0);
    m.mpopt.assign(p, 0.0);
    m.mpopt1.assign(p, 0.0);
    m.lu.clear();
    if (lupars)
    {
        m.lu.resize(2 * p);
        for (int k = 0; k < p; ++k)
        {
            m.lu[k] = isfinite(lupars[2 * k]) ? lupars[2 * k] : -INFINITY;
            m.lu[p + k] = isfinite(lupars[2 * k + 1]) ? lupars[2 * k + 1] : INFINITY;
        }
    }
    for (int k = 0; k < p; ++k)
    {
        if (!m.has_start[2 * k] || !m.has_start[2 * k + 1])
        {
            m.diag[k] = 1.0;
            m.all_start = false;
        }
        else
        {
            m.diag[k] = 0.75;
            if (m.start[2 * k] + m.xtol > m.start[2 * k + 1])
                m.rejectscl = -1.0;
        }
    }
    m.next_draw = 0;
    m.total_fits = 0;
}

} // namespace gslnls
