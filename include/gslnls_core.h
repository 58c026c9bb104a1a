/*
 * gslnls_core.h -- C ABI of the MI355X-native nonlinear least-squares core
 * (libgslnls_hip.so).  Plain pointers and sizes only; no R, GSL or torch types.
 *
 * This is the drop-in boundary for the reference's .Call entries
 *     C_nls        src/init.c:15, src/nls.c:54   (12 SEXP arguments)
 *     C_nls_large  src/init.c:16, src/nls_large.c:66
 * An R-side shim (integration/r_shim/) unpacks the SEXPs exactly
 * as src/nls.c:76-263 does and calls gslnls_nls(); the list it returns is filled
 * from gslnls_result in the order of src/nls.c:636-645.
 *
 * The one thing that cannot cross unchanged is the model: the reference evaluates
 * R closures fn/jac/fvv with Rf_eval (src/nls.c:836-837,:885-886,:946-948).  Here
 * `fn` becomes a gslnls_model: a registry id of a device row model plus the data
 * columns the closure's environment held (R/nls.R:565).  jac / fvv become flags
 * ("analytic available and requested" vs "finite differences").
 *
 * All numerics are IEEE fp64.  Matrices follow R's layout (column-major) wherever
 * the reference hands R matrices across the boundary.
 */
#ifndef GSLNLS_CORE_H
#define GSLNLS_CORE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* device row models (gslnls_amd/csrc/models.hpp) */
#define GSLNLS_MODEL_EXPDECAY 1 /* A*exp(-lam*x)+b           p=3  R/nls.R:143-151 */
#define GSLNLS_MODEL_MISRA1A 2  /* b1*(1-exp(-b2*x))         p=2  R/nls_test.R:174,:793 */
#define GSLNLS_MODEL_GAUSSPK 3  /* a*exp(-(x-b)^2/(2c^2))    p=3  README.md:545 */
#define GSLNLS_MODEL_GAUSS1 4   /* NIST Gauss1 family        p=8  R/nls_test.R:301 */
#define GSLNLS_MODEL_EXPR 100  /* any formula right-hand side: compiled to a device program with symbolic
                                   gradient (csrc/expr_compile.hpp); p <= 512, at most 8 regressor columns (beyond 9
                                   parameters or 3 columns: the wide path; beyond 64 parameters: the Jacobian as a
                                   matrix in HBM, csrc/bd_host.hpp) */
#define GSLNLS_MODEL_GLMEXP 5   /* exp(a_i . theta), dense A n x p ROW-major in `x`, nx = p in {16,32,64};
                                   gsl_nls_large only */

/* status codes placed in `conv` (GSL errno values) */
#define GSLNLS_SUCCESS 0
#define GSLNLS_FAILURE (-1)
#define GSLNLS_CONTINUE (-2)
#define GSLNLS_EINVAL 4
#define GSLNLS_EBADFUNC 9
#define GSLNLS_EMAXITER 11
#define GSLNLS_ENOPROG 27
/* library-level errors (never produced by the reference): returned by the entry
 * points themselves, never stored in `conv` */
#define GSLNLS_E_NODEVICE (-100)   /* no HIP device / HIP runtime failure: the path fails loudly */
#define GSLNLS_E_INTERRUPTED (-102) /* the interrupt hook asked to stop */
#define GSLNLS_E_UNSUPPORTED (-101) /* combination not lowered to the device (e.g. GLS weight matrix) */

typedef struct gslnls_model
{
    int id;           /* GSLNLS_MODEL_* */
    int p;            /* number of parameters (must match the model) */
    int nx;           /* number of regressor columns */
    const double *x;  /* n x nx, column-major; host memory unless x_on_device */
    int x_on_device;  /* x, y and swts are device pointers already resident in HBM */
    /* GSLNLS_MODEL_EXPR only: deparse(formula[[3]]) and the names its symbols resolve to */
    const char *expr;
    const char *const *parnames; /* [p]  parameter names, in the order of `start` */
    const char *const *xnames;   /* [nx] data column names, in the order of the columns of x */
    int lowering;                /* GSLNLS_LOWER_AUTO / _VM / _JIT */
} gslnls_model;

/* how a GSLNLS_MODEL_EXPR reaches the device: interpreted per row (no latency), or printed as a C++ row model and
   compiled IN PROCESS (hiprtc: the compiler that ships with the HIP runtime -- no hipcc, no headers, no child process on
   the host) into a code object that instantiates the same kernels, cached on disk by content hash; it then runs like a
   hand-written model, with results identical bit for bit to the interpreter's.  AUTO: the first fit of a formula runs
   interpreted and starts the build on a background thread, fits that come after the build use the native kernels.
   VM: interpreter only.  JIT: build now (1-3 s once per formula and Jacobian kind), fail if that is impossible.
   p > 9 (the wide path, MFMA J^T J tiles) always runs native code. */
#define GSLNLS_LOWER_AUTO 0
#define GSLNLS_LOWER_VM 1
#define GSLNLS_LOWER_JIT 2

/* Build (or find in the cache) the native code of an expression model ahead of time; needs no device and no compiler
   beyond the HIP runtime's own.  The path of the cached code object (analytic-Jacobian unit) is copied to path_out.
   Replaces nothing in the reference: the analogue is the closure construction of R/nls.R:565,588-599, done once per
   formula.  gslnls_expr_native_state: 0 not requested, 1 being built, 2 ready, -1 failed (jac: 1 analytic, 0 forward). */
int gslnls_expr_build(const gslnls_model *fn, char *path_out, int path_cap);
int gslnls_expr_native_state(const gslnls_model *fn, int jac);
/* start the build on a background thread and return at once (what the first GSLNLS_LOWER_AUTO fit does by itself) */
int gslnls_expr_prefetch(const gslnls_model *fn, int jac);
/* Call before the process exits when GSLNLS_LOWER_AUTO or gslnls_expr_prefetch may have started background builds:
   queued builds are dropped, the one in flight is waited for (<= a few seconds).  A compiler thread that is still
   running when the C runtime tears down the compiler's own static objects takes the process with it.  Idempotent;
   afterwards new formulas are served by the interpreter (or built synchronously under GSLNLS_LOWER_JIT). */
void gslnls_shutdown(void);

/* mirrors the VECSXP C_nls returns (src/nls.c:632-812).  Pointers may be NULL to skip. */
typedef struct gslnls_result
{
    double *par;      /* [p]   estimated parameters (start values on failure) */
    double *covar;    /* [p*p] column-major (J^T J)^-1 ; NaN on failure */
    double *resid;    /* [n]   weighted residuals sqrt(w)(f - y) */
    double *grad;     /* [n*p] column-major weighted Jacobian (src/nls.c:718) */
    int niter;
    int conv;         /* status code */
    double ssr;       /* chisq1 */
    double ssrtol;    /* chisq0 - chisq1 */
    int neval[3];     /* f, J, fvv with the reference's accounting (App. A.8) */
    int info;
    double chisq_init;
    /* irls slot (src/nls.c:756-791); arrays [n] or NULL */
    double *irls_weights, *irls_psi, *irls_dpsi;
    double irls_sigma, irls_tol;
    int irls_status, irls_niter;
    /* traces when control_int[1] != 0: (maxiter+1) x p column-major, and maxiter+1 */
    double *partrace, *ssrtrace;
    /* multi-start bookkeeping (not part of the R list; exposed for tests) */
    int mstart_nsp, mstart_nwsp, mstart_iters, mstart_stop;
    double mstart_ssropt;
    /* host wall clock of the solve loop, milliseconds: first launch enqueued ... completion word seen (the device
     * time of the same loop, by HIP events, accumulates in gslnls_dense_loop_event_stats) */
    float loop_ms;
    int n_launches;   /* step-kernel launches issued for this call */
    /* 2-norm condition number of the column-scaled normal matrix C = S J^T J S, S = diag(J^T J)^-1/2, at the final
     * point (NaN when the fit failed).  The device path always solves the normal equations (control_int[4] ==
     * cholesky in the reference's encoding, R/nls.R:1186); a caller that asked for "qr" or "svd" compares this with
     * GSLNLS_COND_LIMIT and keeps its own QR path beyond it -- see gslnls_solver_served(). */
    double jtj_cond;
    int n_steps;      /* passes over the rows (trial steps incl. the initial point) the device ran for this call; equals
                         n_launches on the launch-per-step kernel, while the one-launch-per-fit kernel has n_launches = 1 */
    int code_path;    /* which device code evaluated the model rows of the (last) solve: 0 hand-written row model,
                         1 interpreted expression, 2 expression compiled in process (hiprtc), 3 the same on the wide path
                         (p > 9: MFMA J^T J tiles), 4 Jacobian as a matrix in HBM (function models, p > 64: csrc/bd_host.hpp) */
} gslnls_result;

/* Solver routing rule of the boundary.  control_int[4]: 0 qr (the R default), 1 cholesky, 2 svd (R/nls.R:702).
 * cholesky requests are always served.  qr / svd requests are served on the normal equations as long as
 * kappa_2(C) <= GSLNLS_COND_LIMIT: the normal equations lose about log10(kappa(C))/2 more digits than a QR of J, so
 * below 1e10 the coefficients keep >= 6 significant digits (the reference's own tests compare at eps^(1/4) = 1.22e-4).
 * Beyond the limit gslnls_solver_served() returns 0 and the R shim re-runs the fit through the unchanged GSL path. */
#define GSLNLS_COND_LIMIT 1e10
int gslnls_solver_served(const int *control_int, const gslnls_result *res);

/*
 * gslnls_nls -- replaces C_nls (src/nls.c:54-813).
 *   fn, y, n       : model + response (R: fn, y, env)
 *   jac, fvv       : 1 = analytic derivative requested (R: !is.null(jac) / !is.null(fvv))
 *   start          : p values, or 2 x p column-major [lower, upper] ranges when start_is_matrix
 *   swts           : sqrt(weights) [n] or NULL;  swts_is_matrix = 1 (n x n t(chol(W))) is refused
 *                    with GSLNLS_E_UNSUPPORTED (GLS needs an n x n matrix)
 *   lupars         : 2 x p column-major [lower, upper] with +-Inf, or NULL (src/nls.c:248-263)
 *   control_int    : 15 ints  (R/nls.R:693-709)
 *   control_dbl    : 11 doubles (R/nls.R:710-713, App. C.2)
 *   has_start      : 2 x p logical, multi-start only (src/nls.c:307)
 *   loss_rho, loss_cc : loss_config (R/nls.R:663)
 * Returns the solver status (== out->conv) or a GSLNLS_E_* library error.
 */
int gslnls_nls(const gslnls_model *fn, const double *y, int n, int jac, int fvv, const double *start,
               int start_is_matrix, const double *swts, int swts_is_matrix, const double *lupars,
               const int *control_int, const double *control_dbl, const int *has_start, int loss_rho,
               const double *loss_cc, gslnls_result *out);

/*
 * gslnls_nls_fn -- gsl_nls() on an R `function` (gsl_nls.function, R/nls.R:778; the reference's unit tests 2.2-2.3 and
 * README example 4 call it this way): the model, its Jacobian and its second directional derivative stay HOST callbacks,
 * evaluated on the calling thread exactly where the reference evaluates the closures (gsl_f / gsl_df / gsl_fvv,
 * src/nls.c:815-978); everything after them runs on the device: weighting, ||f||^2, the difference Jacobian's columns
 * (p + 1 calls of f, src/fdjac.c) , J^T J on the matrix cores, J^T f, the damped solve, rho -- csrc/bd_host.hpp.  Any
 * p <= 4096 (the reference's n x p workspace has no limit, src/nls.c:266).
 *   f    : model values m(theta) into fval[n] (NOT residuals: the core subtracts y); non-zero return = EBADFUNC
 *   jac  : n x p column-major dm/dtheta as an R matrix is laid out, or NULL for finite differences (control_int[5])
 *   fvv  : D^2 m[v, v] into out[n], or NULL (lmaccel then differences f, src/fdfvv.c)
 *   start p values; swts sqrt(weights) [n] or NULL; lupars 2 x p or NULL; control_int / control_dbl as gslnls_nls.
 * gslnls_nls_fn_loss: the same with a robust loss (loss_rho, loss_cc as gslnls_nls; 0 = default): the IRLS driver of
 * src/nls_irls.c:412-546 around the solve, residual median and re-weighting on the device.  out->code_path = 4.
 * gslnls_nls_fn_mstart: `start` of C_nls is a 2 x p matrix of ranges (Rf_isMatrix(start), src/nls.c:274-532; the
 * reference's unit tests 4.2.x / 4.3.x): the whole multi-start procedure -- quasi-random points, det filter,
 * concentration fits of mstart_p iterations, local searches, stopping rule, robust second pass, final solve / IRLS -- with
 * the closures as the model (gsl_multistart_driver, src/nls_mstart.c:24-350).  start2p / has_start: 2 x p column-major as
 * gslnls_nls takes them.  The points are fitted one after the other, as the closures can only be evaluated one parameter
 * vector at a time on the calling thread; each fit's n x p and p x p work runs on the device.
 */
typedef int (*gslnls_fn_cb)(const double *theta, int p, double *fval, int n, void *user);
typedef int (*gslnls_jac_cb)(const double *theta, int p, double *J, int n, void *user);
typedef int (*gslnls_fvv_cb)(const double *theta, const double *v, int p, double *out, int n, void *user);
int gslnls_nls_fn(int n, int p, const double *y, gslnls_fn_cb f, gslnls_jac_cb jac, gslnls_fvv_cb fvv, void *user,
                  const double *start, const double *swts, const double *lupars, const int *control_int,
                  const double *control_dbl, gslnls_result *out);
int gslnls_nls_fn_loss(int n, int p, const double *y, gslnls_fn_cb f, gslnls_jac_cb jac, gslnls_fvv_cb fvv, void *user,
                       const double *start, const double *swts, const double *lupars, const int *control_int,
                       const double *control_dbl, int loss_rho, const double *loss_cc, gslnls_result *out);
int gslnls_nls_fn_mstart(int n, int p, const double *y, gslnls_fn_cb f, gslnls_jac_cb jac, gslnls_fvv_cb fvv, void *user,
                         const double *start2p, const int *has_start, const double *swts, const double *lupars,
                         const int *control_int, const double *control_dbl, int loss_rho, const double *loss_cc,
                         gslnls_result *out);

/* trace = TRUE (control_int[1] != 0).  The reference prints while it runs: one line per iteration of the final solve
 * (src/nls.c:980-995), one per accepted stationary point of the multi-start stage and its closing lines
 * (src/nls_mstart.c:331-337, src/nls.c:510-517), one per IRLS iteration (src/nls_irls.c:466-472), the summary block
 * (src/nls.c:610-630); on the large path one line per iteration with |x|^2 and cond(J) and its own summary
 * (src/nls_large.c:259-273, :715-739).  The core collects exactly that text, in that order, during gslnls_nls /
 * gslnls_nls_fn* / gslnls_nls_large / gslnls_large_solve; the binding prints it after the call has returned (Rprintf may
 * long-jump on a user interrupt and must not run below the core's C++ frames).  gslnls_trace_text copies the text of the
 * last call (NUL-terminated, truncated to cap) and returns its full length; an empty text when trace was off.
 * gslnls_format_trace: the iteration lines + summary block alone, from a result that carries partrace / ssrtrace (what
 * the entry points append themselves; exposed so that the format can be checked without a device). */
size_t gslnls_trace_text(char *buf, size_t cap);
/* par_order[j] = the caller's index of the core's j-th parameter (what gslnls_lower_formula returned; NULL = identity):
 * the parameter vectors of the NEXT verbose call are printed in the caller's order.  Consumed by that call. */
int gslnls_trace_set_order(const int *par_order, int p);
size_t gslnls_format_trace(const gslnls_result *res, int n, int p, const int *control_int, int loss_rho, char *buf, size_t cap);

/* Where the wall time of the last gslnls_nls() call of this process went, milliseconds:
 *   ms[0] create (allocation / re-binding of a parked problem), [1] H2D of x, y, swts, [2] the solve loop (plus
 *   multi-start / IRLS driver work), [3] the finalize kernel (resid, grad, covar), [4] D2H of the result vectors,
 *   [5] destroy / parking, [6] total.  Returns the number of values (7) and writes min(cap, 7).  What .Call(C_nls)
 *   delivers is the total, not the resident loop ("also report end-to-end"); the copies the reference
 *   itself makes are src/nls.c:695-720.  (GSLNLS_PREFAULT=1: result vectors of >= 1 MB are faulted in by a helper thread
 *   while the data uploads -- measured slower on the MI355X host, see csrc/dense_host.hpp; off by default.) */
int gslnls_last_call_profile(double *ms, int cap);

/* The same for the last fit on the matrix path (function models, formulas beyond 64 parameters; csrc/bd_host.hpp), 12 values:
 *   [0] set-up (lowering of the formula, kernels from the cache / the in-process compiler, buffers, upload), [1] the solve
 *   loop, of which [2] damped solves (with the fused trial evaluation behind them when [10] = 1), [3] Jacobians with
 *   J^T J and J^T f, [4] residual evaluations outside the fused step; [5] covariance, [6] resid + grad to the host,
 *   [7] condition diagnostic; [8] trial steps, [9] Jacobians, [10] 1 = one host synchronisation per trial step, [11] p. */
int gslnls_last_matrix_path_profile(double *ms, int cap);

/*
 * Model lowering: match the deparsed right-hand side of the model formula (formula[[3]], R/nls.R:565)
 * against the device registry, up to renaming of parameters / data columns and parameter order.
 * Returns the GSLNLS_MODEL_* id (> 0) or 0 when the expression is not registered.
 *   par_order[k]  = index into parnames of the k-th device parameter
 *   varnames_out  = comma-separated data-column names in device regressor order
 */
int gslnls_lower_formula(const char *rhs, int p, const char *const *parnames, int *par_order, char *varnames_out,
                         int varnames_cap);

/* ---- resident-data API: the same solve with the data already in HBM -------------------
 * (what bench.py times: H2D once at create, then repeated solves) */
typedef struct gslnls_dense gslnls_dense;

gslnls_dense *gslnls_dense_create(const gslnls_model *fn, const double *y, int n, const double *swts,
                                  int *err);
void gslnls_dense_destroy(gslnls_dense *h);
/* Destroyed problems of the built-in models that own their data are parked (at most two per model and process) and
 * re-bound by the next create of the same model -- a one-shot gslnls_nls() on a small problem is otherwise dominated by
 * ~3 ms of device allocation around ~0.1 ms of fitting.  gslnls_trim_cache() frees what is parked. */
void gslnls_trim_cache(void);
/* single-start solve (default loss) on resident data.  chunk = 0 (default): the whole fit runs in ONE launch of the
 * resident kernel (csrc/dense_persist.hpp) when the device can hold the grid, else as below.  chunk > 0 selects the
 * launch-per-step kernel with that many step launches enqueued per host check of the device's completion word;
 * chunk < 0 the same kernel with its adaptive chunking: 16, or -- after a fit of the same kind on this handle -- as many
 * launches as that fit needed, then top-ups of 4 (launches enqueued beyond the one that ends a fit still run, as no-ops) */
int gslnls_dense_solve(gslnls_dense *h, int jac, int fvv, const double *start, const double *lupars,
                       const int *control_int, const double *control_dbl, int chunk, gslnls_result *out);
/* time `reps` back-to-back launches of the pass kernel at `theta` with HIP events on the
 * library's stream; returns average milliseconds per launch (negative on error) */
float gslnls_dense_time_pass(gslnls_dense *h, int jac, const double *theta, int reps);
/* Device time of the solve loops run on this handle since the last reset: every fit brackets its step launches with a
 * pair of HIP events on the library's stream (first launch ... last launch of the last chunk, trailing launches
 * included); the pairs sit in a ring and are read back when it wraps or when the totals are asked for, so that no fit
 * waits for its own trailing launches.  ms_total / launches_total may be NULL. */
int gslnls_dense_loop_event_stats(gslnls_dense *h, double *ms_total, long long *launches_total, int reset);
/* swap the weights of a resident problem (IRLS) */
/* Post-fit diagnostics on the resident data: hat values h_i = J_i (J^T J)^-1 J_i^T and Cook's distances
 * e_i^2 / (p s^2) * h_i / (1 - h_i)^2 at `par` (hat_values / cooks_d, src/nls_utils.c:88-150; the reference's S3 methods
 * hatvalues() and cooks.distance() build them from the n x p gradient on the host).  Either output may be NULL. */
int gslnls_dense_diagnostics(gslnls_dense *h, int jac, const double *par, const int *control_int,
                             const double *control_dbl, double *hat, double *cooks);
int gslnls_dense_set_swts(gslnls_dense *h, const double *swts);

/* ---- multi-start (src/nls_mstart.c, src/nls.c:274-532) ----------------------------------------
 * gslnls_nls() with start_is_matrix runs the whole procedure.  The pieces below expose the
 * resident-data form and the sharding hooks for one-process-per-GPU jobs.
 *
 * Per-point record of a batch: K = 3p + 8 doubles
 *   x[p] (where the fit ended), diag[p] (trust-region scaling at the end), x0[p] (sampled point),
 *   chisq0, chisq1, det0, det1, ssr_start, niter, status, nevalf.                                   */
typedef int (*gslnls_allgather_fn)(void *ctx, int per_points, int K);
/* Communicator for sharding the N sample points over `world` ranks: each rank fits a contiguous
 * block of ceil(N/world) points into shard_buf, fn() must gather the equal-sized shards of all
 * ranks into all_buf on every rank (RCCL all-gather over xGMI when the buffers are device
 * memory).  world = 1 (default) disables it. */
int gslnls_set_comm(int rank, int world, gslnls_allgather_fn fn, void *ctx, double *shard_buf, double *all_buf,
                    long long cap_points, int buffers_on_device);
/* The same exchange inside the library: an RCCL communicator owned by libgslnls_hip.so.  The batch kernel writes
 * its shard into the communicator's device buffer and ONE ncclAllGather over xGMI is enqueued right behind it on
 * the library's stream; the host only waits for the gathered records.  RCCL is bound at run time (dlopen; the copy
 * the process already holds, if any), so single-GPU hosts need none.
 * Bootstrap, once per job: rank 0 calls gslnls_comm_get_unique_id (128 bytes, ncclGetUniqueId), the host application
 * hands the id to the other ranks over whatever channel it has (MPI, a socket, a file), every rank calls
 * gslnls_comm_init_rank after gslnls_set_device.  gslnls_comm_init_file does both over a file all ranks can see
 * (rank 0 removes a stale file, writes the new one atomically and removes it again once every rank has joined; the
 * others wait up to timeout_s seconds; GSLNLS_COMM_NONCE in the environment, when set, must match across the job).  Every gslnls_nls() / gslnls_dense_mstart()
 * with start ranges is then sharded; all ranks must make the same calls with the same arguments and get identical
 * results.  Replaces nothing in the reference (its loop over the sample points, src/nls_mstart.c:42-128, is sequential
 * in one process, src/nls.c:372-399). */
#define GSLNLS_COMM_ID_BYTES 128
int gslnls_comm_get_unique_id(char *id128);
int gslnls_comm_init_rank(const char *id128, int rank, int world);
int gslnls_comm_init_file(const char *path, int rank, int world, int timeout_s);
void gslnls_comm_destroy(void);
long long gslnls_comm_allgather_count(void); /* collectives issued so far (tests, benchmark) */
/* Optional HIP-event pair around every ncclAllGather the library issues (off by default: two event records per batch):
 * gslnls_comm_allgather_ms returns the milliseconds summed since set_timing(1) and, in *timed, how many collectives
 * they cover -- the device time of the exchange alone, next to kernel_ms of gslnls_mstart_batch. */
void gslnls_comm_set_timing(int on);
double gslnls_comm_allgather_ms(long long *timed);
const char *gslnls_comm_last_error(void);
/* multi-start + final solve on resident data; start2p = 2 x p column-major ranges */
int gslnls_dense_mstart(gslnls_dense *h, int jac, int fvv, const double *start2p, const double *lupars,
                        const int *control_int, const double *control_dbl, const int *has_start, gslnls_result *out);
/* one concentration batch (src/nls_mstart.c:42-128) of `count` fresh Sobol points with global draw
 * indices first_draw..first_draw+count-1; computes the records of points [lo, hi) into
 * records[0 : (hi-lo)*K) (host or device memory).  kernel_ms: HIP-event time of the batch kernel.
 * lo < 0: the whole batch through the bound communicator -- this rank fits its block of ceil(count/world) points, one
 * all-gather completes the array on every rank, records (host memory, count x K, may be NULL) receives all of it:
 * one concentration stage exactly as gslnls_nls() runs it. */
int gslnls_mstart_batch(gslnls_dense *h, int jac, const double *ranges, const double *kd, long long first_draw,
                        int count, int lo, int hi, int maxiter, double dtol, const int *control_int,
                        const double *control_dbl, const double *lupars, double *records, int records_on_device,
                        float *kernel_ms);
int gslnls_mstart_record_size(int p);

/* ---- batched robust fits (BASELINE config C5) ------------------------------------------------------
 * B independent data sets of n rows each, one model, one start, one loss: every data set goes through
 * exactly the procedure gsl_nls(loss = ...) runs for a single one (src/nls_irls.c:412-546), one
 * workgroup per data set.  No batched form exists in the reference; the single-fit contract is kept per
 * data set.  Layouts: x [B][nx][n], y [B][n], swts [B][n] (sqrt of user weights) or NULL.
 * Outputs per data set d in [lo, hi): par[(d-lo)*p ..], scal[(d-lo)*4 ..] = sigma, weighted ssr,
 * irls_tol, initial ssr; ints[(d-lo)*4 ..] = conv, irls_status, irls_niter, niter of the last solve. */
typedef struct gslnls_batch gslnls_batch;
gslnls_batch *gslnls_batch_create(int model_id, int p, int nx, const double *x, const double *y, const double *swts,
                                  int n, int B, int *err);
void gslnls_batch_destroy(gslnls_batch *h);
int gslnls_batch_irls(gslnls_batch *h, int lo, int hi, int jac, int fvv, const double *start, const double *lupars,
                      const int *control_int, const double *control_dbl, int loss_rho, const double *loss_cc,
                      double *par, double *scal, int *ints, float *kernel_ms);

/* work done by the last gslnls_batch_irls() on this handle, summed over its data sets: passes over the n rows made by
 * the LM solves (one per trial step + one per initial point) and number of re-weightings (each: one pass for the
 * residuals, the median select, one pass for the weights) -- the units of SURVEY.md 8(d)'s C5 byte accounting */
int gslnls_batch_last_passes(gslnls_batch *h, long long *lm_passes, long long *reweightings);
/* One process per GPU (SURVEY.md 8(e), row "batched IRLS"): the B_total data sets are cut into contiguous blocks of
 * ceil(B_total / world) per rank, `h` holds exactly this rank's block (created from its slice of x / y); every rank
 * fits its block with no traffic, then ONE all-gather of (p + 8) doubles per data set completes par / scal / ints
 * (B_total entries each, same layout as above) on every rank -- through the in-library RCCL communicator when one is
 * bound (gslnls_comm_init_*), else through the gslnls_set_comm callback.  With one rank it is gslnls_batch_irls.
 * Every rank must call it, with a handle: a rank whose block is empty (B_total = 9 over 8 ranks: ranks 5..7) passes one
 * created with B = 0 (x = y = NULL).  A rank-local failure (wrong block size, a failed fit, a HIP error) does not return
 * before the collective: it is carried through it and every rank returns a failure afterwards. */
int gslnls_batch_irls_gather(gslnls_batch *h, int B_total, int jac, int fvv, const double *start, const double *lupars,
                             const int *control_int, const double *control_dbl, int loss_rho, const double *loss_cc,
                             double *par, double *scal, int *ints, float *kernel_ms);

/* ---- gsl_nls_large: replaces C_nls_large (src/init.c:16, src/nls_large.c:66-424) --------------
 * 9 SEXP arguments there: fn, y, jac, fvv, env, start, weights, control_int[7], control_dbl[8]
 * (SURVEY.md App. C.3).  jac is always analytic on device (the reference requires one too), fvv /
 * lmaccel and the dogleg family are refused.  control_int[2]: 0 = lm (normal equations), 5 = cgst. */
typedef struct gslnls_large_result
{
    double *par;    /* [p] */
    double *covar;  /* [p*p] column-major (J^T J)^-1, NaN on failure */
    double *resid;  /* [n] weighted residual */
    int niter, conv, info;
    double ssr, ssrtol, chisq_init;
    int neval[4];   /* f, dfu, df2, fvv (src/nls_large.c:395-401) */
    double *partrace, *ssrtrace;
    int n_passes;   /* device passes over the rows issued for this call */
    float last_pass_ms;
} gslnls_large_result;

int gslnls_nls_large(const gslnls_model *fn, const double *y, int n, const double *start, const double *weights,
                     const int *control_int, const double *control_dbl, gslnls_large_result *out);

typedef struct gslnls_large gslnls_large;
gslnls_large *gslnls_large_create(const gslnls_model *fn, const double *y, int n, const double *weights, int *err);
void gslnls_large_destroy(gslnls_large *h);
int gslnls_large_solve(gslnls_large *h, const double *start, const int *control_int, const double *control_dbl,
                       gslnls_large_result *out);
/* average milliseconds of `reps` passes: mode 0 = EVAL pass at x, 1 = fused J^T J u pass (HIP events),
 * 2 = full J^T J (p = 64: MFMA kernel + reduction + readback, host clock) */
float gslnls_large_time_pass(gslnls_large *h, int mode, const double *x, const double *u, int reps);

/* ---- gsl_nls_large with a sparse Jacobian supplied by the caller ---------------------------------------
 * The model and its Jacobian stay host callbacks, as the R closures fn / jac are in the reference
 * (gsl_f_large, gsl_df_large: src/nls_large.c:426-653); J arrives as a Matrix-package style sparse matrix
 * (dgRMatrix / dgCMatrix / dgTMatrix, match_dg_class src/nls_large.c:16-49).  Everything after the callbacks
 * -- weighting, ssr, J^T f, diag(J^T J), the CG products J^T (J u), dense J^T J for `lm`, covariance -- runs
 * on the device (csrc/sparse_large.hpp).  f is called at every trial point, jac once per accepted point. */
#define GSLNLS_SPARSE_CSR 0 /* dgRMatrix: p = row pointers [nrow+1], j = column indices [nnz] */
#define GSLNLS_SPARSE_CSC 1 /* dgCMatrix: p = column pointers [ncol+1], i = row indices [nnz] */
#define GSLNLS_SPARSE_COO 2 /* dgTMatrix: i, j triplets [nnz]; duplicates are summed */
#define GSLNLS_SPARSE_DENSE 3 /* base matrix / dgeMatrix (jacclass -2 / -1 of the reference, src/nls_large.c:504, :625-633):
                                 x = the n x p block column-major as R holds it, nnz = n p; p, i, j unused.  No index arrays
                                 exist anywhere: J u and J^T w are dense matrix-vector kernels, J^T J the matrix cores' SYRK */
typedef struct gslnls_sparse
{
    int format, nrow, ncol;
    long nnz;
    const int *p, *i, *j; /* 0-based, as the Matrix package stores them */
    const double *x;      /* [nnz] */
} gslnls_sparse;
/* model values m(theta) (NOT residuals; the core subtracts y) into fval[n]; non-zero return aborts the fit */
typedef int (*gslnls_large_f_cb)(const double *theta, int p, double *fval, int n, void *user);
/* fill *J with pointers to host arrays that stay valid until the next call of the same callback */
typedef int (*gslnls_large_jac_cb)(const double *theta, int p, gslnls_sparse *J, void *user);
gslnls_large *gslnls_large_create_sparse(int n, int p, const double *y, const double *weights, gslnls_large_f_cb f,
                                         gslnls_large_jac_cb jac, void *user, int *err);

/* Long device loops poll this hook between launch chunks (dense), passes (large) and multi-start batches; a
 * non-zero return abandons the fit with GSLNLS_E_INTERRUPTED.  The R shim installs a wrapper around
 * R_CheckUserInterrupt (the reference relies on R_ExecWithCleanup, src/nls.c:61, NEWS.md 1.1.1). */
void gslnls_set_interrupt_hook(int (*check)(void));

/* ---- test hooks ------------------------------------------------------------------------- */
/* the damped solve (J^T J + mu D^2) sol = rhs of the wide path (p <= 64, packed lower triangle) on the device: GSL's
 * pivoted modified Cholesky (gsl_linalg_mcholesky_*; multifit_nlinear/cholesky.c) spread over one wavefront */
int gslnls_debug_wide_solve(int p, const double *Ap, const double *diag, double mu, const double *rhs, double *sol);
/* (A + mu diag(d)^2) sol = rhs for the gsl_nls_large(algorithm = "lm") step on the device (A: p x p symmetric,
 * row-major, p <= 4096; d may be NULL): the same pivoted modified Cholesky, panels of pivot steps by one workgroup and
 * grid-wide trailing updates (csrc/mchol_device.hip; gsl_multilarge_nlinear's lm step, multilarge_nlinear/cholesky.c) */
int gslnls_debug_mchol_solve(int p, const double *A, const double *diag, double mu, const double *rhs, double *sol);
/* device memory through the HIP runtime this library is linked against (test / measurement hooks: a process may hold a
 * second copy of the runtime, e.g. PyTorch's, whose allocations belong to another context); to_device: 1 host -> device,
 * 0 device -> host */
int gslnls_debug_device_alloc(void **p, size_t bytes);
int gslnls_debug_device_free(void *p);
int gslnls_debug_device_copy(void *dst, const void *src, size_t bytes, int to_device);
/* device milliseconds of one J^T J of the matrix path (bd_syrk_kernel + its reduction) on an n x p matrix of noise, HIP
 * events over `reps` repetitions; < 0 on error */
double gslnls_debug_bd_syrk_ms(int n, int p, int reps);
/* device milliseconds of the last natural-order solve (HIP events around its kernels), < 0 when not available.  The
 * events are recorded only between gslnls_debug_mchol_timing(1) and gslnls_debug_mchol_timing(0): they cost a solve ~6 us
 * (a marker packet in front of a caller's tail), so the product path runs without them */
double gslnls_debug_mchol_last_device_ms(void);
void gslnls_debug_mchol_timing(int on);
/* the same with J^T J resident in device memory (jtj_dev: p x p doubles, row-major, left as it is): the call of the lm step */
int gslnls_debug_mchol_solve_resident(int p, const double *jtj_dev, const double *diag, double mu, const double *rhs,
                                      double *sol);
/* the same solve by the host routine that serves the lm step below the device threshold (no device needed) */
int gslnls_debug_host_mchol_solve(int p, const double *A, const double *diag, double mu, const double *rhs, double *sol);
/* the sums of one pass over the rows of a wide problem (p > 9) at theta: totals[0] = ssr, [1] = non-finite flag,
 * then the packed lower triangle of J^T J (p (p + 1) / 2, row by row) and J^T f (p); jac: 1 analytic, 0 finite
 * differences (fdtype: 0 forward, 1 central) */
int gslnls_debug_wide_sums(gslnls_dense *h, int jac, int fdtype, const double *theta, double *totals);

/* ---- introspection ---------------------------------------------------------------------- */
const char *gslnls_strerror(int code);     /* gsl_strerror strings, App. C.4 */
const char *gslnls_algorithm_name(int trs); /* gsl_multifit_nlinear_trs_name */
int gslnls_device_count(void);
int gslnls_set_device(int ordinal);        /* hipSetDevice for this process (one process per GPU) */
const char *gslnls_version(void);

#ifdef __cplusplus
}
#endif
#endif
