// vm_models.hip -- the device paths instantiated for expression models ModelVM<P>.  Compiled once per
// parameter count (-DGSLNLS_VM_P=1..9 -> _obj/vm_p<k>.o, in parallel) and once without the macro for the
// dispatcher that compiles the expression and picks the instantiation.
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include "../../include/gslnls_core.h"
#include "dense_host.hpp"
#include "mstart_host.hpp"
#include "irls_host.hpp"
#include "robust_host.hpp"
#include "large_host.hpp"
#include "expr_compile.hpp"
#include "vm_model.hpp"
#include "rtc_host.hpp"

namespace gslnls
{

#ifdef GSLNLS_VM_P

__constant__ VmProgram c_vm_prog; // one copy per translation unit (= per parameter count)

template <int P>
struct VmDenseFit : DenseFit<ModelVM<P>>
{
    using Base = DenseFit<ModelVM<P>>;
    VmProgram prog;
    // native code of this program (rtc_host.hpp): one compiled unit per Jacobian kind, requested when a fit of that
    // kind comes along -- in the background under GSLNLS_LOWER_AUTO (this fit runs on the interpreter, a later one
    // finds the kernels), at once under GSLNLS_LOWER_JIT
    int lowering = GSLNLS_LOWER_AUTO, nx_model = 1;
    std::string rtc_src;
    std::shared_ptr<RtcEntry> rtc[3];
    void set_program(const VmProgram &pr, int lower, int nxm)
    {
        prog = pr;
        lowering = lower;
        nx_model = nxm;
        rtc_src.clear();
        for (int k = 0; k < 3; ++k)
        {
            rtc[k].reset();
            this->native_step[k] = this->native_finalize[k] = nullptr;
        }
    }
    // 0 or GSLNLS_E_UNSUPPORTED (native code was demanded and cannot be had)
    int bind_native(int jac, const int *ci)
    {
        const int jm = jac ? 0 : (ci[5] ? 2 : 1);
        this->native_step[jm] = this->native_finalize[jm] = nullptr;
        if (lowering == GSLNLS_LOWER_VM)
            return 0;
        if (rtc_src.empty())
            rtc_src = rtc_dense_source(prog, nx_model);
        const std::string es = rtc_step_expr(jm, Base::T), ef = rtc_finalize_expr(jm, Base::T);
        const bool wait = lowering == GSLNLS_LOWER_JIT;
        if (!rtc[jm] || rtc[jm]->state.load() == RTC_NONE || (wait && rtc[jm]->state.load() != RTC_READY && rtc[jm]->state.load() != RTC_FAILED))
            rtc[jm] = rtc_request(rtc_src, {es, ef}, wait);
        RtcEntry &e = *rtc[jm];
        const int st = e.state.load(std::memory_order_acquire);
        std::string msg;
        if (st == RTC_READY)
        {
            this->native_step[jm] = rtc_function(e, es, msg);
            this->native_finalize[jm] = this->native_step[jm] ? rtc_function(e, ef, msg) : nullptr;
            if (!this->native_finalize[jm])
                this->native_step[jm] = nullptr;
        }
        else if (st == RTC_FAILED)
            msg = e.log;
        if (wait && !this->native_step[jm])
        {
            fprintf(stderr, "gslnls: native lowering failed: %s\n", msg.c_str());
            return GSLNLS_E_UNSUPPORTED;
        }
        return 0;
    }
    int upload()
    {
        GSLNLS_HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(c_vm_prog), &prog, sizeof(VmProgram)));
        // slot file of the step kernel in LDS when it fits (vm_model.hpp); GSLNLS_VM_LDS=0 keeps it in scratch memory
        const int ops = prog.nfvv > prog.nops ? prog.nfvv : prog.nops;
        const int slots = 2 * prog.p + prog.nx + prog.nconst + ops;
        const char *e = getenv("GSLNLS_VM_LDS");
        this->vm_lds_slots = (e && e[0] == '0') ? 0 : (slots <= Base::VM_LDS_CAP ? slots : 0);
        return 0;
    }
    int solve(int jac, int fvv, const double *start, const double *lupars, const int *ci, const double *cd, int chunk,
              gslnls_result *out) override
    {
        if (fvv && prog.nfvv == 0)
            return GSLNLS_E_UNSUPPORTED; // second derivatives too large for the program: use fvv = FALSE (finite differences)
        if (upload())
            return GSLNLS_E_NODEVICE;
        if (const int rc = bind_native(jac, ci))
            return rc;
        return Base::solve(jac, fvv, start, lupars, ci, cd, chunk, out);
    }
    int irls(int jac, int fvv, const double *start, const double *lupars, const int *ci, const double *cd, int loss_rho,
             const double *loss_cc, gslnls_result *out) override
    {
        if (fvv && prog.nfvv == 0)
            return GSLNLS_E_UNSUPPORTED;
        if (upload())
            return GSLNLS_E_NODEVICE;
        if (const int rc = bind_native(jac, ci))
            return rc;
        return Base::irls(jac, fvv, start, lupars, ci, cd, loss_rho, loss_cc, out);
    }
    int mstart(int jac, int fvv, const double *start2p, const double *lupars, const int *ci, const double *cd,
               const int *has_start, const MsComm &comm, int loss_rho, const double *loss_cc, gslnls_result *out) override
    {
        if (fvv && prog.nfvv == 0)
            return GSLNLS_E_UNSUPPORTED;
        if (upload())
            return GSLNLS_E_NODEVICE;
        return Base::mstart(jac, fvv, start2p, lupars, ci, cd, has_start, comm, loss_rho, loss_cc, out);
    }
    int prepare_device() override { return upload(); }
    float time_pass(int jac, const double *theta, int reps) override
    {
        upload();
        const int ci[15] = {0};
        if (bind_native(jac, ci))
            return -1.f;
        return Base::time_pass(jac, theta, reps);
    }
    int diagnostics(int jac, const double *theta, const int *ci, const double *cd, double *hat, double *cooks) override
    {
        if (upload())
            return GSLNLS_E_NODEVICE;
        return Base::diagnostics(jac, theta, ci, cd, hat, cooks);
    }
};

template <int P>
static DenseBase *make_vm_impl(const VmProgram &prog, int lowering, const gslnls_model *fn, const double *y, int n,
                               const double *swts, int *err)
{
    // interpreted expression problems are containers for any program with P parameters: parked ones are re-bound
    // (every DenseFit<ModelVM<P>> that exists is a VmDenseFit<P>: they are only created here)
    auto *d = fn->x_on_device ? nullptr : static_cast<VmDenseFit<P> *>(DenseFit<ModelVM<P>>::acquire());
    if (!d)
        d = new VmDenseFit<P>();
    d->set_program(prog, lowering, fn->nx > 0 ? fn->nx : 1);
    // the expression always sees VM_NX regressor columns; pad the missing ones with zeros
    gslnls_model padded = *fn;
    std::vector<double> xpad;
    if (!fn->x_on_device && fn->nx < VM_NX)
    {
        xpad.assign((size_t)n * VM_NX, 0.0);
        for (int c = 0; c < fn->nx; ++c)
            for (int i = 0; i < n; ++i)
                xpad[(size_t)c * n + i] = fn->x[(size_t)c * n + i];
        padded.x = xpad.data();
        padded.nx = VM_NX;
    }
    else if (fn->nx != VM_NX)
    {
        *err = GSLNLS_E_UNSUPPORTED; // device-resident data must already carry VM_NX columns
        delete d;
        return nullptr;
    }
    *err = d->init(&padded, y, n, swts);
    if (*err != GSLNLS_SUCCESS)
    {
        delete d;
        return nullptr;
    }
    return d;
}

#define GSLNLS_CAT2(a, b) a##b
#define GSLNLS_CAT(a, b) GSLNLS_CAT2(a, b)
DenseBase *GSLNLS_CAT(make_vm_p, GSLNLS_VM_P)(const VmProgram &prog, int lowering, const gslnls_model *fn, const double *y,
                                              int n, const double *swts, int *err)
{
    return make_vm_impl<GSLNLS_VM_P>(prog, lowering, fn, y, n, swts, err);
}

void GSLNLS_CAT(trim_vm_p, GSLNLS_VM_P)() { DenseFit<ModelVM<GSLNLS_VM_P>>::trim_pool(); }

#else // dispatcher

#define GSLNLS_VM_DECL(k)                                                                                              \
    DenseBase *make_vm_p##k(const VmProgram &, int, const gslnls_model *, const double *, int, const double *, int *);
GSLNLS_VM_DECL(1) GSLNLS_VM_DECL(2) GSLNLS_VM_DECL(3) GSLNLS_VM_DECL(4) GSLNLS_VM_DECL(5) GSLNLS_VM_DECL(6)
GSLNLS_VM_DECL(7) GSLNLS_VM_DECL(8) GSLNLS_VM_DECL(9)

#define GSLNLS_VM_TRIM_DECL(k) void trim_vm_p##k();
GSLNLS_VM_TRIM_DECL(1) GSLNLS_VM_TRIM_DECL(2) GSLNLS_VM_TRIM_DECL(3) GSLNLS_VM_TRIM_DECL(4) GSLNLS_VM_TRIM_DECL(5)
GSLNLS_VM_TRIM_DECL(6) GSLNLS_VM_TRIM_DECL(7) GSLNLS_VM_TRIM_DECL(8) GSLNLS_VM_TRIM_DECL(9)
void trim_dense_expr()
{
    trim_vm_p1(); trim_vm_p2(); trim_vm_p3(); trim_vm_p4(); trim_vm_p5(); trim_vm_p6(); trim_vm_p7(); trim_vm_p8(); trim_vm_p9();
}

DenseBase *make_dense_wide(const gslnls_model *fn, const double *y, int n, const double *swts, int *err); // wide_models.hip

DenseBase *make_dense_expr(const gslnls_model *fn, const double *y, int n, const double *swts, int *err)
{
    if (!fn->expr || !fn->parnames || (fn->nx > 0 && !fn->xnames) || fn->nx > WIDE_NX || fn->p < 1 || fn->p > WIDE_MAX_P)
    {
        *err = (fn->nx > WIDE_NX || fn->p > WIDE_MAX_P) ? GSLNLS_E_UNSUPPORTED : GSLNLS_EINVAL;
        return nullptr;
    }
    // more parameters than the register-resident state machine holds, or more data columns than the interpreter's rows
    // carry: the wide path (J^T J on the matrix cores, kernels compiled in process)
    if (fn->p > 9 || fn->nx > VM_NX)
    {
        const char *ev = getenv("GSLNLS_LOWERING");
        if (fn->lowering == GSLNLS_LOWER_VM || (ev && !strcmp(ev, "vm")))
        {
            *err = GSLNLS_E_UNSUPPORTED; // the wide path has no interpreted form
            return nullptr;
        }
        return make_dense_wide(fn, y, n, swts, err);
    }
    std::vector<std::string> pn(fn->parnames, fn->parnames + fn->p), vn(fn->xnames, fn->xnames + fn->nx);
    VmProgram prog;
    const std::string e = compile_expression(fn->expr, pn, vn, prog);
    if (!e.empty())
    {
        fprintf(stderr, "gslnls: cannot lower model expression: %s\n", e.c_str());
        *err = GSLNLS_E_UNSUPPORTED;
        return nullptr;
    }
    // lowering: 0 auto (the interpreter now, native code -- compiled in process on a background thread -- for the fits
    // that come after the build), 1 interpreter only, 2 native code now.  GSLNLS_LOWERING=vm|jit overrides.
    int mode = fn->lowering;
    if (const char *e = getenv("GSLNLS_LOWERING"))
        mode = !strcmp(e, "jit") ? GSLNLS_LOWER_JIT : (!strcmp(e, "vm") ? GSLNLS_LOWER_VM : mode);
    if (fn->x_on_device && mode == GSLNLS_LOWER_AUTO)
        mode = GSLNLS_LOWER_VM;
    switch (fn->p)
    {
    case 1: return make_vm_p1(prog, mode, fn, y, n, swts, err);
    case 2: return make_vm_p2(prog, mode, fn, y, n, swts, err);
    case 3: return make_vm_p3(prog, mode, fn, y, n, swts, err);
    case 4: return make_vm_p4(prog, mode, fn, y, n, swts, err);
    case 5: return make_vm_p5(prog, mode, fn, y, n, swts, err);
    case 6: return make_vm_p6(prog, mode, fn, y, n, swts, err);
    case 7: return make_vm_p7(prog, mode, fn, y, n, swts, err);
    case 8: return make_vm_p8(prog, mode, fn, y, n, swts, err);
    case 9: return make_vm_p9(prog, mode, fn, y, n, swts, err);
    default:
        *err = GSLNLS_E_UNSUPPORTED;
        return nullptr;
    }
}

#endif

} // namespace gslnls

#ifndef GSLNLS_VM_P
// ahead-of-time build of the native code of an expression (no device needed: the in-process compiler targets
// gfx950 whatever the host): the step and finalize kernels for the analytic and the forward-difference Jacobian go
// into the cache, where the first fit of any later process finds them
namespace gslnls
{
int wide_expr_build(const gslnls_model *fn, std::string &first_path); // wide_models.hip
}
extern "C" int gslnls_expr_build(const gslnls_model *fn, char *path_out, int path_cap)
{
    using namespace gslnls;
    if (!fn || fn->id != GSLNLS_MODEL_EXPR || !fn->expr || !fn->parnames || (fn->nx > 0 && !fn->xnames) || fn->nx > WIDE_NX ||
        fn->p < 1 || fn->p > WIDE_MAX_P)
        return GSLNLS_EINVAL;
    if (fn->p > 9 || fn->nx > VM_NX)
    {
        std::string path;
        const int rc = wide_expr_build(fn, path);
        if (rc == GSLNLS_SUCCESS && path_out && path_cap > 0)
            snprintf(path_out, (size_t)path_cap, "%s", path.c_str());
        return rc;
    }
    std::vector<std::string> pn(fn->parnames, fn->parnames + fn->p), vn(fn->xnames, fn->xnames + fn->nx);
    VmProgram prog;
    const std::string e = compile_expression(fn->expr, pn, vn, prog);
    if (!e.empty())
    {
        fprintf(stderr, "gslnls: cannot lower model expression: %s\n", e.c_str());
        return GSLNLS_E_UNSUPPORTED;
    }
    const std::string src = rtc_dense_source(prog, fn->nx > 0 ? fn->nx : 1);
    const int NV = 2 + fn->p * (fn->p + 1) / 2 + fn->p;
    const int T = (NV <= 24) ? 512 : (NV <= 70 ? 256 : 128); // DenseFit<M>::T
    std::string first;
    for (int jm = 0; jm < 2; ++jm)
    {
        auto ent = rtc_request(src, {rtc_step_expr(jm, T), rtc_finalize_expr(jm, T)}, true);
        if (ent->state.load() != RTC_READY)
        {
            fprintf(stderr, "gslnls: native lowering failed: %s\n", ent->log.c_str());
            return GSLNLS_E_UNSUPPORTED;
        }
        if (jm == 0)
            first = ent->cache_path;
    }
    if (path_out && path_cap > 0)
        snprintf(path_out, (size_t)path_cap, "%s", first.c_str());
    return GSLNLS_SUCCESS;
}

// state of the native code of an expression model for one Jacobian kind (jac: 1 analytic, 0 forward differences):
// 0 not requested yet, 1 being built, 2 ready (in memory or in the cache), -1 the build failed
extern "C" int gslnls_expr_native_state(const gslnls_model *fn, int jac)
{
    using namespace gslnls;
    if (!fn || fn->id != GSLNLS_MODEL_EXPR || !fn->expr || !fn->parnames || (fn->nx > 0 && !fn->xnames) || fn->nx > VM_NX ||
        fn->p > 9)
        return RTC_FAILED;
    std::vector<std::string> pn(fn->parnames, fn->parnames + fn->p), vn(fn->xnames, fn->xnames + fn->nx);
    VmProgram prog;
    if (!compile_expression(fn->expr, pn, vn, prog).empty())
        return RTC_FAILED;
    const std::string src = rtc_dense_source(prog, fn->nx > 0 ? fn->nx : 1);
    const int NV = 2 + fn->p * (fn->p + 1) / 2 + fn->p;
    const int T = (NV <= 24) ? 512 : (NV <= 70 ? 256 : 128);
    const int jm = jac ? 0 : 1;
    const int st = rtc_request_peek(src, {rtc_step_expr(jm, T), rtc_finalize_expr(jm, T)});
    return st == RTC_QUEUED ? RTC_BUILDING : st;
}

// stop the background compiler before the process exits (rtc_host.hpp, rtc_at_exit)
extern "C" void gslnls_shutdown(void) { gslnls::rtc_at_exit(); }

// start building the native code of an expression model on a background thread and return at once (what the first
// GSLNLS_LOWER_AUTO fit does on its own; a front end can call this as soon as it has parsed the formula).  Returns the
// state as gslnls_expr_native_state does.
extern "C" int gslnls_expr_prefetch(const gslnls_model *fn, int jac)
{
    using namespace gslnls;
    if (!fn || fn->id != GSLNLS_MODEL_EXPR || !fn->expr || !fn->parnames || (fn->nx > 0 && !fn->xnames) || fn->nx > VM_NX ||
        fn->p > 9)
        return RTC_FAILED;
    std::vector<std::string> pn(fn->parnames, fn->parnames + fn->p), vn(fn->xnames, fn->xnames + fn->nx);
    VmProgram prog;
    if (!compile_expression(fn->expr, pn, vn, prog).empty())
        return RTC_FAILED;
    const std::string src = rtc_dense_source(prog, fn->nx > 0 ? fn->nx : 1);
    const int NV = 2 + fn->p * (fn->p + 1) / 2 + fn->p;
    const int T = (NV <= 24) ? 512 : (NV <= 70 ? 256 : 128);
    const int jm = jac ? 0 : 1;
    const int st = rtc_request(src, {rtc_step_expr(jm, T), rtc_finalize_expr(jm, T)}, false)->state.load();
    return st == RTC_QUEUED ? RTC_BUILDING : st;
}
#endif
