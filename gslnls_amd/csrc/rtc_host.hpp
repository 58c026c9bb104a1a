// rtc_host.hpp -- expression models at native speed, compiled IN PROCESS.
//
// The reference needs nothing at run time to evaluate a new formula: R builds the closure and its derivative
// (stats::deriv) on the spot (R/nls.R:565,588-599).  Here the compiled program (vm_program.hpp) is printed as a
// straight-line C++ row model and handed, together with the SAME kernel templates the library itself is built from
// (dense_kernels.hpp ... embedded in the shared object as text, rtc_embed.cpp), to hiprtc -- the compiler that ships
// with the HIP runtime (libhiprtc / comgr).  No hipcc, no source tree, no child process on the deployment box.  The
// resulting code object is cached on disk by content hash, loaded with hipModuleLoadData, and its kernels are
// launched with hipModuleLaunchKernel by the host classes that otherwise launch the interpreter's kernels: same
// signatures, same state layout, same arithmetic in the same order -- the interpreted and the native fit of one
// formula agree bit for bit (tests/test_gpu_expr.py).
//
// GSLNLS_LOWER_AUTO: the first fit of a formula starts the build on a background thread and runs on the
// interpreter; calls that come after the build has finished bind the native kernels.  GSLNLS_LOWER_JIT builds
// synchronously (about 1-3 s once per formula and Jacobian kind).  hiprtc is bound with dlopen at first use, so a
// host without it still loads the library (and is served by the interpreter; the wide path, p > 9, needs it).
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "../../include/gslnls_core.h"
#include "vm_program.hpp"

namespace gslnls
{

// ---------------------------------------------------------------------------------------------------------
// text of the device headers, linked into the library by rtc_embed.cpp (.incbin)
struct RtcHeader
{
    const char *name;
    const char *begin, *end;
};
const RtcHeader *rtc_embedded_headers(int *count); // rtc_embed.cpp

inline std::string rtc_fmt_double(double v)
{
    char buf[64];
    if (v != v)
        return "NAN";
    if (v == INFINITY)
        return "INFINITY";
    if (v == -INFINITY)
        return "(-INFINITY)";
    snprintf(buf, sizeof buf, "%.17g", v);
    std::string s(buf);
    if (s.find_first_of(".eEn") == std::string::npos)
        s += ".0";
    return v < 0 ? "(" + s + ")" : s;
}

// C++ text of instructions [0, upto) of the program; `sink` (value + gradient form of the wide path): gradient k is
// handed to out.set(k, value) as soon as it exists instead of living in an array of P registers
template <class Prog>
inline std::string rtc_emit_ops(const Prog &pr, int upto, bool sink = false)
{
    const int base = 2 * pr.p + pr.nx + pr.nconst;
    auto ref = [&](int slot) -> std::string {
        if (slot < pr.p)
            return "th[" + std::to_string(slot) + "]";
        if (slot < pr.p + pr.nx)
            return "xr[" + std::to_string(slot - pr.p) + "]";
        if (slot < 2 * pr.p + pr.nx)
            return "dir[" + std::to_string(slot - pr.p - pr.nx) + "]";
        if (slot < base)
            return rtc_fmt_double(pr.consts[slot - 2 * pr.p - pr.nx]);
        return "v" + std::to_string(slot - base);
    };
    static const char *fn1[] = {"", "", "", "", "", "", "", "gexp", "log", "sin", "cos", "tan", "atan", "sqrt", "fabs", "tanh", "",
                                "sinh", "cosh", "asin", "acos", "log1p", "expm1", "", "tgamma", "lgamma", "", "", "", "", ""};
    std::string s;
    // The order of the statements.  Plain form: program order (the value's closure first, then the gradient's).  Sink form
    // (round 5): the compiler schedules straight-line code for latency and, handed the whole value first, keeps every term's
    // intermediates alive until the gradient entries that reuse them come by -- at p = 32 (ten Gaussians) the step kernel ran
    // with 142 spilled registers and 464 B of scratch per lane, 245 MB of scratch writes per pass over 1e6 rows (PMC:
    // profiles/r05_wide_pmc.json).  Here the statements are emitted gradient entry by gradient entry (depth first from each
    // entry's root, the value's own sum last), every entry handed to the sink the moment it exists, and a scheduling barrier
    // after every few entries keeps the compiler from pulling the next terms' chains in front: a few terms in flight, not
    // all of them.  Same statements, same operands: the arithmetic does not change.  GSLNLS_RTC_SINK_GROUP: entries per
    // group (0 = program order, no barriers).
    std::vector<int> order;
    std::vector<std::vector<int>> sets_of(upto);
    std::vector<char> barrier_after(upto, 0);
    int sink_group = 6;
    if (const char *e = getenv("GSLNLS_RTC_SINK_GROUP"))
        sink_group = atoi(e);
    if (sink && sink_group > 0)
    {
        std::vector<char> seen(upto, 0);
        std::vector<int> stack;
        auto visit = [&](int root) {
            if (root < base || root >= base + upto)
                return;
            stack.push_back(root - base);
            while (!stack.empty())
            {
                const int i = stack.back();
                if (seen[i] == 2)
                {
                    stack.pop_back();
                    continue;
                }
                if (seen[i] == 0)
                {
                    seen[i] = 1;
                    for (int opnd : {(int)pr.a[i], (int)pr.b[i]})
                        if (opnd >= base && opnd < base + upto && seen[opnd - base] == 0)
                            stack.push_back(opnd - base);
                }
                else
                {
                    seen[i] = 2;
                    order.push_back(i);
                    stack.pop_back();
                }
            }
        };
        int in_group = 0;
        for (int k = 0; k < pr.p; ++k)
        {
            const int root = pr.grad_slot[k];
            if (root < base || root >= base + upto)
                continue; // (not the result of an instruction -- a parameter, a column, a constant: handed out at the end)
            visit(root);
            sets_of[root - base].push_back(k); // (handed out behind the statement that produces it: it is in `order` by now)
            if (++in_group >= sink_group && !order.empty())
            {
                barrier_after[order.back()] = 1;
                in_group = 0;
            }
        }
        visit(pr.value_slot);
        for (int i = 0; i < upto; ++i) // (anything the roots do not reach: dead code of the program, kept for completeness)
            if (seen[i] == 0)
                visit(base + i);
    }
    else
        for (int i = 0; i < upto; ++i)
            order.push_back(i);
    const bool scheduled = sink && sink_group > 0;
    for (int i : order)
    {
        const std::string a = ref(pr.a[i]), b = ref(pr.b[i]);
        std::string e;
        switch (pr.op[i])
        {
        case VM_ADD: e = a + " + " + b; break;
        case VM_SUB: e = a + " - " + b; break;
        case VM_MUL: e = a + " * " + b; break;
        case VM_DIV: e = a + " / " + b; break;
        case VM_NEG: e = "-" + a; break;
        case VM_POW: e = "pow(" + a + ", " + b + ")"; break;
        case VM_SIGN: e = "(" + a + " > 0.0 ? 1.0 : (" + a + " < 0.0 ? -1.0 : 0.0))"; break;
        case VM_PNORM: e = "0.5 * erfc(-" + a + " * 0.70710678118654752440)"; break;
        case VM_PSI0: case VM_PSI1: case VM_PSI2: case VM_PSI3: case VM_PSI4:
            e = "gpsigamma(" + a + ", " + std::to_string((int)pr.op[i] - (int)VM_PSI0) + ")";
            break;
        default: e = std::string(fn1[pr.op[i]]) + "(" + a + ")"; break;
        }
        s += "        const double v" + std::to_string(i) + " = " + e + ";\n";
        if (scheduled)
        {
            for (int k : sets_of[i])
                s += "        out.set(" + std::to_string(k) + ", v" + std::to_string(i) + ");\n";
            if (barrier_after[i])
                s += "        __builtin_amdgcn_sched_barrier(0);\n";
        }
        else if (sink)
            for (int k = 0; k < pr.p; ++k)
                if (pr.grad_slot[k] == base + i)
                    s += "        out.set(" + std::to_string(k) + ", v" + std::to_string(i) + ");\n";
    }
    if (sink) // (gradient entries that are not the result of an instruction: a parameter, a data column, a constant)
        for (int k = 0; k < pr.p; ++k)
            if (pr.grad_slot[k] < base || pr.grad_slot[k] >= base + upto)
                s += "        out.set(" + std::to_string(k) + ", " + ref(pr.grad_slot[k]) + ");\n";
    return s;
}

// ---- round 5: what depends on the parameters alone is computed ONCE, not per row -----------------------------------------
// A row of a sum of ten Gaussians a exp(-((x - m) / s)^2) cost ~980 vector instructions (PMC, profiles/r05_wide_pmc*.json), most of
// them the fp64 divisions by s, s^2, s^3 of the value and its gradient -- ~30 instructions each, on quantities that are the
// same for every row.  The value + gradient closure handed to a sink is therefore split:
//   prologue(th, pre)                 every instruction whose operands are parameters and constants only, plus 1 / d for
//                                     every such d a row divides by; run once per wavefront, results in LDS;
//   value_grad_sink_pre(th, pre, ..)  the rest, per row: a parameter-only result is read back from LDS (a broadcast, like
//                                     theta itself), a division by a parameter-only d is a multiplication by its reciprocal.
// x / d becomes x * (1 / d): one more rounding, at most an ulp per division -- this closure has no interpreted twin to
// agree with bit for bit (the wide and the matrix path are always native), its results stay within the tolerances of
// their tests against the oracle, and the other closures (value for difference Jacobians, fvv) are emitted as before.
template <class Prog>
struct RtcHoist
{
    int base = 0, nops = 0;
    std::vector<char> po;          // instruction is parameter-only
    std::vector<int> export_of;    // instruction -> index in pre[], or -1
    std::vector<int> recip_of;     // slot (any kind) -> index in pre[] of its reciprocal, or -1
    std::vector<int> recip_slots;  // the slots whose reciprocal is exported, in export order
    int npre = 0;
    void build(const Prog &pr)
    {
        base = 2 * pr.p + pr.nx + pr.nconst;
        nops = pr.nops;
        auto uniform = [&](int slot) {
            if (slot < pr.p)
                return true; // a parameter
            if (slot < 2 * pr.p + pr.nx)
                return false; // a data column, a component of the direction
            if (slot < base)
                return true; // a constant
            return slot - base < nops && po[slot - base] != 0;
        };
        po.assign(nops, 0);
        for (int i = 0; i < nops; ++i)
            po[i] = uniform(pr.a[i]) && uniform(pr.b[i]);
        export_of.assign(nops, -1);
        recip_of.assign(base + nops, -1);
        auto want = [&](int slot) {
            if (slot >= base && slot - base < nops && po[slot - base] && export_of[slot - base] < 0)
                export_of[slot - base] = npre++;
        };
        for (int i = 0; i < nops; ++i)
        {
            if (po[i])
                continue;
            if (pr.op[i] == VM_DIV && uniform(pr.b[i]))
            {
                if (recip_of[pr.b[i]] < 0)
                {
                    recip_of[pr.b[i]] = npre++;
                    recip_slots.push_back(pr.b[i]);
                }
                want(pr.a[i]);
                continue;
            }
            want(pr.a[i]);
            want(pr.b[i]);
        }
        for (int k = 0; k < pr.p; ++k)
            want(pr.grad_slot[k]);
        want(pr.value_slot);
    }
};

template <class Prog>
inline std::string rtc_emit_hoisted(const Prog &pr)
{
    RtcHoist<Prog> H;
    H.build(pr);
    const int base = H.base, nops = H.nops;
    static const char *fn1[] = {"", "", "", "", "", "", "", "gexp", "log", "sin", "cos", "tan", "atan", "sqrt", "fabs", "tanh", "",
                                "sinh", "cosh", "asin", "acos", "log1p", "expm1", "", "tgamma", "lgamma", "", "", "", "", ""};
    auto expr_of = [&](int i, const std::string &a, const std::string &b) -> std::string {
        switch (pr.op[i])
        {
        case VM_ADD: return a + " + " + b;
        case VM_SUB: return a + " - " + b;
        case VM_MUL: return a + " * " + b;
        case VM_DIV: return a + " / " + b;
        case VM_NEG: return "-" + a;
        case VM_POW: return "pow(" + a + ", " + b + ")";
        case VM_SIGN: return "(" + a + " > 0.0 ? 1.0 : (" + a + " < 0.0 ? -1.0 : 0.0))";
        case VM_PNORM: return "0.5 * erfc(-" + a + " * 0.70710678118654752440)";
        case VM_PSI0: case VM_PSI1: case VM_PSI2: case VM_PSI3: case VM_PSI4:
            return "gpsigamma(" + a + ", " + std::to_string((int)pr.op[i] - (int)VM_PSI0) + ")";
        default: return std::string(fn1[pr.op[i]]) + "(" + a + ")";
        }
    };
    // operands inside the prologue: parameters, constants, earlier parameter-only results
    auto pref = [&](int slot) -> std::string {
        if (slot < pr.p)
            return "th[" + std::to_string(slot) + "]";
        if (slot < base)
            return rtc_fmt_double(pr.consts[slot - 2 * pr.p - pr.nx]);
        return "u" + std::to_string(slot - base);
    };
    // operands inside the row closure
    auto rref = [&](int slot) -> std::string {
        if (slot < pr.p)
            return "th[" + std::to_string(slot) + "]";
        if (slot < pr.p + pr.nx)
            return "xr[" + std::to_string(slot - pr.p) + "]";
        if (slot < base)
            return rtc_fmt_double(pr.consts[slot - 2 * pr.p - pr.nx]);
        if (H.po[slot - base])
            return "pre[" + std::to_string(H.export_of[slot - base]) + "]";
        return "v" + std::to_string(slot - base);
    };
    const std::string off = "        _Pragma(\"clang fp contract(off)\")\n";
    std::string s;
    s += "    static constexpr int NPRE = " + std::to_string(H.npre) + ";\n";
    s += "    template <class TH, class PRE> __device__ __forceinline__ static void prologue(const TH &th, PRE &pre) {\n" + off;
    for (int i = 0; i < nops; ++i)
    {
        if (!H.po[i])
            continue;
        s += "        const double u" + std::to_string(i) + " = " + expr_of(i, pref(pr.a[i]), pref(pr.b[i])) + ";\n";
        if (H.export_of[i] >= 0)
            s += "        pre.set(" + std::to_string(H.export_of[i]) + ", u" + std::to_string(i) + ");\n";
    }
    for (int slot : H.recip_slots)
        s += "        pre.set(" + std::to_string(H.recip_of[slot]) + ", 1.0 / " + pref(slot) + ");\n";
    s += "    }\n";
    // the row closure: gradient entry by gradient entry (see rtc_emit_ops), parameter-only instructions left out
    std::vector<int> order;
    std::vector<std::vector<int>> sets_of(nops);
    std::vector<char> barrier_after(nops, 0), seen(nops, 0);
    int sink_group = 6;
    if (const char *e = getenv("GSLNLS_RTC_SINK_GROUP"))
        sink_group = atoi(e) > 0 ? atoi(e) : 6;
    std::vector<int> stack;
    auto is_row = [&](int slot) { return slot >= base && slot - base < nops && !H.po[slot - base]; };
    auto visit = [&](int root) {
        if (!is_row(root))
            return;
        stack.push_back(root - base);
        while (!stack.empty())
        {
            const int i = stack.back();
            if (seen[i] == 2)
            {
                stack.pop_back();
                continue;
            }
            if (seen[i] == 0)
            {
                seen[i] = 1;
                for (int opnd : {(int)pr.a[i], (int)pr.b[i]})
                    if (is_row(opnd) && seen[opnd - base] == 0)
                        stack.push_back(opnd - base);
            }
            else
            {
                seen[i] = 2;
                order.push_back(i);
                stack.pop_back();
            }
        }
    };
    int in_group = 0;
    std::vector<int> late_sets;
    for (int k = 0; k < pr.p; ++k)
    {
        const int root = pr.grad_slot[k];
        if (!is_row(root))
        {
            late_sets.push_back(k); // (a parameter, a column, a constant, a parameter-only result)
            continue;
        }
        visit(root);
        sets_of[root - base].push_back(k);
        if (++in_group >= sink_group && !order.empty())
        {
            barrier_after[order.back()] = 1;
            in_group = 0;
        }
    }
    visit(pr.value_slot);
    s += "    template <class TH, class PRE, class XR, class SINK> __device__ __forceinline__ static double value_grad_sink_pre(const TH &th, const PRE &pre, const XR &xr, SINK &out) {\n" + off;
    for (int i : order)
    {
        std::string e;
        if (pr.op[i] == VM_DIV && H.recip_of[pr.b[i]] >= 0)
            e = rref(pr.a[i]) + " * pre[" + std::to_string(H.recip_of[pr.b[i]]) + "]";
        else
            e = expr_of(i, rref(pr.a[i]), rref(pr.b[i]));
        s += "        const double v" + std::to_string(i) + " = " + e + ";\n";
        for (int k : sets_of[i])
            s += "        out.set(" + std::to_string(k) + ", v" + std::to_string(i) + ");\n";
        if (barrier_after[i])
            s += "        __builtin_amdgcn_sched_barrier(0);\n";
    }
    for (int k : late_sets)
        s += "        out.set(" + std::to_string(k) + ", " + rref(pr.grad_slot[k]) + ");\n";
    s += "        return " + rref(pr.value_slot) + ";\n    }\n";
    return s;
}

// struct ModelJit: the row-model interface of models.hpp for this program.  One interpreted instruction = one
// statement, contraction off inside the bodies: a product and the sum that follows it stay two roundings, as in the
// interpreter -- that is what makes the native and the interpreted fit agree bit for bit.
template <class Prog>
inline std::string rtc_emit_model(const Prog &pr, int nx_model)
{
    const int base = 2 * pr.p + pr.nx + pr.nconst;
    auto ref = [&](int slot) -> std::string {
        if (slot < pr.p)
            return "th[" + std::to_string(slot) + "]";
        if (slot < pr.p + pr.nx)
            return "xr[" + std::to_string(slot - pr.p) + "]";
        if (slot < 2 * pr.p + pr.nx)
            return "dir[" + std::to_string(slot - pr.p - pr.nx) + "]";
        if (slot < base)
            return rtc_fmt_double(pr.consts[slot - 2 * pr.p - pr.nx]);
        return "v" + std::to_string(slot - base);
    };
    const std::string off = "        _Pragma(\"clang fp contract(off)\")\n";
    std::string s;
    s += "namespace gslnls {\nstruct ModelJit {\n";
    s += "    static constexpr int ID = 101, P = " + std::to_string(pr.p) + ", NX = " + std::to_string(nx_model) + ";\n";
    s += std::string("    static constexpr bool HAS_FVV = ") + (pr.nfvv > 0 ? "true" : "false") + ";\n";
    s += "    template <class TH, class XR> __device__ __forceinline__ static double value(const TH &th, const XR &xr) {\n" + off;
    s += rtc_emit_ops(pr, pr.nvalue);
    s += "        return " + ref(pr.value_slot) + ";\n    }\n";
    s += "    template <class TH, class XR> __device__ __forceinline__ static double value_grad(const TH &th, const XR &xr, double *g) {\n" + off;
    s += rtc_emit_ops(pr, pr.nops);
    for (int k = 0; k < pr.p; ++k)
        s += "        g[" + std::to_string(k) + "] = " + ref(pr.grad_slot[k]) + ";\n";
    s += "        return " + ref(pr.value_slot) + ";\n    }\n";
    // gradient entries handed out one by one (wide path: straight into the LDS tile, no P-register array)
    s += "    template <class TH, class XR, class SINK> __device__ __forceinline__ static double value_grad_sink(const TH &th, const XR &xr, SINK &out) {\n" + off;
    s += rtc_emit_ops(pr, pr.nops, true);
    s += "        return " + ref(pr.value_slot) + ";\n    }\n";
    // ... and the same closure with the parameter-only part hoisted (round 5; GSLNLS_RTC_NO_HOIST=1: NPRE = 0 and the
    // kernels take the closure above)
    if (getenv("GSLNLS_RTC_NO_HOIST"))
        s += "    static constexpr int NPRE = 0;\n    template <class TH, class PRE> __device__ static void prologue(const TH &, PRE &) {}\n"
             "    template <class TH, class PRE, class XR, class SINK> __device__ __forceinline__ static double value_grad_sink_pre(const TH &th, const PRE &, const XR &xr, SINK &out) { return value_grad_sink(th, xr, out); }\n";
    else
        s += rtc_emit_hoisted(pr);
    if (pr.nfvv > 0)
    {
        s += "    template <class TH, class DIR, class XR> __device__ __forceinline__ static double fvv(const TH &th, const DIR &dir, const XR &xr) {\n" + off;
        s += rtc_emit_ops(pr, pr.nfvv);
        s += "        return " + ref(pr.fvv_slot) + ";\n    }\n";
    }
    else
        s += "    template <class TH, class DIR, class XR> __device__ static double fvv(const TH &, const DIR &, const XR &) { return NAN; }\n";
    s += "};\n}\n";
    return s;
}

inline unsigned long long rtc_hash(const std::string &s)
{
    unsigned long long h = 1469598103934665603ull;
    for (unsigned char c : s)
    {
        h ^= c;
        h *= 1099511628211ull;
    }
    return h;
}

// A cache directory is only used when it belongs to this user and nobody else can write to it: code objects found
// there are loaded onto the device of the calling process, so a directory another local user could have prepared (or
// can write to) must never be trusted.
inline bool rtc_dir_is_private(const std::string &d)
{
    struct stat st;
    if (lstat(d.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) // lstat: a symlink planted under /tmp is not a directory
        return false;
    if (st.st_uid != getuid() || (st.st_mode & (S_IWGRP | S_IWOTH)) != 0)
        return false;
    return access(d.c_str(), W_OK) == 0;
}

// "" = no usable private cache directory: every process then compiles for itself (nothing is written)
inline std::string rtc_cache_dir()
{
    std::vector<std::string> cand;
    if (const char *e = getenv("GSLNLS_JIT_CACHE"))
        cand.push_back(e);
    else if (const char *h = getenv("HOME"))
        cand.push_back(std::string(h) + "/.cache/gslnls_amd");
    cand.push_back("/tmp/gslnls_amd_jit_" + std::to_string((long)getuid()));
    for (const std::string &d : cand)
    {
        const size_t k = d.rfind('/');
        if (k != std::string::npos && k > 0)
            mkdir(d.substr(0, k).c_str(), 0700);
        mkdir(d.c_str(), 0700);
        if (rtc_dir_is_private(d))
            return d;
    }
    return "";
}

// ---------------------------------------------------------------------------------------------------------
struct RtcApi
{
    void *handle = nullptr;
    decltype(&hiprtcCreateProgram) CreateProgram = nullptr;
    decltype(&hiprtcCompileProgram) CompileProgram = nullptr;
    decltype(&hiprtcDestroyProgram) DestroyProgram = nullptr;
    decltype(&hiprtcGetCodeSize) GetCodeSize = nullptr;
    decltype(&hiprtcGetCode) GetCode = nullptr;
    decltype(&hiprtcGetProgramLogSize) GetProgramLogSize = nullptr;
    decltype(&hiprtcGetProgramLog) GetProgramLog = nullptr;
    decltype(&hiprtcAddNameExpression) AddNameExpression = nullptr;
    decltype(&hiprtcGetLoweredName) GetLoweredName = nullptr;
    std::string err;

    bool load()
    {
        if (handle)
            return true;
        if (const char *e = getenv("GSLNLS_HIPRTC")) // test hook: "none" simulates a host without the compiler
            if (!strcmp(e, "none"))
            {
                err = "hiprtc disabled (GSLNLS_HIPRTC=none)";
                return false;
            }
        for (const char *nm : {"libhiprtc.so", "libhiprtc.so.7"})
            if ((handle = dlopen(nm, RTLD_NOW | RTLD_NOLOAD)))
                break;
        if (!handle)
            for (const char *nm : {"libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so.7", "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so"})
                if ((handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL)))
                    break;
        if (!handle)
        {
            err = std::string("hiprtc not found: ") + dlerror();
            return false;
        }
#define GSLNLS_RTC_SYM(f) f = (decltype(f))dlsym(handle, "hiprtc" #f)
        GSLNLS_RTC_SYM(CreateProgram);
        GSLNLS_RTC_SYM(CompileProgram);
        GSLNLS_RTC_SYM(DestroyProgram);
        GSLNLS_RTC_SYM(GetCodeSize);
        GSLNLS_RTC_SYM(GetCode);
        GSLNLS_RTC_SYM(GetProgramLogSize);
        GSLNLS_RTC_SYM(GetProgramLog);
        GSLNLS_RTC_SYM(AddNameExpression);
        GSLNLS_RTC_SYM(GetLoweredName);
#undef GSLNLS_RTC_SYM
        if (!CreateProgram || !CompileProgram || !DestroyProgram || !GetCodeSize || !GetCode || !AddNameExpression ||
            !GetLoweredName)
        {
            err = "hiprtc lacks a symbol";
            handle = nullptr;
            return false;
        }
        return true;
    }
};

inline RtcApi &rtc_api()
{
    static RtcApi api;
    return api;
}

inline std::atomic<bool> &rtc_shutting_down();

// one compiled translation unit: code object + the mangled names of the kernels that were asked for
struct RtcUnit
{
    std::vector<char> code;
    std::map<std::string, std::string> lowered;
};

// developer switch: GSLNLS_RTC_EXTRA_FLAGS = more compiler options (space separated; part of the cache key)
inline const std::vector<std::string> &rtc_extra_flags()
{
    static const std::vector<std::string> v = [] {
        std::vector<std::string> out;
        if (const char *e = getenv("GSLNLS_RTC_EXTRA_FLAGS"))
        {
            std::string cur;
            for (const char *c = e;; ++c)
            {
                if (*c == ' ' || *c == '\0')
                {
                    if (!cur.empty())
                        out.push_back(cur);
                    cur.clear();
                    if (!*c)
                        break;
                }
                else
                    cur.push_back(*c);
            }
        }
        return out;
    }();
    return v;
}
inline std::string rtc_flags_text()
{
    std::string s = "--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -mllvm -amdgpu-kernarg-preload-count=16";
    for (const std::string &f : rtc_extra_flags())
        s += " " + f;
    return s;
}

// source -> code object (no device needed; thread-safe with respect to the HIP runtime: hiprtc / comgr only)
inline bool rtc_compile(const std::string &source, const std::vector<std::string> &exprs, RtcUnit &out, std::string &log)
{
    static std::mutex mu; // one compilation at a time per process
    std::lock_guard<std::mutex> lock(mu);
    if (rtc_shutting_down().load())
    {
        log = "process is exiting";
        return false;
    }
    if (const char *dump = getenv("GSLNLS_RTC_DUMP")) // developer aid: the generated translation unit as text
    {
        char name[512];
        snprintf(name, sizeof name, "%s/gslnls_rtc_%016llx.hip", dump, rtc_hash(source));
        if (FILE *fh = fopen(name, "w"))
        {
            fputs(source.c_str(), fh);
            fclose(fh);
        }
    }
    RtcApi &api = rtc_api();
    if (!api.load())
    {
        log = api.err;
        return false;
    }
    int nh = 0;
    const RtcHeader *hd = rtc_embedded_headers(&nh);
    std::vector<std::string> texts(nh);
    std::vector<const char *> hsrc(nh), hname(nh);
    for (int k = 0; k < nh; ++k)
    {
        texts[k].assign(hd[k].begin, hd[k].end);
        hsrc[k] = texts[k].c_str();
        hname[k] = hd[k].name;
    }
    hiprtcProgram prog = nullptr;
    if (api.CreateProgram(&prog, source.c_str(), "gslnls_model.hip", nh, hsrc.data(), hname.data()) != HIPRTC_SUCCESS)
    {
        log = "hiprtcCreateProgram failed";
        return false;
    }
    for (const std::string &e : exprs)
        (void)api.AddNameExpression(prog, e.c_str());
    std::vector<const char *> opts = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on", "-mllvm",
                                      "-amdgpu-kernarg-preload-count=16"};
    for (const std::string &f : rtc_extra_flags())
        opts.push_back(f.c_str());
    const hiprtcResult r = api.CompileProgram(prog, (int)opts.size(), opts.data());
    size_t ls = 0;
    if (api.GetProgramLogSize && api.GetProgramLog && api.GetProgramLogSize(prog, &ls) == HIPRTC_SUCCESS && ls > 1)
    {
        log.assign(ls, '\0');
        (void)api.GetProgramLog(prog, &log[0]);
    }
    bool ok = (r == HIPRTC_SUCCESS);
    if (ok)
    {
        size_t cs = 0;
        ok = api.GetCodeSize(prog, &cs) == HIPRTC_SUCCESS && cs > 0;
        if (ok)
        {
            out.code.resize(cs);
            ok = api.GetCode(prog, out.code.data()) == HIPRTC_SUCCESS;
        }
        for (const std::string &e : exprs)
        {
            const char *low = nullptr;
            if (ok && api.GetLoweredName(prog, e.c_str(), &low) == HIPRTC_SUCCESS && low)
                out.lowered[e] = low;
            else
                ok = false;
        }
    }
    else if (log.empty())
        log = "hiprtcCompileProgram failed";
    (void)api.DestroyProgram(&prog);
    return ok;
}

// cache file: "GSLRTC1\n" | n | n x (len expr, expr, len name, name) | len code | code
inline bool rtc_cache_read(const std::string &path, RtcUnit &u)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f)
        return false;
    bool ok = false;
    char magic[8];
    unsigned long long n = 0;
    auto rd_str = [&](std::string &s) {
        unsigned long long len = 0;
        if (fread(&len, sizeof len, 1, f) != 1 || len > (1u << 20))
            return false;
        s.resize(len);
        return len == 0 || fread(&s[0], 1, len, f) == len;
    };
    if (fread(magic, 1, 8, f) == 8 && !memcmp(magic, "GSLRTC1\n", 8) && fread(&n, sizeof n, 1, f) == 1 && n < 64)
    {
        ok = true;
        for (unsigned long long k = 0; ok && k < n; ++k)
        {
            std::string e, l;
            ok = rd_str(e) && rd_str(l);
            if (ok)
                u.lowered[e] = l;
        }
        unsigned long long cs = 0;
        ok = ok && fread(&cs, sizeof cs, 1, f) == 1 && cs > 0 && cs < (1ull << 30);
        if (ok)
        {
            u.code.resize(cs);
            ok = fread(u.code.data(), 1, cs, f) == cs;
        }
    }
    fclose(f);
    return ok;
}

inline void rtc_cache_write(const std::string &path, const RtcUnit &u)
{
    const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f)
        return;
    auto wr_str = [&](const std::string &s) {
        const unsigned long long len = s.size();
        fwrite(&len, sizeof len, 1, f);
        fwrite(s.data(), 1, s.size(), f);
    };
    fwrite("GSLRTC1\n", 1, 8, f);
    const unsigned long long n = u.lowered.size(), cs = u.code.size();
    fwrite(&n, sizeof n, 1, f);
    for (const auto &kv : u.lowered)
    {
        wr_str(kv.first);
        wr_str(kv.second);
    }
    fwrite(&cs, sizeof cs, 1, f);
    fwrite(u.code.data(), 1, u.code.size(), f);
    const bool ok = !ferror(f);
    fclose(f);
    if (ok)
        rename(tmp.c_str(), path.c_str()); // atomic: concurrent processes see nothing or the whole file
    else
        unlink(tmp.c_str());
}

// ---------------------------------------------------------------------------------------------------------
enum
{
    RTC_NONE = 0,
    RTC_BUILDING = 1, // inside the compiler
    RTC_READY = 2,
    RTC_QUEUED = 3,   // waiting for the background compiler (reported as "being built" at the C ABI)
    RTC_FAILED = -1
};

struct RtcEntry
{
    std::atomic<int> state{RTC_NONE};
    RtcUnit unit;
    std::string log, cache_path;
    double build_s = 0.0;
    bool from_cache = false;
    // modules per device ordinal (loaded by the thread that launches)
    std::map<int, hipModule_t> modules;
    std::map<std::pair<int, std::string>, hipFunction_t> functions;
};

struct RtcJob
{
    std::shared_ptr<RtcEntry> entry;
    std::string source;
    std::vector<std::string> exprs;
};

// ONE background compiler thread per process, fed through a queue (a thread per formula would be unbounded: every new
// formula x Jacobian kind under GSLNLS_LOWER_AUTO asks for a build).  The newest request is served first -- a process that
// walks through many formulas (a test suite, a model search) must not make the current one wait behind all the others
struct RtcRegistry
{
    std::mutex mu;
    std::map<unsigned long long, std::shared_ptr<RtcEntry>> entries;
    std::mutex qmu;
    std::condition_variable qcv;
    std::deque<RtcJob> queue;
    std::thread worker;
    bool stop = false;
    static constexpr size_t QUEUE_CAP = 256; // beyond that a request is dropped (the interpreter keeps serving it)
    ~RtcRegistry() { shutdown(); }
    void shutdown()
    {
        {
            std::lock_guard<std::mutex> lock(qmu);
            stop = true;
            for (RtcJob &j : queue)
            {
                int expect = RTC_QUEUED;
                j.entry->state.compare_exchange_strong(expect, RTC_FAILED);
            }
            queue.clear();
        }
        qcv.notify_all();
        // a build still running at process exit is waited for (a compiler thread torn down in the middle of comgr
        // takes the process with it)
        if (worker.joinable())
            worker.join();
    }
};

inline RtcRegistry &rtc_registry()
{
    static RtcRegistry r;
    return r;
}

inline double rtc_now()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

// Process exit: a compiler thread must not run into the teardown of comgr / LLVM (their static objects are destroyed
// while the thread is still compiling: "LLVM ERROR", heap corruption or a hang -- all three seen).  rtc_at_exit turns
// queued builds into no-ops and waits for the one in flight.  It has to run BEFORE LLVM's exit handlers, and those are
// registered lazily, whenever a compile first touches a static: so (1) the host application calls gslnls_shutdown()
// before it exits (the Python mirror registers it with atexit, the R shim calls it from .onUnload / at session end:
// INTEGRATION.md) -- that is the guarantee; (2) as a net for hosts that do not, the same function is registered with
// atexit() again after every finished compile, i.e. behind everything LLVM has registered so far.
inline std::atomic<bool> &rtc_shutting_down()
{
    static std::atomic<bool> f{false};
    return f;
}
inline void rtc_at_exit()
{
    if (rtc_shutting_down().exchange(true))
        return;
    rtc_registry().shutdown();
}

inline void rtc_build(RtcEntry *e, const std::string &source, const std::vector<std::string> &exprs)
{
    if (rtc_shutting_down().load())
    {
        e->state.store(RTC_FAILED, std::memory_order_release);
        return;
    }
    const double t0 = rtc_now();
    RtcUnit u;
    std::string log;
    const bool ok = rtc_compile(source, exprs, u, log);
    e->log = log;
    e->build_s = rtc_now() - t0;
    if (ok)
    {
        e->unit = std::move(u);
        if (!e->cache_path.empty())
            rtc_cache_write(e->cache_path, e->unit);
        e->state.store(RTC_READY, std::memory_order_release);
    }
    else
        e->state.store(RTC_FAILED, std::memory_order_release);
    static std::atomic<int> nreg{0};
    if (!rtc_shutting_down().load() && nreg.fetch_add(1) < 24) // (atexit guarantees 32 slots)
        (void)atexit(rtc_at_exit);
}

inline void rtc_worker_main()
{
    RtcRegistry &reg = rtc_registry();
    for (;;)
    {
        RtcJob job;
        {
            std::unique_lock<std::mutex> lock(reg.qmu);
            reg.qcv.wait(lock, [&] { return reg.stop || !reg.queue.empty(); });
            if (reg.stop)
                return;
            job = std::move(reg.queue.front());
            reg.queue.pop_front();
        }
        int expect = RTC_QUEUED;
        if (job.entry->state.compare_exchange_strong(expect, RTC_BUILDING)) // (else: a waiting caller took it over)
            rtc_build(job.entry.get(), job.source, job.exprs);
    }
}

// The code object of `source` (which must name every kernel in `exprs`).  wait: build now if it is neither in memory
// nor in the disk cache; otherwise a missing one is built on a background thread and nullptr-like (state BUILDING)
// comes back at once.  Never blocks on a build another call started unless `wait`.
inline unsigned long long rtc_key(const std::string &source, const std::vector<std::string> &exprs)
{
    std::string keytext = source + "\n//" + rtc_flags_text() + "\n//" + gslnls_version();
    int nh = 0;
    const RtcHeader *hd = rtc_embedded_headers(&nh);
    static const unsigned long long hh = [&]() {
        unsigned long long a = 7;
        for (int k = 0; k < nh; ++k) // the kernels' own text is part of the key: a rebuilt library never picks up stale code
            a = a * 1099511628211ull ^ rtc_hash(std::string(hd[k].begin, hd[k].end));
        return a;
    }();
    unsigned long long h = rtc_hash(keytext) * 1099511628211ull ^ hh;
    for (const std::string &e : exprs)
        h = h * 1099511628211ull ^ rtc_hash(e);
    return h;
}

// state of a unit without starting anything (RTC_NONE when nobody asked for it yet and the cache does not hold it)
inline int rtc_request_peek(const std::string &source, const std::vector<std::string> &exprs)
{
    const unsigned long long h = rtc_key(source, exprs);
    RtcRegistry &reg = rtc_registry();
    {
        std::lock_guard<std::mutex> lock(reg.mu);
        auto it = reg.entries.find(h);
        if (it != reg.entries.end())
            return it->second->state.load(std::memory_order_acquire);
    }
    const std::string dir = rtc_cache_dir();
    if (dir.empty())
        return RTC_NONE;
    char name[64];
    snprintf(name, sizeof name, "/gslnls_rtc_%016llx.bin", h);
    return access((dir + name).c_str(), R_OK) == 0 ? RTC_READY : RTC_NONE;
}

inline std::shared_ptr<RtcEntry> rtc_request(const std::string &source, const std::vector<std::string> &exprs, bool wait)
{
    const unsigned long long h = rtc_key(source, exprs);
    RtcRegistry &reg = rtc_registry();
    std::shared_ptr<RtcEntry> ent;
    {
        std::lock_guard<std::mutex> lock(reg.mu);
        auto it = reg.entries.find(h);
        if (it != reg.entries.end())
            ent = it->second;
        else
        {
            ent = std::make_shared<RtcEntry>();
            reg.entries[h] = ent;
            const std::string dir = rtc_cache_dir();
            if (!dir.empty())
            {
                char name[64];
                snprintf(name, sizeof name, "/gslnls_rtc_%016llx.bin", h);
                ent->cache_path = dir + name;
                RtcUnit u;
                if (rtc_cache_read(ent->cache_path, u))
                {
                    bool all = true;
                    for (const std::string &e : exprs)
                        all = all && u.lowered.count(e);
                    if (all)
                    {
                        ent->unit = std::move(u);
                        ent->from_cache = true;
                        ent->state.store(RTC_READY, std::memory_order_release);
                    }
                }
            }
        }
    }
    int st = ent->state.load(std::memory_order_acquire);
    if (st == RTC_NONE)
    {
        int expect = RTC_NONE;
        if (wait)
        {
            if (ent->state.compare_exchange_strong(expect, RTC_BUILDING))
                rtc_build(ent.get(), source, exprs);
        }
        else if (!rtc_api().load())
        {
            if (ent->state.compare_exchange_strong(expect, RTC_FAILED))
                ent->log = rtc_api().err;
        }
        else if (ent->state.compare_exchange_strong(expect, RTC_QUEUED))
        {
            static const bool registered = (atexit(rtc_at_exit), true); // after hiprtc's own: runs before them
            (void)registered;
            std::lock_guard<std::mutex> lock(reg.qmu);
            if (reg.stop || reg.queue.size() >= RtcRegistry::QUEUE_CAP)
                ent->state.store(RTC_NONE, std::memory_order_release); // dropped: a later request may try again
            else
            {
                reg.queue.push_front({ent, source, exprs}); // (newest first: it is the formula being fitted right now)
                if (!reg.worker.joinable())
                    reg.worker = std::thread(rtc_worker_main);
                reg.qcv.notify_one();
            }
        }
        st = ent->state.load(std::memory_order_acquire);
    }
    if (wait && st == RTC_QUEUED)
    {
        // still waiting for the background compiler: build it here and now instead (the worker skips what it no
        // longer finds QUEUED)
        int expect = RTC_QUEUED;
        if (ent->state.compare_exchange_strong(expect, RTC_BUILDING))
            rtc_build(ent.get(), source, exprs);
        st = ent->state.load(std::memory_order_acquire);
    }
    if (wait)
        while (ent->state.load(std::memory_order_acquire) == RTC_BUILDING) // the background compiler is in it
            usleep(1000);
    return ent;
}

// kernel `expr` of a READY entry on the current device (module loaded on first use)
inline hipFunction_t rtc_function(RtcEntry &e, const std::string &expr, std::string &msg)
{
    if (e.state.load(std::memory_order_acquire) != RTC_READY)
        return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess)
    {
        msg = "no current device";
        return nullptr;
    }
    auto fk = std::make_pair(dev, expr);
    auto fit = e.functions.find(fk);
    if (fit != e.functions.end())
        return fit->second;
    auto mit = e.modules.find(dev);
    if (mit == e.modules.end())
    {
        hipModule_t mod = nullptr;
        const hipError_t r = hipModuleLoadData(&mod, e.unit.code.data());
        if (r != hipSuccess)
        {
            msg = std::string("hipModuleLoadData: ") + hipGetErrorString(r);
            (void)hipGetLastError();
            return nullptr;
        }
        mit = e.modules.insert({dev, mod}).first;
    }
    auto lit = e.unit.lowered.find(expr);
    if (lit == e.unit.lowered.end())
    {
        msg = "kernel not in this code object: " + expr;
        return nullptr;
    }
    hipFunction_t fn = nullptr;
    const hipError_t r = hipModuleGetFunction(&fn, mit->second, lit->second.c_str());
    if (r != hipSuccess)
    {
        msg = std::string("hipModuleGetFunction: ") + hipGetErrorString(r);
        (void)hipGetLastError();
        return nullptr;
    }
    e.functions[fk] = fn;
    return fn;
}

// ---------------------------------------------------------------------------------------------------------
// the translation unit of the dense (p <= 9) kernels of one program and one Jacobian kind
inline std::string rtc_dense_source(const VmProgram &pr, int nx_model)
{
    std::string s = "// generated by gslnls rtc_host.hpp\n#include \"dense_kernels.hpp\"\n";
    s += rtc_emit_model(pr, nx_model);
    return s;
}
inline std::string rtc_step_expr(int jacmode, int T)
{
    return "&gslnls::lm_step_kernel<gslnls::ModelJit, " + std::to_string(jacmode) + ", " + std::to_string(T) + ">";
}
inline std::string rtc_finalize_expr(int jacmode, int T)
{
    return "&gslnls::lm_finalize_kernel<gslnls::ModelJit, " + std::to_string(jacmode) + ", " + std::to_string(T) + ">";
}

} // namespace gslnls
