// vm_program.hpp -- the lowered form of an arbitrary model formula: a straight-line program over
// numbered slots, evaluated per observation by a tiny interpreter (host + device).
//
// This is the general answer to SURVEY.md 0.3 / 8(f).1: the reference evaluates
//   eval(formula[[3]], c(as.list(par), .data))                       (R/nls.R:565)
// and, for jac = TRUE, the closure stats::deriv() builds from the same expression
// (R/nls.R:588-599).  expr_compile.hpp turns the expression into a DAG, differentiates it
// symbolically with respect to every parameter (common subexpressions shared, constants folded)
// and emits one program whose first `nvalue` instructions compute the model value and whose
// remaining instructions add the P partial derivatives.
//
// Slot layout: [0, p) parameters | [p, p+nx) regressors of the row | [p+nx, 2p+nx) direction v of the second
// directional derivative (lmaccel, fvv = TRUE: the reference differentiates the formula twice with
// stats::deriv(..., hessian = TRUE), R/nls.R:600-640) | nconst constants | then one slot per instruction
// (instruction i writes slot base + i).  Three nested closures: [0, nvalue) value, [0, nops) + gradient,
// [0, nfvv) + D^2 f[v, v].
#pragma once
#include "lm_core.hpp"
#include "devmath.hpp"

namespace gslnls
{

constexpr int VM_MAX_OPS = 256;
constexpr int VM_MAX_CONST = 32;
constexpr int VM_MAX_P = 12;
constexpr int VM_NX = 3; // regressor columns carried per row (unused ones are zero)
constexpr int VM_MAX_SLOTS = 2 * VM_MAX_P + VM_NX + VM_MAX_CONST + VM_MAX_OPS;

enum VmOp : unsigned char
{
    VM_ADD = 1,
    VM_SUB,
    VM_MUL,
    VM_DIV,
    VM_NEG,
    VM_POW,
    VM_EXP,
    VM_LOG,
    VM_SIN,
    VM_COS,
    VM_TAN,
    VM_ATAN,
    VM_SQRT,
    VM_ABS,
    VM_TANH,
    VM_SIGN,
    // the rest of stats::deriv's table (R/nls.R:588-599 differentiates the formula with it): functions of one argument
    VM_SINH,
    VM_COSH,
    VM_ASIN,
    VM_ACOS,
    VM_LOG1P,
    VM_EXPM1,
    VM_PNORM, // standard normal distribution function: 0.5 erfc(-x / sqrt(2))
    // the gamma family of the same table (round 5): gamma, lgamma, and psigamma(x, n) for n = 0 (digamma), 1 (trigamma), 2, 3, 4
    // -- each the derivative of the one before it (devmath.hpp: gpsigamma)
    VM_GAMMA,
    VM_LGAMMA,
    VM_PSI0,
    VM_PSI1,
    VM_PSI2,
    VM_PSI3,
    VM_PSI4
};

template <int MAX_OPS, int MAX_CONST, int MAX_P, bool PACKED>
struct VmProgramT
{
    static constexpr int CAP_OPS = MAX_OPS, CAP_CONST = MAX_CONST, CAP_P = MAX_P;
    int p, nx, nconst, nops, nvalue; // nvalue: instructions needed for the value alone; nops: value + gradient
    int nfvv;                        // value + gradient + second directional derivative (0: not available)
    int value_slot, fvv_slot;
    int grad_slot[MAX_P];
    unsigned char op[MAX_OPS];
    unsigned short a[MAX_OPS], b[MAX_OPS];
    // the same instruction as one word for the interpreter: op | a << 8 | b << 20 (one scalar load instead of three
    // sub-word ones); op / a / b above stay the form the native lowering and the tests read
    unsigned int word[PACKED ? MAX_OPS : 1];
    double consts[MAX_CONST];
};
// the interpreter's program (constant memory, slot numbers packed in 12 bits)
using VmProgram = VmProgramT<VM_MAX_OPS, VM_MAX_CONST, VM_MAX_P, true>;
// the wide path's program (p <= 64, up to WIDE_NX data columns): host side only -- it is never interpreted, only printed
// as C++ for the in-process compiler (rtc_host.hpp), so its size is bounded by compile time, not by constant memory
constexpr int WIDE_MAX_P = 64, WIDE_NX = 8, WIDE_MAX_OPS = 8192, WIDE_MAX_CONST = 256;
using WideProgram = VmProgramT<WIDE_MAX_OPS, WIDE_MAX_CONST, WIDE_MAX_P, false>;
// formulas with more than 64 parameters (csrc/bd_host.hpp: the Jacobian as a matrix in HBM): slot numbers are 16 bits
constexpr int BIG_MAX_P = 512, BIG_MAX_OPS = 49152, BIG_MAX_CONST = 2048;
using BigProgram = VmProgramT<BIG_MAX_OPS, BIG_MAX_CONST, BIG_MAX_P, false>;
static_assert(2 * BIG_MAX_P + WIDE_NX + BIG_MAX_CONST + BIG_MAX_OPS < 65536, "slot numbers are unsigned short");
static_assert(VM_MAX_SLOTS < 4096, "slot numbers are packed in 12 bits");

GSLNLS_HD bool vm_is_binary(unsigned op) { return op <= VM_DIV || op == VM_POW; }

// The library functions with long bodies (pow and the trigonometric ones carry their argument reduction: hundreds to
// thousands of instructions each) stay out of line: inlined into the interpreter's switch, in every unrolled copy of the
// row loop, they made the loop tens of KB of code that no instruction cache holds, and every interpreted instruction
// paid for fetching its case.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __attribute__((noinline)) inline double vm_apply_long(unsigned int op, double x, double y)
#else
inline double vm_apply_long(unsigned int op, double x, double y)
#endif
{
    switch (op)
    {
    case VM_POW: return pow(x, y);
    case VM_LOG: return log(x);
    case VM_SIN: return sin(x);
    case VM_COS: return cos(x);
    case VM_TAN: return tan(x);
    case VM_ATAN: return atan(x);
    case VM_TANH: return tanh(x);
    case VM_SINH: return sinh(x);
    case VM_COSH: return cosh(x);
    case VM_ASIN: return asin(x);
    case VM_ACOS: return acos(x);
    case VM_LOG1P: return log1p(x);
    case VM_EXPM1: return expm1(x);
    case VM_PNORM: return 0.5 * erfc(-x * 0.70710678118654752440);
    case VM_GAMMA: return tgamma(x);
    case VM_LGAMMA: return lgamma(x);
    case VM_PSI0: return gpsigamma(x, 0);
    case VM_PSI1: return gpsigamma(x, 1);
    case VM_PSI2: return gpsigamma(x, 2);
    case VM_PSI3: return gpsigamma(x, 3);
    case VM_PSI4: return gpsigamma(x, 4);
    default: return NAN;
    }
}

GSLNLS_HD double vm_apply(unsigned char op, double x, double y)
{
    // the one-instruction operations (most of any program) are computed side by side and selected: one test instead
    // of the compare-and-branch tree of a switch, which cost more than the operation itself
    if (op <= VM_MUL)
    {
        const double s = (op == VM_SUB) ? -y : y;
        return (op == VM_MUL) ? x * y : x + s;
    }
    if (op == VM_NEG)
        return -x;
    if (op == VM_EXP)
        return gexp(x);
    if (op == VM_DIV)
        return x / y;
    switch (op)
    {
    case VM_SQRT: return sqrt(x);
    case VM_ABS: return fabs(x);
    case VM_SIGN: return x > 0.0 ? 1.0 : (x < 0.0 ? -1.0 : 0.0);
    default: return vm_apply_long(op, x, y);
    }
}

// everything that is not add / subtract / multiply / negate
GSLNLS_HD double vm_apply_rest(unsigned int op, double x, double y)
{
    if (op == VM_EXP)
        return gexp(x);
    if (op == VM_DIV)
        return x / y;
    switch (op)
    {
    case VM_SQRT: return sqrt(x);
    case VM_ABS: return fabs(x);
    case VM_SIGN: return x > 0.0 ? 1.0 : (x < 0.0 ? -1.0 : 0.0);
    default: return vm_apply_long(op, x, y);
    }
}

// `slot`: anything indexable that yields double lvalues -- a local array (scratch memory on the device), or the
// strided view of the workgroup's LDS that vm_model.hpp uses in the step kernel
template <class Slots>
GSLNLS_HD void vm_run(const VmProgram &prog, const double *th, const double *xr, const double *v, int upto, Slots &&slot)
{
    const int p = prog.p, nx = prog.nx, nc = prog.nconst;
    for (int k = 0; k < p; ++k)
        slot[k] = th[k];
    for (int c = 0; c < nx; ++c)
        slot[p + c] = xr[c];
    for (int k = 0; k < p; ++k)
        slot[p + nx + k] = v ? v[k] : 0.0;
    for (int c = 0; c < nc; ++c)
        slot[2 * p + nx + c] = prog.consts[c];
    const int base = 2 * p + nx + nc;
    // The result of an instruction is very often an operand of the next one: it is kept in a register and forwarded, so
    // that the chain does not go store -> load through the slot file at every step; the second operand is only read for
    // the binary operations.  (Program, operands and operations are the same for every lane: the branches are scalar.)
    int last = -1;
    double lastv = 0.0;
    // (Tried and measured on the C2 formula, 8 instructions per row: requesting the next word one instruction ahead,
    // and holding the program in registers read with v_readlane so that no scalar load sits in the loop -- both within
    // 1 % of this form.)
    // Counters of the interpreted step kernel (scripts/dev_vm_pmc.sh) said what an instruction cost: 14 scalar
    // branches and 40 scalar instructions, at ~18 cycles per instruction issued -- the loop was control flow.  So
    // the four one-instruction operations are computed side by side and selected, the second operand is always read
    // (unary instructions carry b = a), and one test sends everything else to vm_apply_rest: four branches per
    // instruction (loop, two forwards, the class).
    for (int i = 0; i < upto; ++i)
    {
        const unsigned int w = prog.word[i];
        const unsigned int op = w & 0xffu;
        const int a = (int)((w >> 8) & 0xfffu), b = (int)(w >> 20);
        double x, y;
        if (a == last)
            x = lastv;
        else
            x = slot[a];
        if (b == last)
            y = lastv;
        else
            y = slot[b];
        const double ys = (op == VM_SUB) ? -y : y;
        double r = (op == VM_MUL) ? x * y : x + ys;
        r = (op == VM_NEG) ? -x : r;
        if (op == VM_DIV || op > VM_NEG)
            r = vm_apply_rest(op, x, y);
        lastv = r;
        last = base + i;
        slot[last] = lastv;
    }
}

} // namespace gslnls
