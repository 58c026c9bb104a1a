// vm_model.hpp -- ModelVM<P>: the row-model interface (models.hpp) implemented by interpreting the
// compiled expression program (vm_program.hpp).  This is what lets an ARBITRARY formula
// `y ~ expr(x1, x2, x3, theta)` run on the device paths (dense LM, multi-start, IRLS): value and
// analytic gradient come from one straight-line program, finite differences re-run its value part.
// The program lives in constant memory of the translation unit that instantiates the kernels
// (vm_models.hip); it is re-uploaded at every API entry, so several expression problems can coexist
// in a (single-threaded) process.
#pragma once
#include "vm_program.hpp"

namespace gslnls
{

#if defined(__HIPCC__)
extern __constant__ VmProgram c_vm_prog;
#endif

template <int P_>
struct ModelVM
{
    static constexpr int ID = 100, P = P_, NX = VM_NX;
    static constexpr bool HAS_FVV = true; // symbolic, when the program carries the third closure (nfvv > 0)
#if defined(__HIP_DEVICE_COMPILE__)
    __device__ static double value(const double *th, const double *xr)
    {
        double slot[VM_MAX_SLOTS];
        vm_run(c_vm_prog, th, xr, nullptr, c_vm_prog.nvalue, slot);
        return slot[c_vm_prog.value_slot];
    }
    __device__ static double value_grad(const double *th, const double *xr, double *g)
    {
        double slot[VM_MAX_SLOTS];
        vm_run(c_vm_prog, th, xr, nullptr, c_vm_prog.nops, slot);
#pragma unroll
        for (int k = 0; k < P; ++k)
            g[k] = slot[c_vm_prog.grad_slot[k]];
        return slot[c_vm_prog.value_slot];
    }
    __device__ static double fvv(const double *th, const double *v, const double *xr)
    {
        double slot[VM_MAX_SLOTS];
        vm_run(c_vm_prog, th, xr, v, c_vm_prog.nfvv, slot);
        return slot[c_vm_prog.fvv_slot];
    }
#else
    // host pass of hipcc: never executed (kernels only); keeps the templates well-formed
    static double value(const double *, const double *) { return NAN; }
    static double value_grad(const double *, const double *, double *) { return NAN; }
    static double fvv(const double *, const double *, const double *) { return NAN; }
#endif
};

} // namespace gslnls
