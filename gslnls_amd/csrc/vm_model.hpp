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

#if defined(__HIP_DEVICE_COMPILE__)
// The slot file of one thread as a strided view of the workgroup's dynamic LDS: slot i of thread t at
// [i * blockDim.x + t] (consecutive lanes, consecutive banks).  A slot file in a local array is scratch memory: every
// interpreted instruction then waits for two loads that follow the previous instruction's store through the vector
// memory path (~1200 cycles per instruction); through LDS the same chain is two ds_read behind a ds_write.
extern __shared__ double vm_dyn_lds[];
struct VmLdsSlots
{
    double *base; // &vm_dyn_lds[threadIdx.x]
    int stride;   // blockDim.x
    __device__ double &operator[](int i) const { return base[(size_t)i * stride]; }
};
#endif

// LDS_SLOTS: the twin used by lm_step_kernel when the program's slot file fits the workgroup's LDS (dense_host.hpp
// decides per problem and sizes the launch); same program, same arithmetic, same order
template <int P_, bool LDS_SLOTS_ = false>
struct ModelVM
{
    static constexpr int ID = 100, P = P_, NX = VM_NX;
    static constexpr bool HAS_FVV = true; // symbolic, when the program carries the third closure (nfvv > 0)
    static constexpr bool LDS_SLOTS = LDS_SLOTS_;
    using LdsTwin = ModelVM<P_, true>;
#if defined(__HIP_DEVICE_COMPILE__)
    __device__ static double value(const double *th, const double *xr)
    {
        if constexpr (LDS_SLOTS)
        {
            const VmLdsSlots slot{vm_dyn_lds + threadIdx.x, (int)blockDim.x};
            vm_run(c_vm_prog, th, xr, nullptr, c_vm_prog.nvalue, slot);
            return slot[c_vm_prog.value_slot];
        }
        else
        {
            double slot[VM_MAX_SLOTS];
            vm_run(c_vm_prog, th, xr, nullptr, c_vm_prog.nvalue, slot);
            return slot[c_vm_prog.value_slot];
        }
    }
    __device__ static double value_grad(const double *th, const double *xr, double *g)
    {
        if constexpr (LDS_SLOTS)
        {
            const VmLdsSlots slot{vm_dyn_lds + threadIdx.x, (int)blockDim.x};
            vm_run(c_vm_prog, th, xr, nullptr, c_vm_prog.nops, slot);
#pragma unroll
            for (int k = 0; k < P; ++k)
                g[k] = slot[c_vm_prog.grad_slot[k]];
            return slot[c_vm_prog.value_slot];
        }
        else
        {
            double slot[VM_MAX_SLOTS];
            vm_run(c_vm_prog, th, xr, nullptr, c_vm_prog.nops, slot);
#pragma unroll
            for (int k = 0; k < P; ++k)
                g[k] = slot[c_vm_prog.grad_slot[k]];
            return slot[c_vm_prog.value_slot];
        }
    }
    __device__ static double fvv(const double *th, const double *v, const double *xr)
    {
        if constexpr (LDS_SLOTS)
        {
            const VmLdsSlots slot{vm_dyn_lds + threadIdx.x, (int)blockDim.x};
            vm_run(c_vm_prog, th, xr, v, c_vm_prog.nfvv, slot);
            return slot[c_vm_prog.fvv_slot];
        }
        else
        {
            double slot[VM_MAX_SLOTS];
            vm_run(c_vm_prog, th, xr, v, c_vm_prog.nfvv, slot);
            return slot[c_vm_prog.fvv_slot];
        }
    }
#else
    // host pass of hipcc: never executed (kernels only); keeps the templates well-formed
    static double value(const double *, const double *) { return NAN; }
    static double value_grad(const double *, const double *, double *) { return NAN; }
    static double fvv(const double *, const double *, const double *) { return NAN; }
#endif
};

} // namespace gslnls
