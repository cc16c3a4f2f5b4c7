// os2r_jit_unit.hip -- step kernels specialised for ONE robot, built at run time into a code object.
//
// gym_os2r_amd/jit.py writes the robot's constants as `os2r::gen::Tables<100>` (the same constexpr
// form as os2r_models_gen.hpp), then
//   hipcc --genco --offload-arch=gfx950 -O3 ... -DOS2R_REAL=double -DOS2R_JIT_CONTACT=1
//         -DOS2R_JIT_TABLES='"<generated header>"' os2r_jit_unit.hip -o <hash>.hsaco
// and hands the file to os2r_register_model_kernels (include/os2r.h).  The kernels are the same
// step_body as the compiled-in variants: every constant of the robot is folded into the code.
#include "os2r_kernels.hpp"

#ifndef OS2R_REAL
#error "OS2R_REAL must be float or double"
#endif
#ifndef OS2R_JIT_TABLES
#error "OS2R_JIT_TABLES must name the generated table header"
#endif
#ifndef OS2R_JIT_CONTACT
#error "OS2R_JIT_CONTACT must be 0 or 1"
#endif
#include OS2R_JIT_TABLES

namespace os2r {
using JitReal = OS2R_REAL;
using JitModel = StModel<JitReal, 100>;
}  // namespace os2r

// NAME: sweep counts read from the handle; NAME_s: the default counts as compile-time loop bounds
#define OS2R_JIT_KERNEL(NAME, CONTACT, DR)                                                        \
  extern "C" __global__ OS2R_STEP_KERNEL_ATTRS(OS2R_REAL) void NAME(const os2r::StepArgs<os2r::JitReal> A) { \
    os2r::step_body<os2r::JitReal, os2r::JitModel, CONTACT, DR, false>(A);                        \
  }                                                                                               \
  extern "C" __global__ OS2R_STEP_KERNEL_ATTRS(OS2R_REAL) void NAME##_s(const os2r::StepArgs<os2r::JitReal> A) { \
    os2r::step_body<os2r::JitReal, os2r::JitModel, CONTACT, DR, true>(A);                         \
  }

#if OS2R_JIT_CONTACT
OS2R_JIT_KERNEL(os2r_jit_step_c1_d0, true, false)
OS2R_JIT_KERNEL(os2r_jit_step_c1_d1, true, true)
// NAME_l: additionally the observation layout of the task the code object was built for (StLayout), announced
// in os2r_jit_layout = {kinds, sources, slots}; handles with another layout use the kernels above
#ifdef OS2R_JIT_LAYOUT_DIM
extern "C" __device__ __attribute__((used)) const unsigned long long os2r_jit_layout[3] = {
    OS2R_JIT_LAYOUT_KINDS, OS2R_JIT_LAYOUT_SRCS, OS2R_JIT_LAYOUT_DIM};
namespace os2r { using JitLayout = StLayout<OS2R_JIT_LAYOUT_KINDS, OS2R_JIT_LAYOUT_SRCS, OS2R_JIT_LAYOUT_DIM>; }
#define OS2R_JIT_LAYOUT_KERNEL(NAME, DR)                                                          \
  extern "C" __global__ OS2R_STEP_KERNEL_ATTRS(OS2R_REAL) void NAME(const os2r::StepArgs<os2r::JitReal> A) { \
    os2r::step_body<os2r::JitReal, os2r::JitModel, true, DR, true, os2r::JitLayout>(A);             \
  }
OS2R_JIT_LAYOUT_KERNEL(os2r_jit_step_c1_d0_l, false)
OS2R_JIT_LAYOUT_KERNEL(os2r_jit_step_c1_d1_l, true)
#endif
#else
OS2R_JIT_KERNEL(os2r_jit_step_c0_d0, false, false)
OS2R_JIT_KERNEL(os2r_jit_step_c0_d1, false, true)
#endif
