// os2r_inst.hip — one instantiation unit of the step / reset kernels.
// Compiled once per (OS2R_REAL, OS2R_NQ) so the eight units build in parallel.
//
// Contact masks instantiated per chain length (bit b: body b carries ground-contact candidates):
//   0      contact off (bring-up configuration C2) or a model that cannot reach the ground
//   STD    the mask of the reference's URDF variant with that many dofs
//   FULL   every body (fallback for user-supplied models; empty bodies cost an empty loop)
// Joint axes: the reference's variants are compiled with their axes fixed (AXSTD: yaw about z
// for the 4/5-dof chains, x otherwise; all later joints about x); the FULL-mask fallback reads
// the axes from the model at run time.
#include "os2r_kernels.hpp"

#ifndef OS2R_REAL
#error "OS2R_REAL must be float or double"
#endif
#ifndef OS2R_NQ
#error "OS2R_NQ must be 1..5"
#endif

namespace os2r {

using T = OS2R_REAL;
constexpr int NQ = OS2R_NQ;
constexpr unsigned FULL = (1u << NQ) - 1u;
// monopod (5): bodies 1-4; monopod-fixed_hip (4): 1-3; monopod-fixed (3): 0-2; monopod-simple (2): none
constexpr unsigned STD = NQ == 5 ? 0x1Eu : (NQ == 4 ? 0x0Eu : FULL);
constexpr int AXSTD = NQ >= 4 ? 2 : 0;

template <unsigned CM, bool DR, int AX>
static void launch_step(const StepArgs<T>& a, hipStream_t s) {
  const dim3 grid((unsigned)((a.N + kWave - 1) / kWave)), block(kWave);
  hipLaunchKernelGGL((step_kernel<T, NQ, CM, DR, AX>), grid, block, 0, s, a);
}

template <typename R, int N_>
int step_unit(unsigned cmask, bool dr, bool std_axes, const StepArgs<R>& a, hipStream_t s);
template <typename R, int N_>
int reset_unit(bool dr, const StepArgs<R>& a, hipStream_t s);

// std_axes: the model's joint axes are (AXSTD, x, x, ...)
template <>
int step_unit<T, NQ>(unsigned cmask, bool dr, bool std_axes, const StepArgs<T>& a, hipStream_t s) {
  if ((cmask & ~FULL) != 0u) return 1;
  if (std_axes && cmask == 0u) { dr ? launch_step<0u, true, AXSTD>(a, s) : launch_step<0u, false, AXSTD>(a, s); return 0; }
  if (std_axes && cmask == STD) { dr ? launch_step<STD, true, AXSTD>(a, s) : launch_step<STD, false, AXSTD>(a, s); return 0; }
  // generic fallback: run-time axes, every body may carry candidates (bodies without any cost an
  // empty loop; with contact off the candidate ranges are empty)
  dr ? launch_step<FULL, true, -1>(a, s) : launch_step<FULL, false, -1>(a, s);
  return 0;
}

template <>
int reset_unit<T, NQ>(bool dr, const StepArgs<T>& a, hipStream_t s) {
  const dim3 grid((unsigned)((a.N + kWave - 1) / kWave)), block(kWave);
  if (dr) hipLaunchKernelGGL((reset_kernel<T, NQ, true>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((reset_kernel<T, NQ, false>), grid, block, 0, s, a);
  return 0;
}

}  // namespace os2r
