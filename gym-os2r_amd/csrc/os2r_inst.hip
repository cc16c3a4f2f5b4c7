// os2r_inst.hip — one instantiation unit of the step / reset kernels.
// Compiled once per (OS2R_REAL, OS2R_NQ) so the eight units build in parallel.
//
// Contact masks instantiated per chain length (bit b: body b carries ground-contact candidates):
//   0      contact off (bring-up configuration C2) or a model that cannot reach the ground
//   STD    the mask of the reference's URDF variant with that many dofs
//   FULL   every body (fallback for user-supplied models; empty bodies cost an empty loop)
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

template <unsigned CM, bool DR>
static void launch_step(const StepArgs<T>& a, hipStream_t s) {
  const dim3 grid((unsigned)((a.N + kWave - 1) / kWave)), block(kWave);
  hipLaunchKernelGGL((step_kernel<T, NQ, CM, DR>), grid, block, 0, s, a);
}

template <typename R, int N_>
int step_unit(unsigned cmask, bool dr, const StepArgs<R>& a, hipStream_t s);
template <typename R, int N_>
int reset_unit(bool dr, const StepArgs<R>& a, hipStream_t s);

template <>
int step_unit<T, NQ>(unsigned cmask, bool dr, const StepArgs<T>& a, hipStream_t s) {
  if (cmask == 0u) { dr ? launch_step<0u, true>(a, s) : launch_step<0u, false>(a, s); return 0; }
  if (cmask == STD) { dr ? launch_step<STD, true>(a, s) : launch_step<STD, false>(a, s); return 0; }
  if constexpr (STD != FULL) {
    if ((cmask & ~FULL) == 0u) { dr ? launch_step<FULL, true>(a, s) : launch_step<FULL, false>(a, s); return 0; }
  } else {
    if ((cmask & ~FULL) == 0u) { dr ? launch_step<FULL, true>(a, s) : launch_step<FULL, false>(a, s); return 0; }
  }
  return 1;
}

template <>
int reset_unit<T, NQ>(bool dr, const StepArgs<T>& a, hipStream_t s) {
  const dim3 grid((unsigned)((a.N + kWave - 1) / kWave)), block(kWave);
  if (dr) hipLaunchKernelGGL((reset_kernel<T, NQ, true>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((reset_kernel<T, NQ, false>), grid, block, 0, s, a);
  return 0;
}

}  // namespace os2r
