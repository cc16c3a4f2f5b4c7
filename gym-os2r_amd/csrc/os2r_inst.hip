// os2r_inst.hip — one instantiation unit of the step kernels; compiled once per
// (OS2R_REAL, OS2R_UNIT) so the units build in parallel.
//
//   OS2R_UNIT = 0..3   static model of os2r_models_gen.hpp (monopod, -fixed_hip, -fixed, -simple):
//                      robot constants folded into the code; contact {on, off} x DR {on, off}
//   OS2R_UNIT = 12..15 run-time model with 2..5 dofs (any compiled serial chain; every body may
//                      carry candidates, joint axes read from the model); also the reset kernels
#include "os2r_kernels.hpp"

#ifndef OS2R_REAL
#error "OS2R_REAL must be float or double"
#endif
#ifndef OS2R_UNIT
#error "OS2R_UNIT must be 0..3 (static model id) or 12..15 (run-time model, nq = unit - 10)"
#endif

namespace os2r {

using T = OS2R_REAL;

// Observation layouts of the reference's task modes on this unit's robot (tasks/monopod.py:105-200 evaluated
// on models/config/default/settings.yaml; {kind OS2R_OBS_*, source dof} per slot): the step kernel exists with
// each of them folded in (default sweep counts); any other layout runs the generic one.
#define OS2R_LAYOUT(NAME, D, ...)                                                                   \
  struct NAME##_t { static constexpr int k[D][2] = {__VA_ARGS__}; };                                 \
  constexpr unsigned long long NAME##_pack(int col) {                                                \
    int v[D] = {};                                                                                   \
    for (int i = 0; i < D; ++i) v[i] = NAME##_t::k[i][col];                                          \
    return pack_layout(v, D);                                                                        \
  }                                                                                                  \
  using NAME = StLayout<NAME##_pack(0), NAME##_pack(1), D>;
#if OS2R_UNIT == 0   // monopod: free_hip
OS2R_LAYOUT(LayA, 10, {0, 3}, {1, 4}, {0, 1}, {1, 0}, {0, 2}, {2, 3}, {2, 4}, {2, 1}, {2, 0}, {2, 2})
#elif OS2R_UNIT == 1  // monopod-fixed_hip: fixed_hip, fixed_hip_simple (the default env id)
OS2R_LAYOUT(LayA, 8, {0, 2}, {1, 3}, {0, 1}, {1, 0}, {2, 2}, {2, 3}, {2, 1}, {2, 0})
OS2R_LAYOUT(LayB, 5, {0, 2}, {1, 3}, {0, 1}, {2, 2}, {2, 3})
#elif OS2R_UNIT == 2  // monopod-fixed: fixed
OS2R_LAYOUT(LayA, 6, {0, 1}, {1, 2}, {0, 0}, {2, 1}, {2, 2}, {2, 0})
#elif OS2R_UNIT == 3  // monopod-simple: simple
OS2R_LAYOUT(LayA, 4, {0, 0}, {1, 1}, {2, 0}, {2, 1})
#endif

template <typename LAY>
static bool layout_is(const StepArgs<T>& a) {
  return a.layout_dim == LAY::kDim && a.layout_kinds == LAY::kKinds && a.layout_srcs == LAY::kSrcs;
}

// fp64 only: the counting variants (os2r_set_work_counters) of the layout kernels
[[maybe_unused]] constexpr bool kHaveCounting = sizeof(T) == 8;

#define OS2R_LAUNCH(STD, ...) hipLaunchKernelGGL((step_kernel<T, MD, CONTACT, DR, STD, ##__VA_ARGS__>), grid, block, 0, s, a)

template <typename MD, bool CONTACT, bool DR>
static int launch_step(const StepArgs<T>& a, hipStream_t s) {
  const dim3 grid((unsigned)((a.N + kWave - 1) / kWave)), block(kWave);
  // os2r_rollout (rollout_steps > 0): the fused variants -- the compiled-in robots with ground contact, the default solver
  // settings and a reference task layout; 2 = none here, the caller steps launch by launch
  if (a.rollout_steps > 0) {
#if OS2R_UNIT < 10
    if constexpr (MD::kStatic && CONTACT) {
      if (!a.counters && is_std_solver<T>(a.pgs_iters, a.pgs_normal_iters, a.pgs_exact, MD::NQ)) {
        constexpr int kSolver = std_solver(true, sizeof(T) == 8, StdSolver<T>::kExact);
        if (layout_is<LayA>(a)) { hipLaunchKernelGGL((step_kernel<T, MD, CONTACT, DR, true, LayA, false, kSolver, true>), grid, block, 0, s, a); return 0; }
#if OS2R_UNIT == 1
        if (layout_is<LayB>(a)) { hipLaunchKernelGGL((step_kernel<T, MD, CONTACT, DR, true, LayB, false, kSolver, true>), grid, block, 0, s, a); return 0; }
#endif
      }
    }
#endif
    return 2;
  }
  // the compiled-in robots also exist with the default sweep counts as compile-time loop bounds
  if (MD::kStatic && is_std_solver<T>(a.pgs_iters, a.pgs_normal_iters, a.pgs_exact, MD::NQ)) {
#if OS2R_UNIT < 10
    {
      if (layout_is<LayA>(a)) {
        if constexpr (kHaveCounting && CONTACT) {
          if (a.counters) { hipLaunchKernelGGL((step_kernel<T, MD, CONTACT, DR, MD::kStatic, LayA, true>), grid, block, 0, s, a); return 0; }
        }
        if (a.counters) return 1;
        OS2R_LAUNCH(MD::kStatic, LayA);
        return 0;
      }
#if OS2R_UNIT == 1
      if (layout_is<LayB>(a)) {
        if constexpr (kHaveCounting && CONTACT) {
          if (a.counters) { hipLaunchKernelGGL((step_kernel<T, MD, CONTACT, DR, MD::kStatic, LayB, true>), grid, block, 0, s, a); return 0; }
        }
        if (a.counters) return 1;
        OS2R_LAUNCH(MD::kStatic, LayB);
        return 0;
      }
#endif
    }
#endif
    if (a.counters) return 1;
    if constexpr (MD::kStatic) { OS2R_LAUNCH(true); }     // (run-time models never get here: nothing to instantiate for them)
  } else {
    if (a.counters) return 1;
    // other solver settings: a kernel per solver (fp64: exact finish or sweeps only; fp32: sweeps only)
    if constexpr (sizeof(T) == 8) {
      if (a.pgs_exact > 0) hipLaunchKernelGGL((step_kernel<T, MD, CONTACT, DR, false, RtLayout, false, kSolverExact>), grid, block, 0, s, a);
      else hipLaunchKernelGGL((step_kernel<T, MD, CONTACT, DR, false, RtLayout, false, kSolverSweeps>), grid, block, 0, s, a);
    } else {
      OS2R_LAUNCH(false);
    }
  }
  return 0;
}

template <typename R, int UNIT>
int step_unit(bool contact, bool dr, const StepArgs<R>& a, hipStream_t s);
template <typename R, int N_>
int reset_unit(bool dr, const StepArgs<R>& a, hipStream_t s);

#if OS2R_UNIT < 10
using MD = StModel<T, OS2R_UNIT>;
#else
using MD = RtModel<T, OS2R_UNIT - 10>;
#endif

template <>
int step_unit<T, OS2R_UNIT>(bool contact, bool dr, const StepArgs<T>& a, hipStream_t s) {
  if (MD::CMASK == 0u) contact = false;   // a chain that cannot reach the ground
  if (contact) return dr ? launch_step<MD, true, true>(a, s) : launch_step<MD, true, false>(a, s);
  return dr ? launch_step<MD, false, true>(a, s) : launch_step<MD, false, false>(a, s);
}

#if OS2R_UNIT >= 10
template <>
int reset_unit<T, OS2R_UNIT - 10>(bool dr, const StepArgs<T>& a, hipStream_t s) {
  constexpr int NQ = OS2R_UNIT - 10;
  const dim3 grid((unsigned)((a.N + kWave - 1) / kWave)), block(kWave);
  if (dr) hipLaunchKernelGGL((reset_kernel<T, NQ, true>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((reset_kernel<T, NQ, false>), grid, block, 0, s, a);
  return 0;
}
#endif

}  // namespace os2r
