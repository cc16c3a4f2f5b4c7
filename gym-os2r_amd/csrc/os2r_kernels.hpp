// os2r_kernels.hpp — step / reset kernels: substeps + fused epilogue (observation, reward, done,
// auto-reset) in one launch.  Included by the per-dtype instantiation units.
#pragma once
#include "os2r_device.hpp"

namespace os2r {

// ----------------------------------------------------------------------------------------
// epilogue pieces.  Contraction is switched off so that the f64 path performs exactly the
// operations numpy performs in the reference (tasks/monopod.py:257-272, rewards/*.py): the
// affine maps, the periodic wrap and the polynomial sigmoids are then bit-identical.
// ----------------------------------------------------------------------------------------
// tanh for the velocity observations (tasks/monopod.py:270).  The library routine is 165 instructions and the
// epilogue calls it once per observed velocity; this one is ~85: below 0.55 the Taylor series of tanh(x)/x in
// x^2 (19 terms: truncation 5e-19 relative), above it 1 - 2/(exp(2|x|) + 1), whose result lies in [0.5, 1] so
// that the errors of exp, add and divide stay below two spacings.  Within 2 ulp of glibc's tanh on 2e8 arguments
// (tests allow 4).
__device__ __forceinline__ double tanh_t(double x) {
  constexpr double C[19] = {
      -0x1.5555555555555p-2, 0x1.1111111111111p-3, -0x1.ba1ba1ba1ba1cp-5, 0x1.664f4882c10fap-6,
      -0x1.226e355e6c23dp-7, 0x1.d6d3d0e157de0p-9, -0x1.7da36452b75e3p-10, 0x1.3558248036744p-11,
      -0x1.f57d7734d1664p-13, 0x1.967e18afcafadp-14, -0x1.497d8eea25259p-15, 0x1.0b132d39a6050p-16,
      -0x1.b0f72d3ee24e9p-18, 0x1.5ef2da474e5b7p-19, -0x1.1c77df95c1c0dp-20, 0x1.cd299de4ae6bbp-22,
      -0x1.75cde6563fed9p-23, 0x1.2efe8db3aff1fp-24, -0x1.eb3229047434cp-26};
  const double ax = __builtin_fabs(x);
  const bool small = ax < 0.55;
  double r = x;
  if (__ballot(small) != 0ull) {
    const double z = x * x;
    double p = C[18];
#pragma unroll
    for (int i = 17; i >= 0; --i) p = __builtin_fma(p, z, C[i]);
    r = __builtin_fma(x, z * p, x);
  }
  if (__ballot(!small) != 0ull) {
    const double t = exp(2.0 * ax);                 // inf for |x| > 354: the quotient is 0, the result 1
    const double big = __builtin_copysign(1.0 - 2.0 / (t + 1.0), x);
    r = small ? r : big;
  }
  return r;
}
__device__ __forceinline__ float tanh_t(float x) { return tanhf(x); }
__device__ __forceinline__ double fmod_t(double a, double b) { return fmod(a, b); }
__device__ __forceinline__ float fmod_t(float a, float b) { return fmodf(a, b); }
__device__ __forceinline__ double atanh_t(double x) { return atanh(x); }
__device__ __forceinline__ float atanh_t(float x) { return atanhf(x); }

// numpy.mod(x + pi, 2*pi) - pi
template <typename T>
__device__ __forceinline__ T wrap_pi(T x) {
#pragma clang fp contract(off)
  const T pi = T(3.141592653589793238462643383279502884);
  const T two_pi = T(2) * pi;
  T m = fmod_t(x + pi, two_pi);
  if (m != T(0)) {
    if (m < T(0)) m += two_pi;
  } else {
    m = T(0);
  }
  return m - pi;
}

// value before the affine / tanh map, and the observation itself (tasks/monopod.py:238-272)
// Observation layout: which quantity feeds which slot and through which map.  It is data of the task
// (RtLayout: read from the task struct), and for the layouts of the reference's task modes on the compiled-in
// robots also a compile-time variant (StLayout): the slot loop below then folds into straight-line code --
// no per-slot dispatch on the kind, no select chains over the joints (they were ~500 cycles per slot for a
// wave that owns its SIMD: the whole observation 10.7 k cycles per env-step against 3 k of arithmetic).
struct RtLayout {
  static constexpr bool kStatic = false;
  static constexpr int kDim = 0;
  static constexpr int kind(int) { return 0; }
  static constexpr int src(int) { return 0; }
};
template <unsigned long long KINDS, unsigned long long SRCS, int DIM>
struct StLayout {
  static constexpr bool kStatic = true;
  static constexpr unsigned long long kKinds = KINDS, kSrcs = SRCS;
  static constexpr int kDim = DIM;
  static constexpr int kind(int d) { return (int)((KINDS >> (4 * d)) & 15ull); }
  static constexpr int src(int d) { return (int)((SRCS >> (4 * d)) & 15ull); }
};
constexpr unsigned long long pack_layout(const int* v, int n) {
  unsigned long long r = 0;
  for (int i = 0; i < n; ++i) r |= (unsigned long long)(v[i] & 15) << (4 * i);
  return r;
}

template <typename T, int NQ, typename LAY = RtLayout>
__device__ __forceinline__ void observe(TaskPtr<T> ts, const T (&q)[NQ], const T (&qd)[NQ],
                                        T h1a, T h1b, T (&obs)[OS2R_MAX_OBS], bool& done, unsigned& reason) {
#pragma clang fp contract(off)
  done = false;
  reason = 0u;   // bit d: observation slot d is outside the reset space (or not finite): which one ended the episode
  int D = LAY::kStatic ? LAY::kDim : ts->obs_dim;
  if constexpr (!LAY::kStatic) asm volatile("" : "+s"(D));   // one copy in a register: not re-fetched (and waited for) slot after slot
  // The slots' constants are fetched four slots at a time, ahead of the slots' arithmetic: left to the
  // point of use, every slot waits for five scalar loads in turn, which a lone wave cannot hide.
  constexpr int H = 4;
  static_assert(OS2R_MAX_OBS % H == 0, "whole batches");
#pragma unroll
  for (int batch = 0; batch < OS2R_MAX_OBS / H; ++batch) {
    if (LAY::kStatic && batch * H >= LAY::kDim) {
#pragma unroll
      for (int k = 0; k < H; ++k) obs[batch * H + k] = T(0);
      continue;
    }
    int kind_[H], src_[H];
    T lo_[H], hi_[H], dlo_[H], dhi_[H];
#pragma unroll
    for (int k = 0; k < H; ++k) {
      const int d = batch * H + k;
      kind_[k] = LAY::kStatic ? LAY::kind(d) : ts->obs_kind[d];
      src_[k] = LAY::kStatic ? LAY::src(d) : ts->obs_src[d];
      lo_[k] = ts->obs_low[d]; hi_[k] = ts->obs_high[d];
      dlo_[k] = ts->done_lo[d]; dhi_[k] = ts->done_hi[d];
    }
#pragma unroll
    for (int k = 0; k < H; ++k) {   // used here: the loads are not sunk into the branches below
      if constexpr (!LAY::kStatic) asm volatile("" : "+s"(kind_[k]), "+s"(src_[k]));
      asm volatile("" : "+s"(lo_[k]), "+s"(hi_[k]), "+s"(dlo_[k]), "+s"(dhi_[k]));
    }
#pragma unroll
    for (int k = 0; k < H; ++k) {
      const int d = batch * H + k;
      if (d >= D) { obs[d] = T(0); continue; }
      const int kind = kind_[k], s = src_[k];
      T x = T(0);
      if (kind == OS2R_OBS_TORQUE_NORM || kind == OS2R_OBS_TORQUE_RAW) {
        x = s == 0 ? h1a : h1b;
      } else if (kind == OS2R_OBS_VEL_TANH || kind == OS2R_OBS_VEL_RAW) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) x = (s == i) ? qd[i] : x;
      } else {
#pragma unroll
        for (int i = 0; i < NQ; ++i) x = (s == i) ? q[i] : x;
      }
      if (kind == OS2R_OBS_POS_PERIODIC_NORM || kind == OS2R_OBS_POS_PERIODIC_RAW) x = wrap_pi(x);
      // done: the reference tests the observation against reset_space; done_lo/done_hi are the
      // exact pre-images of that test on x (host-side bisection), NaN counts as done
      if (x < dlo_[k] || x > dhi_[k] || !finite_t(x)) { done = true; reason |= 1u << d; }   // NaN counts as done
      T o = x;
      if (kind == OS2R_OBS_POS_NORM || kind == OS2R_OBS_POS_PERIODIC_NORM || kind == OS2R_OBS_TORQUE_NORM) {
        const T lo = lo_[k], hi = hi_[k];
        o = T(2) * (x - lo) / (hi - lo) - T(1);
      } else if (kind == OS2R_OBS_VEL_TANH) {
        o = tanh_t(T(0.05) * x);
      }
      obs[d] = o;
    }
  }
}

// rewards_utils.py:76-122 restricted to the sigmoids the reward classes use
template <typename T>
__device__ __forceinline__ T tol_quadratic(T x, T margin, T value_at_margin) {  // bounds (0,0)
#pragma clang fp contract(off)
  if (x == T(0)) return T(1);
  const T d = (x < T(0) ? T(0) - x : x - T(0)) / margin;
  const T sx = d * sqrt_t(T(1) - value_at_margin);
  return fabs_t(sx) < T(1) ? T(1) - sx * sx : T(0);
}
template <typename T>
__device__ __forceinline__ T tol_linear(T x, T margin, T value_at_margin) {  // bounds (0,0)
#pragma clang fp contract(off)
  if (x == T(0)) return T(1);
  const T d = (x < T(0) ? T(0) - x : x - T(0)) / margin;
  const T sx = d * (T(1) - value_at_margin);
  return fabs_t(sx) < T(1) ? T(1) - sx : T(0);
}

template <typename T>
__device__ __forceinline__ T pick_obs(const T (&obs)[OS2R_MAX_OBS], int idx) {
  T x = T(0);
#pragma unroll
  for (int d = 0; d < OS2R_MAX_OBS; ++d) x = (d == idx) ? obs[d] : x;
  return x;
}

// rewards/__init__.py:66-207.  a0 = actions[0] (just applied), a1 = actions[1] (previous)
template <typename T>
__device__ __forceinline__ T reward_of(TaskPtr<T> ts, const T (&obs)[OS2R_MAX_OBS], T a0x, T a0y,
                                       T a1x, T a1y) {
#pragma clang fp contract(off)
  const T nrm = ts->normalized ? T(1) : T(0);
  const T H = T(0.11) / T(1.57) * nrm + T(0.11) * (T(1) - nrm);
  const T bp = pick_obs(obs, ts->idx_pitch_pos);
  const T lo = H, hi = T(4) * H;
  const bool inb = (lo <= bp) && (bp <= hi);
  const T window = inb ? T(1) : T(0);
  switch (ts->reward_id) {
    case OS2R_REWARD_BALANCING_V1:
    case OS2R_REWARD_STANDING_V1:
      return window;
    case OS2R_REWARD_BALANCING_V2:
      return window * (tol_quadratic(a0x, T(1), T(0.4)) * tol_quadratic(a0y, T(1), T(0.4)));
    case OS2R_REWARD_BALANCING_V3: {
      T bal = T(1);
      if (!inb) {  // long_tail, margin 0.01, value_at_margin 0.1
        const T d = (bp < lo ? lo - bp : bp - hi) / T(0.01);
        const T t = d * sqrt_t(T(1) / T(0.1) - T(1));
        bal = T(1) / (t * t + T(1));
      }
      return bal * (tol_quadratic(a0x - a1x, T(1), T(0.1)) * tol_quadratic(a0y - a1y, T(1), T(0.1)));
    }
    case OS2R_REWARD_HOPPING_V1: {
      const T sd = tol_quadratic(a0x - a1x, T(0.1), T(0)) * tol_quadratic(a0y - a1y, T(0.1), T(0));
      const T hv = pick_obs(obs, ts->idx_yaw_vel);
      T move = T(1);
      if (!((T(0.25) <= hv) && (hv <= T(0.3)))) {  // tanh_squared, margin 0.15, value_at_margin 0.1
        const T d = (hv < T(0.25) ? T(0.25) - hv : hv - T(0.3)) / T(0.15);
        const T th = tanh_t(d * atanh_t(sqrt_t(T(1) - T(0.1))));
        move = T(1) - th * th;
      }
      return window * sd * move;
    }
    case OS2R_REWARD_STRAIGHT_V1: {
      T sc = (tol_quadratic(a0x / T(20), T(1), T(0)) + tol_quadratic(a0y / T(20), T(1), T(0))) / T(2);
      sc = (T(4) + sc) / T(5);
      const T hr = tol_linear(pick_obs(obs, ts->idx_hip_pos), T(1), T(0.1));
      const T kr = tol_linear(pick_obs(obs, ts->idx_knee_pos), T(1), T(0.1));
      return hr * kr * sc;
    }
    default:
      return T(0);
  }
}

// utils/reset.py:4-40
__device__ __forceinline__ void leg_joint_angles(const double (&def6)[6], double pitch, double& hip, double& knee) {
#pragma clang fp contract(off)
  const double ul = def6[0], ll = def6[1], cph = def6[2], lb = def6[3];
  const double lh = (lb * sin(pitch) + cph) / cos(pitch);
  const double lleg = lh - def6[4] - def6[5];
  if (lleg > ul + ll) { hip = 0.0; knee = 0.0; return; }
  const double ua = acos((ul * ul + lleg * lleg - ll * ll) / (2.0 * ul * lleg));
  const double la = asin(ul * sin(ua) / ll) + ua;
  hip = ua;
  knee = -la;
}

// Reset of one environment: randomizers/monopod_no_rand.py:59-98 (fixed) and
// randomizers/monopod.py:89-128,182-215 (randomised pose + parameter resampling).
// freshly sampled per-env parameters of a reset (written back by store_params)
template <typename T, int NQ>
struct ParamVals {
  T ms[NQ], dm[NQ], fr[NQ], mu_[NQ];
  bool fresh = false;
  T gravity = T(0);          // drawn anew after every `gravity_rollouts` rollouts of the environment
  bool gravity_fresh = false;
};

template <typename T, typename MD, bool DR>
__device__ __forceinline__ void reset_env(const StepArgs<T>& A, long long e, uint32_t epi, T (&q)[MD::NQ], T (&qd)[MD::NQ],
                                          ParamVals<T, MD::NQ>& par, uint8_t& pose) {
#pragma clang fp contract(off)
  constexpr int NQ = MD::NQ;
  const TaskPtr<T> ts = as_const(A.task);
  const uint32_t genv = (uint32_t)(A.env_offset + e);
  double u0, u1;
  uniform2(A.seed, genv, kStreamReset, epi, 0, u0, u1);
  int pi = (int)(u0 * ts->n_reset_poses);
  if (pi >= ts->n_reset_poses) pi = ts->n_reset_poses - 1;
  double pitch = 0, hip = 0, knee = 0, yaw = 0;
  int laying = 0, pid = 0;
  for (int k = 0; k < OS2R_MAX_RESET_POSES; ++k) {
    if (k == pi) { pitch = ts->reset_pitch[k]; hip = ts->reset_hip[k]; knee = ts->reset_knee[k]; laying = ts->reset_laying[k]; pid = ts->reset_pose_id[k]; }
  }
  if (ts->reset_mode == OS2R_RESET_FIXED) {
    if (ts->reset_simple) {
      double w0, w1;
      uniform2(A.seed, genv, kStreamReset, epi, 1, w0, w1);
      hip = 2.0 * w0 - 1.0;
      knee = 2.0 * w1 - 1.0;
    }
  } else {
    pitch *= 0.8 + 0.4 * u1;
    double z0, z1, w0, w1, x0, x1;
    normal2(A.seed, genv, kStreamReset, epi, 1, z0, z1);
    uniform2(A.seed, genv, kStreamReset, epi, 2, w0, w1);
    uniform2(A.seed, genv, kStreamReset, epi, 3, x0, x1);
    const double r0 = fabs(0.2 * z0), r1 = fabs(0.2 * z1);
    const double rmax = r0 > r1 ? r0 : r1, rmin = r0 > r1 ? r1 : r0;
    if (!laying) {
      double def6[6];
      for (int k = 0; k < 6; ++k) def6[k] = ts->leg_def[k];
      leg_joint_angles(def6, pitch, hip, knee);
    } else {
      hip = 1.57 - (w0 < 0.5 ? 3.14 : 0.0);
      knee = 0.0;
    }
    hip = hip + (hip > 0.0 ? 1.0 : 0.0) * rmax;
    knee = knee - (knee > 0.0 ? 1.0 : 0.0) * rmin;
    const double dir = 1.0 - (w1 < 0.5 ? 2.0 : 0.0);
    hip *= dir;
    knee *= dir;
    yaw = -0.2 + 0.4 * x0;
  }
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    double v = 0.0;
    if (i == ts->dof_pitch) v = pitch;
    if (i == ts->dof_yaw) v = yaw;
    if (i == ts->dof_hip) v = hip;
    if (i == ts->dof_knee) v = knee;
    q[i] = (T)v;
    qd[i] = T(0);
  }
  pose = (uint8_t)pid;
  if constexpr (DR) {
    if (ts->reset_mode == OS2R_RESET_RANDOM && ts->randomize_params) {
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        double a0, a1, b0, b1;
        uniform2(A.seed, genv, kStreamParams, epi, 2 * i, a0, a1);
        uniform2(A.seed, genv, kStreamParams, epi, 2 * i + 1, b0, b1);
        par.ms[i] = (T)(ts->dr_mass_lo + (ts->dr_mass_hi - ts->dr_mass_lo) * a0);
        par.fr[i] = (T)(ts->dr_friction_lo + (ts->dr_friction_hi - ts->dr_friction_lo) * a1);
        par.dm[i] = (T)(ts->nominal_damping[i] * (ts->dr_damping_lo + (ts->dr_damping_hi - ts->dr_damping_lo) * b0));
        par.mu_[i] = (T)(ts->dr_mu_base * (ts->dr_mu_lo + (ts->dr_mu_hi - ts->dr_mu_lo) * b1));
      }
      par.fresh = true;
    }
    // randomize_physics of the reference runs when its simulator is re-created, every `num_physics_rollouts`
    // rollouts (randomizers/monopod.py:36-41,56-61): `epi` rollouts of this environment are over now
    if (ts->reset_mode == OS2R_RESET_RANDOM && ts->gravity_rollouts > 0 && epi > 0u && epi % (uint32_t)ts->gravity_rollouts == 0u) {
      double z0, z1;
      normal2(A.seed, genv, kStreamGravity, epi / (uint32_t)ts->gravity_rollouts, 0, z0, z1);
      par.gravity = (T)(ts->dr_gravity_mean + ts->dr_gravity_std * z0);
      par.gravity_fresh = true;
    }
  }
}

// Binds the per-lane parameters of one env-step: nominal ones are compile-time / scalar constants; randomised ones are
// fetched from their HBM arrays into this lane's LDS slots (Params<.., true>), once per env-step.
template <typename T, typename MD, bool DR>
__device__ __forceinline__ void bind_params(const StepArgs<T>& A, long long e, const MD& md, Params<T, MD, DR>& par, T* slots) {
  par.m = md;
  if constexpr (!DR) par.g = A.gravity_z;
  if constexpr (DR) {
    constexpr int NQ = MD::NQ;
    par.g = A.gravity[e];
    // joint by joint: four values in flight, not twenty registers held at once (this runs where the state has just been loaded)
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const T ms = A.mass_scale[i * A.N + e], dm = A.damping[i * A.N + e], fr = A.friction[i * A.N + e], mu = A.mu[i * A.N + e];
      slots[(0 * NQ + i) * kWave] = md.mass(i) * ms;
      slots[(1 * NQ + i) * kWave] = dm;
      slots[(2 * NQ + i) * kWave] = fr;
      slots[(3 * NQ + i) * kWave] = mu;
    }
    par.slots = slots;
  }
}
template <typename T, int NQ>
__device__ __forceinline__ void store_params(const StepArgs<T>& A, long long e, const ParamVals<T, NQ>& pv) {
  if (pv.fresh) {
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      A.mass_scale[i * A.N + e] = pv.ms[i];
      A.damping[i * A.N + e] = pv.dm[i];
      A.friction[i * A.N + e] = pv.fr[i];
      A.mu[i * A.N + e] = pv.mu_[i];
    }
  }
  if (pv.gravity_fresh) A.gravity[e] = pv.gravity;
}

// Coalesced [N][D] row-major store of one wave's observation tile: the 64 rows of a wave are
// one contiguous block of 64*D values, so the tile goes through LDS and is written in lane order.
template <typename T>
__device__ __forceinline__ void store_obs_tile(T* __restrict__ dst, const T (&obs)[OS2R_MAX_OBS], int D, long long e0,
                                               long long N, int lane, T* tile) {
#pragma unroll
  for (int d = 0; d < OS2R_MAX_OBS; ++d)
    if (d < D) tile[lane * D + d] = obs[d];
  __syncthreads();
  const long long rows = (N - e0) < (long long)kWave ? (N - e0) : (long long)kWave;
  const int total = (int)rows * D;
  for (int k = lane; k < total; k += kWave) dst[e0 * D + k] = tile[k];
  __syncthreads();
}

// ----------------------------------------------------------------------------------------
// env-step kernel: GazeboRuntime.step (runtimes/gazebo_runtime.py:65-97) for every env
// ----------------------------------------------------------------------------------------
// LDS of one wave: during the physics iterations it holds the Minv*J^T rows of the contact
// problem (3 rows x NQ values per body, one slot per lane); afterwards the same storage is the
// observation tile of the coalesced [N][D] store.
template <int NQ>
constexpr int lds_words() {
  // Cholesky factor of Minv / per-joint ABA data (U, 1/D, u)
  constexpr int rows = NQ * (NQ + 1) / 2, aba = 8 * NQ;
  constexpr int m = rows > aba ? rows : aba;
  return kWave * (m > OS2R_MAX_OBS ? m : OS2R_MAX_OBS);
}
// wave-shared copy of the contact candidate table, behind the per-lane slots
constexpr int kCandWords = 3 * OS2R_MAX_CAND;

template <typename T, typename MD>
__device__ __forceinline__ MD make_model(const StepArgs<T>& A) {
  if constexpr (MD::kStatic) return MD{};
  else return MD{as_const(A.model)};
}

// One env-step of this wave's 64 environments: the body of every step kernel (the templated ones below and
// the model-specialised ones of os2r_jit_unit.hip).  STD_SWEEPS: the solver's sweep counts are the default
// ones and known at compile time (the launcher checks the handle's configuration).
// COUNT: the counting variant (os2r_set_work_counters): same arithmetic, and the wave's work of the launch is added to
// A.counters[0..kWorkCounters) by one lane at the end.
constexpr int kWorkCounters = OS2R_NUM_WORK_COUNTERS;
// SOLVER (os2r_device.hpp): kSolverBoth unless the launcher knows the handle's solver
constexpr int std_solver(bool std_sweeps, bool is_f64, bool std_exact) {
  return !is_f64 ? kSolverSweeps : (std_sweeps ? (std_exact ? kSolverExact : kSolverSweeps) : kSolverBoth);
}
// ROLLOUT: A.rollout_steps env-steps in this launch (os2r_rollout): every wave advances its own 64 environments step
// after step -- no device-wide barrier between the env-steps, where one launch per env-step makes 1024 waves wait for
// the slowest -- with the state in registers from the first step to the last; step k writes its observation, reward,
// done flag (terminal observation, done reason) at offset k of the [K][N]... output arrays and takes its actions from
// slice k of the [K][N][2] input (or draws them with the step counter + k).  The arithmetic of a step is the same code.
template <typename T, typename MD, bool CONTACT, bool DR, bool STD_SWEEPS = false, typename LAY = RtLayout, bool COUNT = false,
          int SOLVER = std_solver(STD_SWEEPS, sizeof(T) == 8, StdSolver<T>::kExact), bool ROLLOUT = false>
__device__ __forceinline__ void step_body(const StepArgs<T>& A) {
  constexpr int NQ = MD::NQ;
  // run-time models scan a wave-shared LDS copy of the candidate table; the compiled-in ones read the
  // (wave-uniform) coordinates through scalar loads, straight into the operands of the scan
  constexpr bool kCandInLds = CONTACT && MD::CMASK != 0u && !MD::kStatic;
#ifdef OS2R_STAMPS
  const unsigned long long stamp_entry = __builtin_amdgcn_s_memtime();
  const unsigned long long real_entry = __builtin_amdgcn_s_memrealtime();
#endif
  // [per-lane slots of the physics iteration | per-lane parameter slots (DR) | wave-shared candidate table (run-time models)]
  constexpr int kParamWords = DR ? 4 * NQ * kWave : 0;
  __shared__ T tile[lds_words<NQ>() + kParamWords + (kCandInLds ? kCandWords : 0)];
  const int lane = threadIdx.x;
  const long long e0 = (long long)blockIdx.x * kWave;
  const bool valid = e0 + lane < A.N;
  const long long e_lane = valid ? e0 + lane : A.N - 1;  // tail lanes shadow the last env, stores are masked
  const MD md = make_model<T, MD>(A);
  T* cand_lds = tile + lds_words<NQ>() + kParamWords;
  if constexpr (kCandInLds) {
    const int nc3 = 3 * md.cand_begin(NQ);
    for (int k = lane; k < nc3; k += kWave) cand_lds[k] = md.cand(k / 3, k % 3);
    __syncthreads();
  }
  const TaskPtr<T> ts_launch = as_const(A.task);
  const int D = LAY::kStatic ? LAY::kDim : ts_launch->obs_dim;
  // The host's view of the clamped-action count (os2r_get_violation_mirror): the first wave of a launch copies the count that
  // every EARLIER launch left -- they have all finished: stream order -- and this launch's step counter into two words of mapped
  // host memory, so a host binding can look at it without a copy, an event or a wait on the step path.
  if (blockIdx.x == 0 && lane == 0 && A.mirror) {
    const unsigned seen = __hip_atomic_load(A.violations, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(A.mirror, seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(A.mirror + 1, (uint32_t)A.step_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
#ifdef OS2R_STAMPS
  unsigned long long stamps[kStamps] = {}, stamp_prev = __builtin_amdgcn_s_memtime();
  stamps[10] = stamp_prev - stamp_entry;   // prologue: loads, action, torques
#endif
  T sn[NQ], cs[NQ];   // sin/cos of the joint angles, carried from one physics iteration to the next
  WorkCounts wc;
  const int nsteps = ROLLOUT ? A.rollout_steps : 1;
  for (int k = 0; k < nsteps; ++k) {
    const long long ko = ROLLOUT ? (long long)k * A.N : 0ll;   // this step's slice of the output arrays
    // (a rollout passes the state from one env-step to the next through memory like separate launches do -- its own
    // lines, L2-resident: held in registers across the epilogue it costs the 5-dof kernels 140-390 B of scratch per lane --
    // and its environment index is a fresh value in every step: otherwise the address of every state array, sixty of
    // them, is hoisted out of the step loop and held in registers across it: 450 B of scratch)
    long long e = e_lane;
    if constexpr (ROLLOUT) asm volatile("" : "+v"(e));
    TaskPtr<T> ts = ts_launch;   // (and the task's constants are fetched where they are used, step by step)
    if constexpr (ROLLOUT) asm volatile("" : "+s"(ts));
    T q[NQ], qd[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      q[i] = A.q[i * A.N + e];
      qd[i] = A.qd[i * A.N + e];
    }
    Params<T, MD, DR> par;
    bind_params<T, MD, DR>(A, e, md, par, tile + lds_words<NQ>() + lane);

    // what the exact finish hands from one physics iteration to the next (os2r_device.hpp): part of the environment's state
    constexpr bool kCarry = sizeof(T) == 8 && SOLVER != kSolverSweeps;
    SolverCarry<T, NQ> carry;
    if constexpr (kCarry) {
      carry.act = A.solver_flags[e];
#pragma unroll
      for (int b = 0; b < NQ; ++b) {
        if (CONTACT && ((MD::CMASK >> b) & 1u)) {
          carry.ln[b] = A.solver_l[(0 * NQ + b) * A.N + e];
          carry.lx[b] = A.solver_l[(1 * NQ + b) * A.N + e];
          carry.ly[b] = A.solver_l[(2 * NQ + b) * A.N + e];
        }
        carry.lf[b] = A.solver_l[(3 * NQ + b) * A.N + e];
      }
    }
    const unsigned long long step_count = A.step_count + (unsigned long long)k;
    // action: caller-provided or drawn from the counter RNG (stream 1, counter = step count)
    T ax, ay;
    if (A.actions) {
      ax = A.actions[2 * (ko + e)];
      ay = A.actions[2 * (ko + e) + 1];
      // (the lanes behind the last environment of a tail wave shadow it and must not count its action again)
      const bool outside = (ax < T(-1) || ax > T(1) || ay < T(-1) || ay > T(1)) && e0 + lane < A.N;
      if (__ballot(outside) != 0ull && outside && A.violations) atomicAdd(A.violations, 1u);
    } else {
      double u0, u1;
      uniform2(A.seed, (uint32_t)(A.env_offset + e), kStreamAction, (uint32_t)step_count, (uint32_t)(step_count >> 32), u0, u1);
      ax = (T)(2.0 * u0 - 1.0);
      ay = (T)(2.0 * u1 - 1.0);
    }
    ax = ax < T(-1) ? T(-1) : (ax > T(1) ? T(1) : ax);
    ay = ay < T(-1) ? T(-1) : (ay > T(1) ? T(1) : ay);
    T tau_hip, tau_knee, asx, asy;
    {
#pragma clang fp contract(off)
      tau_hip = md.max_torque(0) * ax;    // tasks/monopod.py:223
      tau_knee = md.max_torque(1) * ay;
      asx = tau_hip / md.max_torque(0);   // what action_history stores (:233-235)
      asy = tau_knee / md.max_torque(1);
    }

    for (int s = 0; s < A.substeps; ++s) {  // runtimes/gazebo_runtime.py:70-77
      substep<T, MD, CONTACT, DR, COUNT, SOLVER>(
                                         md, par, q, qd, sn, cs, s == 0, tau_hip, tau_knee, A.dt, A.erp, A.max_erv, A.margin,
                                         STD_SWEEPS ? std_iters<T>(NQ) : A.pgs_iters, STD_SWEEPS ? StdSolver<T>::kNormalIters : A.pgs_normal_iters,
                                         A.pgs_exact, A.pgs_tol, tile, cand_lds, as_const(A.model), wc, carry
#ifdef OS2R_STAMPS
                                         , stamps, stamp_prev
#endif
      );
    }
    __syncthreads();
    // Everything behind the physics iterations addresses the environment through a value the compiler cannot see through:
    // otherwise the addresses of the arrays stored below -- sixty 64-bit values -- are computed at the top of the kernel and
    // held in registers across the physics loop (measured: 36 AGPR copies more for the solver's state alone).
    long long ep = e;
    asm volatile("" : "+v"(ep));
    // ... and reads the kernel arguments afresh from the argument segment: held in scalar registers from the top of the kernel
    // they (two dozen pointers) are spilled to VGPR lanes across the physics loop
    const OS2R_CONST StepArgs<T>* args_e = (const OS2R_CONST StepArgs<T>*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(args_e));
    const StepArgs<T>& Ae = *(const StepArgs<T>*)args_e;
    // (This reads the argument segment at offset 0: every kernel that inlines step_body -- step_kernel below, the run-time
    // code objects of os2r_jit_unit.hip -- takes ONE argument, the StepArgs by value.  A wrapper with another leading argument
    // would read garbage here; the counting variants, which the tests and bench.py's replay run, check it and trap.)
    if constexpr (COUNT) {
      if (Ae.N != A.N || Ae.q != A.q || Ae.solver_l != A.solver_l) __builtin_trap();
    }
    // The contact solver's state goes back to HBM here, straight after the last physics iteration: held until the end of
    // the env-step it would be live across the whole epilogue -- forty registers more at the kernel's widest point.
    if constexpr (kCarry) {
      if (valid) {
        Ae.solver_flags[ep] = carry.act;
#pragma unroll
        for (int b = 0; b < NQ; ++b) {
          // (an impulse that is not remembered is stored as zero: the arrays are a function of the trajectory alone)
          const bool cb = ((carry.act >> b) & 1u) != 0u;
          Ae.solver_l[(0 * NQ + b) * Ae.N + ep] = cb ? carry.ln[b] : T(0);
          Ae.solver_l[(1 * NQ + b) * Ae.N + ep] = cb ? carry.lx[b] : T(0);
          Ae.solver_l[(2 * NQ + b) * Ae.N + ep] = cb ? carry.ly[b] : T(0);
          Ae.solver_l[(3 * NQ + b) * Ae.N + ep] = carry.lf[b];
        }
      }
    }

    bool bad = false;
#pragma unroll
    for (int i = 0; i < NQ; ++i) bad = bad || !finite_t(q[i]) || !finite_t(qd[i]) || fabs_t(q[i]) > T(1e30) || fabs_t(qd[i]) > T(1e30);

    // (the bookkeeping of an environment -- action history, step and episode counters -- is fetched where it is used and
    // stored at the end of every env-step, in a rollout as well: nothing of it is held across the physics iterations)
    const T h1x = Ae.hist[0 * Ae.N + ep], h1y = Ae.hist[1 * Ae.N + ep];  // becomes action_history[1]
    T obs[OS2R_MAX_OBS];
    bool dn;
    unsigned why;
    OS2R_STAMP(20);
    observe<T, NQ, LAY>(ts, q, qd, h1x, h1y, obs, dn, why);
    OS2R_STAMP(21);
    const T rew = reward_of<T>(ts, obs, asx, asy, h1x, h1y);
    OS2R_STAMP(22);
    int steps = Ae.steps[ep] + 1;
    const bool trunc = ts->max_episode_steps > 0 && steps >= ts->max_episode_steps;
    const uint8_t flag = (uint8_t)((dn ? 1 : 0) | (trunc ? 2 : 0) | (bad ? 4 : 0));

    if (Ae.term_obs) store_obs_tile<T>(Ae.term_obs + ko * D, obs, D, e0, Ae.N, lane, tile);

    uint32_t epi = Ae.episode[ep];
    uint8_t pose = Ae.pose[ep];
    const bool do_reset = flag != 0 && Ae.auto_reset != 0;
    ParamVals<T, NQ> pv;
    if (__ballot(do_reset) != 0ull) {
      if (do_reset) {
        reset_env<T, MD, DR>(Ae, ep, epi, q, qd, pv, pose);
        epi += 1;
        steps = 0;
        bool dn2;
        unsigned why2;
        observe<T, NQ, LAY>(ts, q, qd, h1x, h1y, obs, dn2, why2);
        // a new episode: the contact solver remembers nothing
        if constexpr (kCarry) {
          if (valid) {
            Ae.solver_flags[ep] = 0u;
#pragma unroll
            for (int k = 0; k < 4 * NQ; ++k) Ae.solver_l[k * Ae.N + ep] = T(0);
          }
        }
      }
    }
    if (Ae.obs) store_obs_tile<T>(Ae.obs + ko * D, obs, D, e0, Ae.N, lane, tile);
    OS2R_STAMP(23);
    if (valid) {
      Ae.hist[2 * Ae.N + ep] = h1x;
      Ae.hist[3 * Ae.N + ep] = h1y;
      Ae.hist[0 * Ae.N + ep] = asx;
      Ae.hist[1 * Ae.N + ep] = asy;
      Ae.steps[ep] = steps;
      if (do_reset) {
        Ae.episode[ep] = epi;
        Ae.pose[ep] = pose;
        store_params<T, NQ>(Ae, ep, pv);
      }
      if (Ae.reward) Ae.reward[ko + ep] = rew;
      if (Ae.done) Ae.done[ko + ep] = flag;
      if (Ae.reason) Ae.reason[ko + ep] = (uint16_t)why;
      if (Ae.done_mask) Ae.done_mask[ko + ep] = flag != 0 ? (uint8_t)1 : (uint8_t)0;
    }
    if (valid) {
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        Ae.q[i * Ae.N + ep] = q[i];
        Ae.qd[i * Ae.N + ep] = qd[i];
      }
    }
    if constexpr (ROLLOUT) {
      // the next step of this lane reads what this one has just written: same lane, same addresses, program order (vector
      // memory operations of a wave are performed in order); the compiler is kept from moving those loads up
      asm volatile("" ::: "memory");
    }
  }
  if constexpr (COUNT) {
    if (A.counters && lane == 0) {
      const unsigned long long v[kWorkCounters] = {(unsigned long long)A.substeps * (unsigned long long)nsteps, wc.scanned, wc.row_bodies, wc.body_sweeps,
                                                    wc.sweeps, wc.lane_contacts, wc.live_lane_sweeps, wc.full_sincos,
                                                    wc.exact_solves, wc.lane_exact_solves};
#pragma unroll
      for (int k = 0; k < kWorkCounters; ++k) atomicAdd(A.counters + k, v[k]);
    }
  }

#ifdef OS2R_STAMPS
  stamps[11] = __builtin_amdgcn_s_memtime() - stamp_prev;   // the state and flag stores, issued
  // the wave's life in shader-clock ticks and in ticks of the constant 100 MHz counter: their quotient is the clock
  // the chip held under this load (MI355X_MICROARCH.md, DVFS give-back (6))
  stamps[24] = __builtin_amdgcn_s_memtime() - stamp_entry;
  const unsigned long long real_exit = __builtin_amdgcn_s_memrealtime();
  stamps[25] = real_exit - real_entry;
#ifdef OS2R_STAMPS_LIGHT
  stamps[22] = real_entry;   // absolute: comparable between the waves of a launch
  stamps[23] = real_exit;
  {
    unsigned hw_id;            // where the wave ran: XCC_ID / SE / CU / SIMD fields of HW_ID and XCC_ID
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    stamps[21] = ((unsigned long long)xcc << 32) | hw_id;
  }
#endif
  if (A.debug && lane == 0)
    for (int k = 0; k < kStamps; ++k) A.debug[blockIdx.x * kStamps + k] = stamps[k];
#endif
}

// Registers decide how many waves share a SIMD.  fp64 state needs ~370 registers (256 VGPR + AGPR copies): one wave.
// The fp32 kernels need 200 -- hardware reciprocal / rsqrt estimates with one Newton step instead of the IEEE divide
// and sqrt expansions, a fp32 Cody-Waite sincos, no SLP vectorisation (csrc/Makefile) -- and are built for two waves
// per SIMD without scratch: the second wave takes the issue slots a lone wave leaves empty (a lone wave issues one
// VALU instruction every 4 cycles, the SIMD takes an fp32 one every 2), 1.6x from 131 072 envs per GPU on.
#define OS2R_STEP_KERNEL_ATTRS(REAL) \
  __launch_bounds__(os2r::kWave) __attribute__((amdgpu_waves_per_eu(sizeof(REAL) == 4 ? 2 : 1)))

template <typename T, typename MD, bool CONTACT, bool DR, bool STD_SWEEPS, typename LAY = RtLayout, bool COUNT = false,
          int SOLVER = std_solver(STD_SWEEPS, sizeof(T) == 8, StdSolver<T>::kExact), bool ROLLOUT = false>
__global__ OS2R_STEP_KERNEL_ATTRS(T) void step_kernel(const StepArgs<T> A) {
  step_body<T, MD, CONTACT, DR, STD_SWEEPS, LAY, COUNT, SOLVER, ROLLOUT>(A);
}

// ----------------------------------------------------------------------------------------
// masked reset kernel: GazeboEnvRandomizer.reset -> randomize_task -> get_observation
// ----------------------------------------------------------------------------------------
template <typename T, int NQ, bool DR>
__global__ __launch_bounds__(kWave) void reset_kernel(const StepArgs<T> A) {
  __shared__ T tile[kWave * OS2R_MAX_OBS];
  const int lane = threadIdx.x;
  const long long e0 = (long long)blockIdx.x * kWave;
  const bool valid = e0 + lane < A.N;
  const long long e = valid ? e0 + lane : A.N - 1;
  const TaskPtr<T> ts = as_const(A.task);
  T q[NQ], qd[NQ];
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    q[i] = A.q[i * A.N + e];
    qd[i] = A.qd[i * A.N + e];
  }
  using MD = RtModel<T, NQ>;
  ParamVals<T, NQ> pv;
  const bool doit = A.reset_mask ? A.reset_mask[e] != 0 : true;
  uint32_t epi = A.episode[e];
  uint8_t pose = A.pose[e];
  if (doit) {
    reset_env<T, MD, DR>(A, e, epi, q, qd, pv, pose);
    epi += 1;
  }
  const T h1x = A.hist[2 * A.N + e], h1y = A.hist[3 * A.N + e];
  T obs[OS2R_MAX_OBS];
  bool dn;
  unsigned why;
  observe<T, NQ>(ts, q, qd, h1x, h1y, obs, dn, why);
  if (A.obs) store_obs_tile<T>(A.obs, obs, ts->obs_dim, e0, A.N, lane, tile);
  if (valid && doit) {
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      A.q[i * A.N + e] = q[i];
      A.qd[i * A.N + e] = qd[i];
    }
    A.episode[e] = epi;
    A.pose[e] = pose;
    A.steps[e] = 0;
    store_params<T, NQ>(A, e, pv);
    A.solver_flags[e] = 0u;   // a new episode: the contact solver remembers nothing
#pragma unroll
    for (int k = 0; k < 4 * NQ; ++k) A.solver_l[k * A.N + e] = T(0);
  }
}

// per-env gravity drawn once at create (randomizers/monopod.py:56-61)
template <typename T>
__global__ void gravity_kernel(T* __restrict__ gravity, long long N, long long env_offset, unsigned long long seed,
                               double mean, double std_) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N) return;
  double z0, z1;
  normal2(seed, (uint32_t)(env_offset + e), kStreamGravity, 0, 0, z0, z1);
  gravity[e] = (T)(mean + std_ * z0);
}

template <typename T>
__global__ void fill_kernel(T* __restrict__ dst, long long n, T value) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = value;
}

// launch tables (defined in the per-dtype instantiation units)
template <typename T>
struct Launcher {
  // model_id: index of the matching constexpr table (os2r_models_gen.hpp) or -1 for the run-time model
  static int step(int nq, int model_id, bool contact, bool dr, const StepArgs<T>& args, hipStream_t stream);
  static int reset(int nq, bool dr, const StepArgs<T>& args, hipStream_t stream);
  static void gravity(T* g, long long N, long long off, unsigned long long seed, double mean, double std_, hipStream_t s);
  static void fill(T* dst, long long n, T value, hipStream_t s);
};

}  // namespace os2r
