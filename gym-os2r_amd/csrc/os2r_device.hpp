// os2r_device.hpp — device code of the batched monopod stepper (gfx950 / CDNA4).
//
// One lane integrates one environment.  State lives in HBM as struct-of-arrays
// ([dof][env], coalesced), is held in registers across the `substeps` physics
// iterations of one env-step, and is written back together with the
// observation / reward / done of the same launch.  Robot constants are
// wave-uniform: they are read through scalar loads from a read-only device
// struct and occupy SGPRs, not per-lane registers.  No MFMA: a 5-dof serial
// chain is not a dense contraction; the binding resource is the fp64 (or fp32)
// vector ALU.
//
// What one physics iteration computes (replaces `gazebo.run()`,
// gym_os2r/runtimes/gazebo_runtime.py:76):
//   1. joint transforms, body velocities
//   2. articulated-body algorithm (Featherstone) with joint damping taken
//      implicitly in the projected articulated inertia  D_i = S'I^A S + dt*d_i
//   3. the lower-triangular factor Lc of the inverse of the damping-augmented mass matrix
//      (M~^-1 = Lc Lc^T), read off the same factorisation with one unit-torque inward sweep
//   4. v* = qd + dt*qdd
//   5. ground contact: per body, the centroid of its collision candidate points that are
//      within `margin` of the ground, weighted by (margin - z), gives one frictional point
//      contact with a gap-based non-penetration target (v_n >= -gap/dt)
//   6. projected Gauss-Seidel on the velocities over [contact rows, joint
//      Coulomb friction rows], fixed sweep counts, cold start: a normal-only phase
//      fixes the friction box bounds, then all rows (a convex boxed QP)
//   7. q += dt*v  (semi-implicit Euler)
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/os2r.h"
#include "os2r_models_gen.hpp"

namespace os2r {

constexpr int kWave = 64;
// the solver settings of the default configuration: kernels built for them have compile-time loop bounds (+2 %).
// fp64: the exact finish with its lower sweep cap; fp32: sweeps only (kExact* below)
// (kExact: the exact finish is on, with any positive cap on the solves -- the cap stays a run-time value)
template <typename T> struct StdSolver { static constexpr int kNormalIters = 2; static constexpr bool kExact = true; };
template <> struct StdSolver<float> { static constexpr int kNormalIters = 2; static constexpr bool kExact = false; };
// The solver state of an environment (round 4): phase 2 of every physics iteration starts from the impulses that ended
// the environment's previous iteration -- of this env-step or of the one before: the state lives in HBM between the
// launches (StepArgs::solver_l / solver_flags, os2r_get/set_solver_state), as the reference's backend keeps one persistent
// constraint solver per world behind gym_os2r/runtimes/gazebo_runtime.py:76,111-114 -- clamped into this iteration's box,
// the velocity following row by row; a row without a remembered impulse (a body that had no contact then; every row
// after a reset) keeps what phase 1 gave it.  The solution moves by a thousandth per iteration, so warm_first(nq) sweeps
// identify the active set before the first check (same-box A/B of 2 / 3 / 4 sweeps in round 3).
__host__ __device__ constexpr int warm_first(int) { return 3; }
// The default cap on the sweeps of phase 2 (first sweeps and re-test sweeps together): sweep_cap_base(nq) + kExactRounds
// (the numbers of round 3, whose cold first iteration ran sweep_cap_base(nq) sweeps before its first check)
__host__ __device__ constexpr int sweep_cap_base(int nq) { return nq >= 5 ? 6 : 4; }
constexpr int kExactRounds = 8;
// flags: bit b -- body b had a contact in the environment's last physics iteration (its three impulses are remembered);
// kCarryJoints -- an iteration has run since the reset (the joint-friction impulses are remembered)
constexpr unsigned kCarryJoints = 0x80000000u;
template <typename T, int NQ_> struct SolverCarry {
  unsigned act = 0u;
  T ln[NQ_] = {}, lx[NQ_] = {}, ly[NQ_] = {}, lf[NQ_] = {};
};
template <typename T> __host__ __device__ constexpr int std_iters(int nq) { return sizeof(T) == 8 ? sweep_cap_base(nq) + kExactRounds : 20; }
template <typename T> __host__ __device__ inline bool is_std_solver(int iters, int normal_iters, int exact, int nq) {
  return iters == std_iters<T>(nq) && normal_iters == StdSolver<T>::kNormalIters && (exact > 0) == StdSolver<T>::kExact;
}
// Phase-2 sweeps run in groups of kPgsGroup; the last sweep of a group measures the energy it moved
// (sum over the rows of |residual * impulse change|, the decrease of the QP objective up to a factor <= 2) and an
// environment whose measure is within pgs_tol stops sweeping.  Per lane: what an
// environment computes does not depend on the company it keeps in its wave.
constexpr int kPgsGroup = 4;
// Exact finish of the fixed-box problem (Os2rConfig.pgs_exact > 0, fp64 only): after warm_first(nq)
// sweeps an environment that has not converged solves its free rows exactly -- (S + eps I) d = -G_F^T w_F with
// S = G_F^T G_F (NQ x NQ whatever the number of free rows), eps = kExactEps * trace S, kExactProx proximal iterations,
// impulses from the residuals -- cuts the step at the first bound it meets, and re-tests every row with one measured sweep.
// (two proximal iterations since the rows are equilibrated -- every free row at unit length, so the regularisation is 1e-6 of each
// row's own scale: a third changes neither the solves an environment needs nor the closed loop; three up to round 5)
constexpr int kExactProx = 2;
constexpr double kExactEps = 1e-6, kExactSnap = 1e-12;
// An inconsistent free set (more sticking rows than the dof they act on) leaves a residual on its rows and multipliers
// that move by the same amount round after round: from an environment's second solve of an iteration on, a solve that
// is not cut and leaves more than kExactIncons of the squared residual it found steps on along its multipliers -- past
// the full step -- to the first bound, and counts as cut.
constexpr double kExactIncons = 1e-4, kExactNoBound = 1e300;

// Work done by one wave in one physics iteration, for the counting kernel variants (wave-uniform values).
struct WorkCounts {
  unsigned scanned = 0;       // bodies whose candidate scan ran
  unsigned row_bodies = 0;    // bodies whose contact rows were set up (some lane of the wave touches)
  unsigned body_sweeps = 0;   // phase-2 sweeps executed x bodies they covered
  unsigned sweeps = 0;        // phase-2 sweeps executed (some lane still live)
  unsigned lane_contacts = 0; // (lane, body) pairs in contact
  unsigned live_lane_sweeps = 0;  // phase-2 sweeps x lanes still live in them
  unsigned full_sincos = 0;   // 1 if some lane evaluated sin/cos in full in this iteration
  unsigned exact_solves = 0;  // exact free-set solves executed (some lane of the wave needed one)
  unsigned lane_exact_solves = 0;  // exact free-set solves x lanes that took part
};

// ----------------------------------------------------------------------------------------
// uniform (per-handle) device data
// ----------------------------------------------------------------------------------------
template <typename T>
struct DevModel {
  int nq;
  int axis[OS2R_MAX_DOF];
  T rfix[OS2R_MAX_DOF][9];
  T rpos[OS2R_MAX_DOF][3];
  T mass[OS2R_MAX_DOF];
  T com[OS2R_MAX_DOF][3];
  T icom[OS2R_MAX_DOF][6];
  T damping[OS2R_MAX_DOF];
  T friction[OS2R_MAX_DOF];
  T mu[OS2R_MAX_DOF];
  int act_dof[2];
  T max_torque[2];
  T gravity_z;
  int cand_begin[OS2R_MAX_DOF + 1];  // candidates of body b: [cand_begin[b], cand_begin[b+1])
  T cand_p[OS2R_MAX_CAND][3];
  T cand_center[OS2R_MAX_DOF][3];    // bounding sphere of body b's candidates
  T cand_radius[OS2R_MAX_DOF];
};

template <typename T>
struct DevTask {
  int obs_dim;
  int obs_kind[OS2R_MAX_OBS];
  int obs_src[OS2R_MAX_OBS];
  T obs_low[OS2R_MAX_OBS];
  T obs_high[OS2R_MAX_OBS];
  T done_lo[OS2R_MAX_OBS];
  T done_hi[OS2R_MAX_OBS];
  int reward_id, normalized;
  int idx_pitch_pos, idx_yaw_vel, idx_hip_pos, idx_knee_pos;
  int max_episode_steps;
  int reset_mode, n_reset_poses;
  int reset_pose_id[OS2R_MAX_RESET_POSES];
  int reset_laying[OS2R_MAX_RESET_POSES];
  double reset_pitch[OS2R_MAX_RESET_POSES];
  double reset_hip[OS2R_MAX_RESET_POSES];
  double reset_knee[OS2R_MAX_RESET_POSES];
  int reset_simple;
  double leg_def[6];
  int dof_yaw, dof_pitch, dof_bc, dof_hip, dof_knee;
  int randomize_params;
  int gravity_rollouts;
  double dr_gravity_mean, dr_gravity_std;
  double dr_mass_lo, dr_mass_hi, dr_friction_lo, dr_friction_hi, dr_damping_lo, dr_damping_hi;
  double dr_mu_base, dr_mu_lo, dr_mu_hi;
  double nominal_damping[OS2R_MAX_DOF];
};

// The uniform structs are never written while a kernel runs.  Reading them through
// constant-address-space pointers tells the compiler exactly that, so every access becomes a
// scalar load (s_load -> SGPR operand) instead of a per-lane vector load + v_readfirstlane.
#define OS2R_CONST __attribute__((address_space(4)))
template <typename T> using ModelPtr = const OS2R_CONST DevModel<T>*;
template <typename T> using TaskPtr = const OS2R_CONST DevTask<T>*;
template <typename T> __device__ __forceinline__ ModelPtr<T> as_const(const DevModel<T>* p) { return (ModelPtr<T>)p; }
template <typename T> __device__ __forceinline__ TaskPtr<T> as_const(const DevTask<T>* p) { return (TaskPtr<T>)p; }

template <typename T>
struct StepArgs {
  const DevModel<T>* __restrict__ model;
  const DevTask<T>* __restrict__ task;
  long long N;
  long long env_offset;
  unsigned long long seed;
  unsigned long long step_count;
  int substeps;
  int rollout_steps;   // env-steps of this launch (os2r_rollout; 1 for os2r_step)
  int pgs_iters;
  int pgs_normal_iters;
  int pgs_exact;
  int auto_reset;
  T dt, erp, max_erv, margin, gravity_z;
  T pgs_tol;   // an environment stops sweeping once a checked sweep moved less energy than this (0: exact fixed points only)
  // state, SoA
  T* __restrict__ q;         // [nq][N]
  T* __restrict__ qd;        // [nq][N]
  T* __restrict__ hist;      // [2][2][N]
  T* __restrict__ mass_scale;  // [nq][N]
  T* __restrict__ damping;     // [nq][N]
  T* __restrict__ friction;    // [nq][N]
  T* __restrict__ mu;          // [nq][N]
  T* __restrict__ gravity;     // [N]
  int32_t* __restrict__ steps;
  uint32_t* __restrict__ episode;
  uint8_t* __restrict__ pose;
  T* __restrict__ solver_l;          // [4*nq][N] impulses that ended the environment's last physics iteration (SolverCarry)
  uint32_t* __restrict__ solver_flags;  // [N] which of them are remembered
  unsigned int* __restrict__ violations;  // count of out-of-range caller actions (nullable)
  // step I/O
  const T* __restrict__ actions;  // [N][2] or null
  T* __restrict__ obs;            // [N][D] or null
  T* __restrict__ reward;         // [N] or null
  uint8_t* __restrict__ done;     // [N] or null
  T* __restrict__ term_obs;       // [N][D] or null
  const uint8_t* __restrict__ reset_mask;  // reset kernel only
  unsigned long long* __restrict__ debug;  // diagnostic stamp builds only (else null)
  unsigned long long* __restrict__ counters;  // work counters of the counting kernel variants (os2r_set_work_counters), else null
  uint16_t* __restrict__ reason;              // [N] which observation slots left the reset space in this step (os2r_set_done_reasons), else null
  uint8_t* __restrict__ done_mask;            // [N] 1 where `done` is non-zero, else 0 (os2r_set_done_mask: the bool a gym-level step returns), else null
  uint32_t* __restrict__ mirror;              // two words of mapped host memory (os2r_get_violation_mirror), written by the first wave of every step launch
  // host side of the launch only: the observation layout of the handle's task, 4 bits per slot (slot 0 lowest),
  // compared with the layouts that exist as compile-time variants of the step kernel
  unsigned long long layout_kinds, layout_srcs;
  int layout_dim;
};

// ----------------------------------------------------------------------------------------
// Model access policies.  The device code reads robot constants only through these:
//   RtModel<T,NQ>  any compiled serial chain, constants fetched with scalar loads at run time
//   StModel<T,ID>  one of the reference's four URDF variants, constants are constexpr tables
//                  (os2r_models_gen.hpp) folded into the instruction stream: no loads, no SGPR
//                  pressure, and the exact zeros / ones of the fixed rotations disappear
// ----------------------------------------------------------------------------------------
template <typename T, int NQ_>
struct RtModel {
  static constexpr int NQ = NQ_;
  static constexpr unsigned CMASK = (1u << NQ_) - 1u;
  static constexpr bool kStatic = false;
  ModelPtr<T> p;
  __device__ __forceinline__ int axis(int i) const { return p->axis[i]; }
  __device__ __forceinline__ T rfix(int i, int k) const { return p->rfix[i][k]; }
  __device__ __forceinline__ T rpos(int i, int k) const { return p->rpos[i][k]; }
  __device__ __forceinline__ T mass(int i) const { return p->mass[i]; }
  __device__ __forceinline__ T com(int i, int k) const { return p->com[i][k]; }
  __device__ __forceinline__ T icom(int i, int k) const { return p->icom[i][k]; }
  __device__ __forceinline__ T damping(int i) const { return p->damping[i]; }
  __device__ __forceinline__ T friction(int i) const { return p->friction[i]; }
  __device__ __forceinline__ T mu(int i) const { return p->mu[i]; }
  __device__ __forceinline__ int act_dof(int k) const { return p->act_dof[k]; }
  __device__ __forceinline__ T max_torque(int k) const { return p->max_torque[k]; }
  __device__ __forceinline__ int cand_begin(int b) const { return p->cand_begin[b]; }
  __device__ __forceinline__ T cand(int k, int j) const { return p->cand_p[k][j]; }
  __device__ __forceinline__ T cand_center(int b, int j) const { return p->cand_center[b][j]; }
  __device__ __forceinline__ T cand_radius(int b) const { return p->cand_radius[b]; }
};

// Per robot, evaluated by the compiler ONCE as a constant object: for every body the coordinate with the fewest runs
// of equal values among its candidates, and for every candidate whether it starts a run of that coordinate.  (As
// plain constexpr functions called with an unrolled loop index these were folded for the 32-point links of the
// monopod but evaluated at RUN time -- dependent scalar loads of the table, a compare per point, every physics
// iteration -- for the 40-point lumped link of monopod-fixed_hip: 58 % of an env-step of Monopod-balance-v1.)
template <int ID>
struct CandMeta {
  using Tb = gen::Tables<ID>;
  struct Data { int axis[OS2R_MAX_DOF]; bool starts[OS2R_MAX_CAND > 0 ? OS2R_MAX_CAND : 1]; };
  static constexpr int runs(int b, int a) {
    int n = 0;
    for (int k = Tb::cand_begin[b]; k < Tb::cand_begin[b + 1]; ++k)
      if (k == Tb::cand_begin[b] || Tb::cand_p[k][a] != Tb::cand_p[k - 1][a]) ++n;
    return n;
  }
  static constexpr Data make() {
    Data d{};
    for (int b = 0; b < Tb::nq; ++b) {
      int best = 0;
      for (int a = 1; a < 3; ++a)
        if (runs(b, a) < runs(b, best)) best = a;
      d.axis[b] = best;
      for (int k = Tb::cand_begin[b]; k < Tb::cand_begin[b + 1]; ++k)
        d.starts[k] = k == Tb::cand_begin[b] || Tb::cand_p[k][best] != Tb::cand_p[k - 1][best];
    }
    return d;
  }
  static constexpr Data value = make();
};

template <typename T, int ID>
struct StModel {
  using Tb = gen::Tables<ID>;
  static constexpr int NQ = Tb::nq;
  static constexpr unsigned CMASK = Tb::cmask;
  static constexpr bool kStatic = true;
  __device__ __forceinline__ constexpr int axis(int i) const { return Tb::axis[i]; }
  __device__ __forceinline__ constexpr T rfix(int i, int k) const { return (T)Tb::rfix[i][k]; }
  __device__ __forceinline__ constexpr T rpos(int i, int k) const { return (T)Tb::rpos[i][k]; }
  __device__ __forceinline__ constexpr T mass(int i) const { return (T)Tb::mass[i]; }
  __device__ __forceinline__ constexpr T com(int i, int k) const { return (T)Tb::com[i][k]; }
  __device__ __forceinline__ constexpr T icom(int i, int k) const { return (T)Tb::icom[i][k]; }
  __device__ __forceinline__ constexpr T damping(int i) const { return (T)Tb::damping[i]; }
  __device__ __forceinline__ constexpr T friction(int i) const { return (T)Tb::friction[i]; }
  __device__ __forceinline__ constexpr T mu(int i) const { return (T)Tb::mu[i]; }
  __device__ __forceinline__ constexpr int act_dof(int k) const { return Tb::act_dof[k]; }
  __device__ __forceinline__ constexpr T max_torque(int k) const { return (T)Tb::max_torque[k]; }
  __device__ __forceinline__ constexpr int cand_begin(int b) const { return Tb::cand_begin[b]; }
  __device__ __forceinline__ constexpr T cand(int k, int j) const { return (T)Tb::cand_p[k][j]; }
  __device__ __forceinline__ constexpr T cand_center(int b, int j) const { return (T)Tb::cand_center[b][j]; }
  __device__ __forceinline__ constexpr T cand_radius(int b) const { return (T)Tb::cand_radius[b]; }
  // Candidate points of a link often share a coordinate (the two faces of a plate, a rim at constant height): the
  // scan evaluates that coordinate's terms once per run of equal values instead of once per point (CandMeta below).
  static constexpr int cand_group_axis(int b) { return CandMeta<ID>::value.axis[b]; }
  static constexpr bool cand_starts_run(int /*b*/, int k, int /*a: the group axis of k's body*/) { return CandMeta<ID>::value.starts[k]; }
};

// ----------------------------------------------------------------------------------------
// counter RNG: Philox4x32-10 (Salmon et al., SC'11).  Streams and counters are part of the
// stepper's specification (DESIGN.md "Random streams").
// ----------------------------------------------------------------------------------------
enum { kStreamAction = 1, kStreamReset = 2, kStreamParams = 3, kStreamGravity = 4 };

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                               uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ double u53(uint32_t a, uint32_t b) {
  return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}

__device__ __forceinline__ void uniform2(unsigned long long seed, uint32_t env, uint32_t stream,
                                          uint32_t ctr, uint32_t blk, double& u0, double& u1) {
  uint32_t o[4];
  philox4x32_10(env, stream, ctr, blk, (uint32_t)(seed & 0xffffffffull), (uint32_t)(seed >> 32), o);
  u0 = u53(o[0], o[1]);
  u1 = u53(o[2], o[3]);
}

__device__ __forceinline__ void normal2(unsigned long long seed, uint32_t env, uint32_t stream,
                                         uint32_t ctr, uint32_t blk, double& z0, double& z1) {
  double u0, u1;
  uniform2(seed, env, stream, ctr, blk, u0, u1);
  const double r = sqrt(-2.0 * log(1.0 - u0));
  const double th = 6.283185307179586476925286766559 * u1;
  z0 = r * cos(th);
  z1 = r * sin(th);
}

// ----------------------------------------------------------------------------------------
// small vector helpers
// ----------------------------------------------------------------------------------------
template <typename T>
struct V3 {
  T x, y, z;
};
template <typename T> __device__ __forceinline__ V3<T> mk(T x, T y, T z) { return V3<T>{x, y, z}; }
template <typename T> __device__ __forceinline__ V3<T> operator+(V3<T> a, V3<T> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <typename T> __device__ __forceinline__ V3<T> operator-(V3<T> a, V3<T> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <typename T> __device__ __forceinline__ V3<T> operator*(T s, V3<T> a) { return {s * a.x, s * a.y, s * a.z}; }
template <typename T> __device__ __forceinline__ T dot(V3<T> a, V3<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename T> __device__ __forceinline__ V3<T> cross(V3<T> a, V3<T> b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template <typename T> __device__ __forceinline__ T comp(V3<T> a, int k) { return k == 0 ? a.x : (k == 1 ? a.y : a.z); }
template <typename T> __device__ __forceinline__ void add_comp(V3<T>& a, int k, T v) {
  if (k == 0) a.x += v; else if (k == 1) a.y += v; else a.z += v;
}
// R (row-major 3x3) * v   and   R^T * v
template <typename T> __device__ __forceinline__ V3<T> rmul(const T (&R)[9], V3<T> v) {
  return {R[0] * v.x + R[1] * v.y + R[2] * v.z, R[3] * v.x + R[4] * v.y + R[5] * v.z, R[6] * v.x + R[7] * v.y + R[8] * v.z};
}
template <typename T> __device__ __forceinline__ V3<T> rtmul(const T (&R)[9], V3<T> v) {
  return {R[0] * v.x + R[3] * v.y + R[6] * v.z, R[1] * v.x + R[4] * v.y + R[7] * v.z, R[2] * v.x + R[5] * v.y + R[8] * v.z};
}

__device__ __forceinline__ double sqrt_t(double x) { return sqrt(x); }
__device__ __forceinline__ float sqrt_t(float x) { return sqrtf(x); }
// sin and cos of a joint angle.  Joint angles stay within a few 1e4 rad even for a spinning
// periodic joint over a 100 000-step episode, so the argument reduction is the two-FMA
// Cody-Waite form (exact product k*C1, FMA rounding relative to the small remainder) with the
// library routine as the wave-uniform fallback beyond 2^19*pi/2; the kernels are the classical
// minimax polynomials on [-pi/4, pi/4] (fdlibm k_sin / k_cos coefficients), ~1 ulp.
__device__ __forceinline__ bool sincos_in_range(double x) { return fabs(x) < 8.2e5; }
__device__ __forceinline__ bool sincos_in_range(float x) { return fabsf(x) < 200.0f; }   // beyond: library routine
__device__ __forceinline__ void sincos_lib(double x, double& s, double& c) { sincos(x, &s, &c); }
__device__ __forceinline__ void sincos_lib(float x, float& s, float& c) { sincosf(x, &s, &c); }
// fp32: three-term Cody-Waite reduction (exact products for |k| < 2^9, i.e. |x| < 200 rad with margin) and the
// minimax kernels of sinf / cosf on [-pi/4, pi/4] (Cephes coefficients), ~1 ulp
__device__ __forceinline__ void sincos_fast(float x, float& s, float& c) {
  const float k = __builtin_rintf(x * 0.636619772f);
  float r = __builtin_fmaf(-k, 1.5703125f, x);                   // pi/2 = 1.5703125 + 4.837512969970703125e-4 + 7.54978995489188e-8
  r = __builtin_fmaf(-k, 4.837512969970703125e-4f, r);
  r = __builtin_fmaf(-k, 7.54978995489188e-8f, r);
  const float z = r * r;
  float ps = -1.9515295891e-4f;
  ps = __builtin_fmaf(ps, z, 8.3321608736e-3f);
  ps = __builtin_fmaf(ps, z, -1.6666654611e-1f);
  const float sr = __builtin_fmaf(r * z, ps, r);
  float pc = 2.443315711809948e-5f;
  pc = __builtin_fmaf(pc, z, -1.388731625493765e-3f);
  pc = __builtin_fmaf(pc, z, 4.166664568298827e-2f);
  const float cr = __builtin_fmaf(z * z, pc, __builtin_fmaf(-0.5f, z, 1.0f));
  const int n = (int)k;
  const float a = (n & 1) ? cr : sr, b = (n & 1) ? sr : cr;
  s = (n & 2) ? -a : a;
  c = ((n + 1) & 2) ? -b : b;
}
__device__ __forceinline__ void sincos_fast(double x, double& s, double& c) {
  const double k = __builtin_rint(x * 0x1.45f306dc9c883p-1);          // x * 2/pi
  double r = __builtin_fma(-k, 0x1.921fb54442d18p+0, x);               // pi/2 = C1 + C2 + ...
  r = __builtin_fma(-k, 0x1.1a62633145c07p-54, r);
  const double z = r * r;
  double ps = 0x1.5d93a5acfd57cp-33;                                    // S6
  ps = __builtin_fma(ps, z, -0x1.ae5e68a2b9cebp-26);
  ps = __builtin_fma(ps, z, 0x1.71de357b1fe7dp-19);
  ps = __builtin_fma(ps, z, -0x1.a01a019c161d5p-13);
  ps = __builtin_fma(ps, z, 0x1.111111110f8a6p-7);
  ps = __builtin_fma(ps, z, -0x1.5555555555549p-3);
  const double sr = __builtin_fma(r * z, ps, r);
  double pc = -0x1.8fae9be8838d4p-37;                                   // C6
  pc = __builtin_fma(pc, z, 0x1.1ee9ebdb4b1c4p-29);
  pc = __builtin_fma(pc, z, -0x1.27e4f809c52adp-22);
  pc = __builtin_fma(pc, z, 0x1.a01a019cb1590p-16);
  pc = __builtin_fma(pc, z, -0x1.6c16c16c15177p-10);
  pc = __builtin_fma(pc, z, 0x1.555555555554cp-5);
  const double cr = __builtin_fma(z * z, pc, __builtin_fma(-0.5, z, 1.0));
  const int n = (int)k;
  const double a = (n & 1) ? cr : sr, b = (n & 1) ? sr : cr;
  s = (n & 2) ? -a : a;
  c = ((n + 1) & 2) ? -b : b;
}

// Reciprocal / reciprocal square root by the hardware estimate plus Newton steps (~1 ulp), an
// order of magnitude fewer instructions than the correctly rounded divide / sqrt.  Used in the
// physics iteration only; the epilogue keeps IEEE division (bit parity with numpy).
__device__ __forceinline__ double rcp_t(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ float rcp_t(float x) {
  float r = __builtin_amdgcn_rcpf(x);              // 1 ulp estimate; one Newton step
  return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
}
__device__ __forceinline__ double rsqrt_t(double x) {
  double r = __builtin_amdgcn_rsq(x);
  // r <- r * (1.5 - 0.5 x r^2), twice
  r = r * __builtin_fma(-0.5 * x * r, r, 1.5);
  r = r * __builtin_fma(-0.5 * x * r, r, 1.5);
  return r;
}
__device__ __forceinline__ float rsqrt_t(float x) {
  float r = __builtin_amdgcn_rsqf(x);
  return r * __builtin_fmaf(-0.5f * x * r, r, 1.5f);
}

__device__ __forceinline__ double fmax_t(double a, double b) { return fmax(a, b); }
__device__ __forceinline__ float fmax_t(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ double fmin_t(double a, double b) { return fmin(a, b); }
__device__ __forceinline__ float fmin_t(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fabs_t(double x) { return fabs(x); }
__device__ __forceinline__ float fabs_t(float x) { return fabsf(x); }

// Finite test on the bit pattern of an *opaque* copy.  Under the no-NaN / no-Inf flags the optimiser
// treats a NaN result as poison and may fold any test of it, bit tests included; the empty asm hides
// where the value came from, so the exponent field is really inspected.
__device__ __forceinline__ bool finite_t(double x) {
  asm volatile("" : "+v"(x));
  return ((__double_as_longlong(x) >> 52) & 0x7ff) != 0x7ff;
}
__device__ __forceinline__ bool finite_t(float x) {
  asm volatile("" : "+v"(x));
  return ((__float_as_int(x) >> 23) & 0xff) != 0xff;
}

// symmetric 3x3 stored as xx xy xz yy yz zz
template <typename T> __device__ __forceinline__ V3<T> symmul(const T (&S)[6], V3<T> v) {
  return {S[0] * v.x + S[1] * v.y + S[2] * v.z, S[1] * v.x + S[3] * v.y + S[4] * v.z, S[2] * v.x + S[4] * v.y + S[5] * v.z};
}

// Articulated inertia  [[A, H], [H^T, M]]  acting on [omega; v]:  n = A w + H v,  f = H^T w + M v
template <typename T>
struct ArtInertia {
  T A[6];  // symmetric
  T H[9];  // general, row-major
  T M[6];  // symmetric
};

// B' = R B R^T for a general 3x3
template <typename T> __device__ __forceinline__ void rot_general(const T (&R)[9], const T (&B)[9], T (&out)[9]) {
  T t[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) t[3 * i + j] = R[3 * i] * B[j] + R[3 * i + 1] * B[3 + j] + R[3 * i + 2] * B[6 + j];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) out[3 * i + j] = t[3 * i] * R[3 * j] + t[3 * i + 1] * R[3 * j + 1] + t[3 * i + 2] * R[3 * j + 2];
}
// S' = R S R^T for a symmetric 3x3 (6 unique outputs)
template <typename T> __device__ __forceinline__ void rot_sym(const T (&R)[9], const T (&S)[6], T (&out)[6]) {
  const T B[9] = {S[0], S[1], S[2], S[1], S[3], S[4], S[2], S[4], S[5]};
  T t[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) t[3 * i + j] = R[3 * i] * B[j] + R[3 * i + 1] * B[3 + j] + R[3 * i + 2] * B[6 + j];
  int k = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = i; j < 3; ++j) out[k++] = t[3 * i] * R[3 * j] + t[3 * i + 1] * R[3 * j + 1] + t[3 * i + 2] * R[3 * j + 2];
}

// ----------------------------------------------------------------------------------------
// per-lane physical parameters: nominal (uniform, SGPR) or randomised (per lane)
// ----------------------------------------------------------------------------------------
template <typename T, typename MD, bool DR>
struct Params;
template <typename T, typename MD>
struct Params<T, MD, false> {
  MD m;
  T g;  // uniform gravity (a config value, not a model constant)
  __device__ __forceinline__ T mass(int i) const { return m.mass(i); }
  __device__ __forceinline__ T damping(int i) const { return m.damping(i); }
  __device__ __forceinline__ T friction(int i) const { return m.friction(i); }
  __device__ __forceinline__ T mu(int i) const { return m.mu(i); }
  __device__ __forceinline__ T gravity() const { return g; }
};
// Randomised parameters wait in per-lane LDS slots (round 4; lds[(field * NQ + i) * 64 + lane], written once per env-step
// by the step kernel: body masses already multiplied out): a physics iteration reads them where it uses them with one
// ds_read each -- no address arithmetic, no base pointers re-fetched from the argument segment, 21 L2 round trips fewer
// per iteration than reading the HBM arrays at the point of use (rounds 1-3), and still no registers held across the kernel.
template <typename T, typename MD>
struct Params<T, MD, true> {
  MD m;
  const T* slots;   // this lane's column of the parameter slots
  T g;
  static constexpr int NQ = MD::NQ;
  __device__ __forceinline__ T mass(int i) const { return slots[(0 * NQ + i) * kWave]; }
  __device__ __forceinline__ T damping(int i) const { return slots[(1 * NQ + i) * kWave]; }
  __device__ __forceinline__ T friction(int i) const { return slots[(2 * NQ + i) * kWave]; }
  __device__ __forceinline__ T mu(int i) const { return slots[(3 * NQ + i) * kWave]; }
  __device__ __forceinline__ T gravity() const { return g; }
};

// Diagnostic phase stamps (separate build with -DOS2R_STAMPS, never in the shipped library):
// shader-clock ticks per phase are summed per wave and written by lane 0 to a debug buffer that
// nothing else reads (cdna_hip_programming.md, "In-kernel stamps").
constexpr int kStamps = 32;   // 0..11 phases, 12..23 finer marks inside the dynamics, 24/25 shader-clock and 100 MHz real-time ticks of the wave, 26..30 the passes of the exact solve
#if defined(OS2R_STAMPS) && defined(OS2R_STAMPS_LIGHT)
// the light stamp build (make stamps_light): no stamp inside the env-step -- the code of the shipped kernel -- only the wave's
// start and end on the 100 MHz clock, its life in shader cycles and its place (tools/dbg/wave_times.py: who ends a launch, and when)
#define OS2R_STAMP(idx) do { } while (0)
#elif defined(OS2R_STAMPS)
#define OS2R_STAMP(idx)                                                                        \
  do {                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    unsigned long long t_;                                                                     \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                 \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    stamps[idx] += t_ - stamp_prev;                                                            \
    stamp_prev = t_;                                                                           \
  } while (0)
#elif defined(OS2R_PHASE_MARKS)
// static markers at the stamp points (diagnostic builds only: tools/phase_insts.py counts the instructions between them
// and sets them against the stamp build's ticks): s_nop 15, then the stamp index in two s_nop immediates
#define OS2R_STAMP(idx) asm volatile("s_nop 15\n\ts_nop %0\n\ts_nop %1" :: "i"((idx) % 8), "i"((idx) / 8) : "memory")
#else
#define OS2R_STAMP(idx) do { } while (0)
#endif

// Static markers in the ISA (diagnostic builds with -DOS2R_ISA_MARKS only: tools/isa_histogram.py --marks): s_nop with a
// distinctive count at the boundaries of the exact solve's passes, so that their instruction counts can be read off
#ifdef OS2R_ISA_MARKS
#define OS2R_ISA_MARK(n) asm volatile("s_nop " #n ::: "memory")
#else
#define OS2R_ISA_MARK(n) do { } while (0)
#endif

// sched_barrier mask: everything may cross except vector-memory instructions
constexpr int kPinVmem = 0x1 | 0x2 | 0x4 | 0x8 | 0x80 | 0x100 | 0x200 | 0x400;

// sched_barrier mask: everything may cross except LDS instructions
constexpr int kPinDs = 0x1 | 0x2 | 0x4 | 0x8 | 0x10 | 0x20 | 0x40 | 0x400;

// Opaque copy: a fresh SSA value the optimiser cannot merge with earlier uses.  Used to
// *re*-compute cheap quantities (a joint rotation is 12 FMAs) instead of holding them in
// registers across the whole physics iteration: fp64 state is register hungry and anything
// beyond 512 registers per lane spills to scratch, i.e. to HBM.
__device__ __forceinline__ double opaque(double x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ float opaque(float x) { asm volatile("" : "+v"(x)); return x; }

// One of three values by a wave-uniform index.  With a compile-time index it folds to the value; with a run-time one
// (the generic kernels: the joint axis is data) it stays two selects -- indexing a local array with it would put
// the array in scratch memory (88-176 B per lane in the run-time-model kernels up to round 2).
template <typename T> __device__ __forceinline__ T pick3(int k, T a, T b, T c) { return k == 0 ? a : (k == 1 ? b : c); }

// R_i = Rfix_i * Rot(axis_i, q_i): child orientation in its parent, from (sin q_i, cos q_i)
template <typename T, typename MD>
__device__ __forceinline__ void joint_rotation(const MD& md, int i, T s_, T c_, T (&R)[9]) {
  const T s = opaque(s_), c = opaque(c_);
  const int ax = md.axis(i);
  // columns ca, cb of Rfix rotate into each other; column ax stays
  const int ca = ax == 0 ? 1 : (ax == 1 ? 2 : 0);
  const int cb = ax == 0 ? 2 : (ax == 1 ? 0 : 1);
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const T fa = md.rfix(i, 3 * r + ca), fb = md.rfix(i, 3 * r + cb);
    const T keep = md.rfix(i, 3 * r + ax), va = c * fa + s * fb, vb = c * fb - s * fa;
#pragma unroll
    for (int j = 0; j < 3; ++j) R[3 * r + j] = j == ax ? keep : (j == ca ? va : vb);
  }
}

// next body above b that carries contact candidates (NB if none)
template <unsigned CMASK, int NB>
__device__ __forceinline__ constexpr int next_cand_body(int b) {
  for (int k = b + 1; k < NB; ++k)
    if ((CMASK >> k) & 1u) return k;
  return NB;
}
// calls f(integral_constant<B>) for the candidate-carrying body B == b; false if there is none
template <int B, int NB, unsigned CMASK, typename F>
__device__ __forceinline__ bool for_body(int b, F&& f) {
  if constexpr (B < NB) {
    // at most three bodies per specialised form: their rows fit the register file, a fourth spills
    if constexpr (((CMASK >> B) & 1u) && __builtin_popcount(CMASK >> B) <= 3) {
      if (b == B) { f(std::integral_constant<int, B>{}); return true; }
    }
    return for_body<B + 1, NB, CMASK>(b, f);
  } else {
    return false;
  }
}

// SOLVER: which phase-2 solver(s) the kernel carries -- kSolverBoth: chosen at run time by pgs_exact (run-time code objects
// built without knowing the handle's settings); kSolverExact: the exact finish (pgs_exact > 0, fixed box), the grouped
// sweeps of the round-1/2 solver are not instantiated; kSolverSweeps: sweeps only (every fp32 kernel; fp64 with pgs_exact = 0).
// One solver per kernel keeps the 5-dof fp64 kernels inside the register file.
enum { kSolverBoth = 0, kSolverExact = 1, kSolverSweeps = 2 };
template <typename T, typename MD, bool CONTACT, bool DR, bool COUNT = false, int SOLVER = kSolverBoth>
__device__ __forceinline__ void substep(const MD& md, const Params<T, MD, DR>& par,
                                        T (&q)[MD::NQ], T (&qd)[MD::NQ], T (&sn)[MD::NQ], T (&cs)[MD::NQ], bool first_iteration,
                                        T tau_hip, T tau_knee, T dt, T erp,
                                        T max_erv, T margin, int pgs_iters, int pgs_normal_iters, int pgs_exact, T pgs_tol, T* __restrict__ lds,
                                        const T* __restrict__ cand_lds, ModelPtr<T> mconst, WorkCounts& wc, SolverCarry<T, MD::NQ>& carry
#ifdef OS2R_STAMPS
                                        , unsigned long long (&stamps)[kStamps], unsigned long long& stamp_prev
#endif
                                        ) {
  constexpr int NQ = MD::NQ;
  constexpr unsigned CMASK = CONTACT ? MD::CMASK : 0u;
  const T inv_dt = rcp_t(dt);
  // the first parameter pair of the inward pass is requested before anything else
  const T m_first = par.mass(NQ - 1), damp_first = par.damping(NQ - 1);
  __builtin_amdgcn_sched_barrier(DR ? kPinDs : kPinVmem);
  // ---- 1. sin/cos of the joint angles; rotations are rebuilt from them where needed ----
  // The first iteration of an env-step evaluates them; the later ones turn (sin, cos) by the angle the
  // joint moved in the previous iteration, d = dt * qd (exactly the increment the integrator applied), with
  // sin d and cos d - 1 from their series (|d| < 0.06, i.e. |qd| < 600 rad/s at dt = 1e-4, beyond the velocity at
  // which an episode ends: truncation below 2e-20): 19 instructions per joint instead of ~45, rounding ~1 ulp per
  // turn, refreshed every env-step.  (Up to round 2 the bound was 0.01 with two terms fewer: without ground contact
  // the free leg spins at 100-370 rad/s and most lanes fell back to the full evaluation -- with both paths executed
  // by a wave whose lanes disagree, 13 % of an env-step of C2.)
  {
    bool ok = true, small = !first_iteration;
#pragma unroll
    for (int i = 0; i < NQ; ++i) { ok = ok && sincos_in_range(q[i]); small = small && fabs_t(dt * qd[i]) < T(0.06); }
    // The choice is made per lane (divergent branches; a wave whose lanes agree, the usual case, executes one
    // side only): what a lane computes must not depend on who shares its wave.
    if constexpr (COUNT) wc.full_sincos += __ballot(!small) != 0ull ? 1u : 0u;
    if (small) {
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        const T d = dt * qd[i], z = d * d;
        T p = fma_t(z, T(-1.0 / 39916800.0), T(1.0 / 362880.0));
        p = fma_t(z, p, T(-1.0 / 5040.0));
        p = fma_t(z, p, T(1.0 / 120.0));
        p = fma_t(z, p, T(-1.0 / 6.0));
        const T sd = fma_t(d * z, p, d);
        T r = fma_t(z, T(-1.0 / 3628800.0), T(1.0 / 40320.0));
        r = fma_t(z, r, T(-1.0 / 720.0));
        r = fma_t(z, r, T(1.0 / 24.0));
        r = fma_t(z, r, T(-0.5));
        const T cm = z * r;
        const T s0 = sn[i], c0 = cs[i];
        sn[i] = s0 + fma_t(c0, sd, s0 * cm);
        cs[i] = c0 + fma_t(-s0, sd, c0 * cm);
      }
    } else if (ok) {
      // one range check for all joints of the lane, so the five evaluations are straight-line code
#pragma unroll
      for (int i = 0; i < NQ; ++i) sincos_fast(q[i], sn[i], cs[i]);
    } else {
#pragma unroll
      for (int i = 0; i < NQ; ++i) sincos_lib(q[i], sn[i], cs[i]);
    }
  }

  OS2R_STAMP(0);
  // Per-lane LDS slots (slot-major: lds[slot * 64 + lane], conflict free).  During the
  // articulated-body passes they hold the per-joint quantities that must survive from one pass
  // to the next (U = I^A S, 1/D, u); afterwards the same storage holds the triangular factor of
  // the inverse mass matrix (mirror for the contact-row set-up).  Keeping these out of the
  // register file is what keeps the fp64 kernel free of scratch spills.
  const int lane_ = threadIdx.x;
  auto L = [&](int slot) -> T& { return lds[slot * kWave + lane_]; };
  constexpr int kUa = 0, kUl = 3 * NQ, kDi = 6 * NQ, kU = 7 * NQ;  // 8*NQ slots
  auto ldv = [&](int base, int i) { return mk(L(base + 3 * i), L(base + 3 * i + 1), L(base + 3 * i + 2)); };
  auto stv = [&](int base, int i, V3<T> x) { L(base + 3 * i) = x.x; L(base + 3 * i + 1) = x.y; L(base + 3 * i + 2) = x.z; };

  // ---- 2a. body velocities (body coordinates), outward ----
  V3<T> wv[NQ], vv[NQ];
  {
    V3<T> w = mk<T>(0, 0, 0), v = mk<T>(0, 0, 0);
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      if (i > 0) {
        const V3<T> r = mk(md.rpos(i, 0), md.rpos(i, 1), md.rpos(i, 2));
        T Ri[9];
        joint_rotation<T>(md, i, sn[i], cs[i], Ri);
        const V3<T> wn = rtmul(Ri, w);
        v = rtmul(Ri, v + cross(w, r));
        w = wn;
      }
      add_comp(w, md.axis(i), qd[i]);
      wv[i] = w;
      vv[i] = v;
    }
  }

  OS2R_STAMP(12);
  // ---- 2b. articulated inertias and bias forces, inward ----
  {
    ArtInertia<T> acc;     // children's contribution, in the current body's frame
    V3<T> pn, pf;          // children's bias force [moment; force]
    T m_next = m_first, damp_next = damp_first;
#pragma unroll
    for (int i = NQ - 1; i >= 0; --i) {
      const int ax = md.axis(i);
      const V3<T> w = wv[i], v = vv[i];
      // rigid-body inertia of body i about its frame origin
      const T m = m_next;
      const T damp = damp_next;
      if (i > 0) {
        // randomised parameters come from their LDS slots: request the next body's pair a whole body ahead
        // (the barrier keeps the requests on this side; at one wave per SIMD nothing else hides them)
        m_next = par.mass(i - 1);
        damp_next = par.damping(i - 1);
        __builtin_amdgcn_sched_barrier(DR ? kPinDs : kPinVmem);
      }
      const V3<T> cm = mk(md.com(i, 0), md.com(i, 1), md.com(i, 2));
      const V3<T> h = m * cm;
      ArtInertia<T> I;
      I.A[0] = md.icom(i, 0) + m * (cm.y * cm.y + cm.z * cm.z);
      I.A[1] = md.icom(i, 1) - m * cm.x * cm.y;
      I.A[2] = md.icom(i, 2) - m * cm.x * cm.z;
      I.A[3] = md.icom(i, 3) + m * (cm.x * cm.x + cm.z * cm.z);
      I.A[4] = md.icom(i, 4) - m * cm.y * cm.z;
      I.A[5] = md.icom(i, 5) + m * (cm.x * cm.x + cm.y * cm.y);
      I.H[0] = 0; I.H[1] = -h.z; I.H[2] = h.y;
      I.H[3] = h.z; I.H[4] = 0; I.H[5] = -h.x;
      I.H[6] = -h.y; I.H[7] = h.x; I.H[8] = 0;
      I.M[0] = m; I.M[1] = 0; I.M[2] = 0; I.M[3] = m; I.M[4] = 0; I.M[5] = m;
      // rigid bias force v x* (I v)
      const V3<T> nI = symmul(I.A, w) + cross(h, v);
      const V3<T> fI = m * v - cross(h, w);
      V3<T> pAn = cross(w, nI) + cross(v, fI);
      V3<T> pAf = cross(w, fI);
      if (i < NQ - 1) {
#pragma unroll
        for (int k = 0; k < 6; ++k) { I.A[k] += acc.A[k]; I.M[k] += acc.M[k]; }
#pragma unroll
        for (int k = 0; k < 9; ++k) I.H[k] += acc.H[k];
        pAn = pAn + pn;
        pAf = pAf + pf;
      }
      // U = I^A S : column `ax` of [[A],[H^T]]
      const V3<T> Ua = mk(pick3(ax, I.A[0], I.A[1], I.A[2]), pick3(ax, I.A[1], I.A[3], I.A[4]), pick3(ax, I.A[2], I.A[4], I.A[5]));
      const V3<T> Ul = mk(pick3(ax, I.H[0], I.H[3], I.H[6]), pick3(ax, I.H[1], I.H[4], I.H[7]), pick3(ax, I.H[2], I.H[5], I.H[8]));
      const T D = comp(Ua, ax) + dt * damp;
      const T Dinv = rcp_t(D);
      T tau = T(0);
      if (i == md.act_dof(0)) tau = tau_hip;
      if (i == md.act_dof(1)) tau = tau_knee;
      const T u = tau - damp * qd[i] - comp(pAn, ax);
      stv(kUa, i, Ua);
      stv(kUl, i, Ul);
      L(kDi + i) = Dinv;
      L(kU + i) = u;
      if (i > 0) {
        // Ia = I^A - U U^T / D
        const V3<T> ka = Dinv * Ua, kl = Dinv * Ul;
        ArtInertia<T> Ia;
        Ia.A[0] = I.A[0] - ka.x * Ua.x; Ia.A[1] = I.A[1] - ka.x * Ua.y; Ia.A[2] = I.A[2] - ka.x * Ua.z;
        Ia.A[3] = I.A[3] - ka.y * Ua.y; Ia.A[4] = I.A[4] - ka.y * Ua.z; Ia.A[5] = I.A[5] - ka.z * Ua.z;
        Ia.H[0] = I.H[0] - ka.x * Ul.x; Ia.H[1] = I.H[1] - ka.x * Ul.y; Ia.H[2] = I.H[2] - ka.x * Ul.z;
        Ia.H[3] = I.H[3] - ka.y * Ul.x; Ia.H[4] = I.H[4] - ka.y * Ul.y; Ia.H[5] = I.H[5] - ka.y * Ul.z;
        Ia.H[6] = I.H[6] - ka.z * Ul.x; Ia.H[7] = I.H[7] - ka.z * Ul.y; Ia.H[8] = I.H[8] - ka.z * Ul.z;
        Ia.M[0] = I.M[0] - kl.x * Ul.x; Ia.M[1] = I.M[1] - kl.x * Ul.y; Ia.M[2] = I.M[2] - kl.x * Ul.z;
        Ia.M[3] = I.M[3] - kl.y * Ul.y; Ia.M[4] = I.M[4] - kl.y * Ul.z; Ia.M[5] = I.M[5] - kl.z * Ul.z;
        // velocity-product acceleration c = v x (S qd)
        V3<T> sq = mk<T>(0, 0, 0);
        add_comp(sq, ax, qd[i]);
        const V3<T> ca = cross(w, sq), cl = cross(v, sq);
        // pa = pA + Ia c + U u / D
        const V3<T> Hc = mk(Ia.H[0] * cl.x + Ia.H[1] * cl.y + Ia.H[2] * cl.z, Ia.H[3] * cl.x + Ia.H[4] * cl.y + Ia.H[5] * cl.z,
                            Ia.H[6] * cl.x + Ia.H[7] * cl.y + Ia.H[8] * cl.z);
        const V3<T> Htc = mk(Ia.H[0] * ca.x + Ia.H[3] * ca.y + Ia.H[6] * ca.z, Ia.H[1] * ca.x + Ia.H[4] * ca.y + Ia.H[7] * ca.z,
                             Ia.H[2] * ca.x + Ia.H[5] * ca.y + Ia.H[8] * ca.z);
        const T ud = u * Dinv;
        const V3<T> pan = pAn + symmul(Ia.A, ca) + Hc + ud * Ua;
        const V3<T> paf = pAf + Htc + symmul(Ia.M, cl) + ud * Ul;
        // express in the parent frame: rotate by R_i, then shift by r_i
        const V3<T> r = mk(md.rpos(i, 0), md.rpos(i, 1), md.rpos(i, 2));
        T Ri[9];
        joint_rotation<T>(md, i, sn[i], cs[i], Ri);
        T Ar[6], Hr[9], Mr[6];
        rot_sym(Ri, Ia.A, Ar);
        rot_general(Ri, Ia.H, Hr);
        rot_sym(Ri, Ia.M, Mr);
        const T Mf[9] = {Mr[0], Mr[1], Mr[2], Mr[1], Mr[3], Mr[4], Mr[2], Mr[4], Mr[5]};
        // H'' = Hr + r^ M  (column j: r x M[:,j])
        T Hs[9];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const V3<T> col = cross(r, mk(Mf[j], Mf[3 + j], Mf[6 + j]));
          Hs[j] = Hr[j] + col.x; Hs[3 + j] = Hr[3 + j] + col.y; Hs[6 + j] = Hr[6 + j] + col.z;
        }
        // A''[i][j] = Ar[i][j] + (r x Hr_row_j)[i] + (r x Hs_row_i)[j]
        V3<T> x0[3], x1[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          x0[k] = cross(r, mk(Hr[3 * k], Hr[3 * k + 1], Hr[3 * k + 2]));
          x1[k] = cross(r, mk(Hs[3 * k], Hs[3 * k + 1], Hs[3 * k + 2]));
        }
        acc.A[0] = Ar[0] + x0[0].x + x1[0].x;
        acc.A[1] = Ar[1] + x0[1].x + x1[0].y;
        acc.A[2] = Ar[2] + x0[2].x + x1[0].z;
        acc.A[3] = Ar[3] + x0[1].y + x1[1].y;
        acc.A[4] = Ar[4] + x0[2].y + x1[1].z;
        acc.A[5] = Ar[5] + x0[2].z + x1[2].z;
#pragma unroll
        for (int k = 0; k < 9; ++k) acc.H[k] = Hs[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) acc.M[k] = Mr[k];
        pf = rmul(Ri, paf);
        pn = rmul(Ri, pan) + cross(r, pf);
      }
      OS2R_STAMP(13 + (NQ - 1 - i));
    }
  }

  // ---- 2c. accelerations, outward; the base accelerates by -g (gravity as a fictitious force) ----
  T vs[NQ];  // predicted velocity v* = qd + dt*qdd
  // the inward pass's hand-over (u, U, 1/D) of joint i + 1 is requested while joint i is worked on: a lone
  // wave sits out every LDS round trip that is requested where it is needed.  (Declared out here: what arrived for the
  // last joint is what the unit-torque sweep of step 3 starts with.)
  T hu[2], hd[2];
  V3<T> hua[2], hul[2];
  {
    V3<T> aa = mk<T>(0, 0, 0), al = mk<T>(0, 0, -par.gravity());
    auto request = [&](int i, int b) { hu[b] = L(kU + i); hua[b] = ldv(kUa, i); hul[b] = ldv(kUl, i); hd[b] = L(kDi + i); };
    auto arrived = [&](int b) {
      asm volatile("" : "+v"(hu[b]), "+v"(hd[b]), "+v"(hua[b].x), "+v"(hua[b].y), "+v"(hua[b].z), "+v"(hul[b].x), "+v"(hul[b].y), "+v"(hul[b].z));
      asm volatile("" ::: "memory");
    };
    request(0, 0);
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int b = i & 1;
      arrived(b);
      if (i + 1 < NQ) request(i + 1, b ^ 1);
      const int ax = md.axis(i);
      const V3<T> r = mk(md.rpos(i, 0), md.rpos(i, 1), md.rpos(i, 2));
      V3<T> sq = mk<T>(0, 0, 0);
      add_comp(sq, ax, qd[i]);
      T Ri[9];
      joint_rotation<T>(md, i, sn[i], cs[i], Ri);
      const V3<T> pa_ = rtmul(Ri, aa);
      const V3<T> pl_ = rtmul(Ri, al + cross(aa, r));
      aa = pa_ + cross(wv[i], sq);
      al = pl_ + cross(vv[i], sq);
      const T qdd = (hu[b] - dot(hua[b], aa) - dot(hul[b], al)) * hd[b];
      add_comp(aa, ax, qdd);
      vs[i] = qd[i] + dt * qdd;
    }
  }

  OS2R_STAMP(1);
  // ---- 3. factor of the inverse of M~ = M + dt*diag(d), straight from the articulated-body quantities ----
  // The inward pass maps joint torques to the "innovations" eps = A tau with A unit upper triangular
  // (eps_i collects the torques of joint i and of its descendants k > i), the outward pass maps
  // nu = D^-1 eps to accelerations with A^T, so
  //        M~^-1 = A^T D^-1 A = Lc Lc^T,   Lc[r][c] = A[c][r] / sqrt(D_c)   (lower triangular, r >= c).
  // Lc is therefore THE Cholesky factor of M~^-1 (it is unique), and it costs the unit-torque inward
  // sweep only: neither M~^-1 itself, nor its outward sweep, nor a factorisation is needed.
  // A[i][k] = uk[k][i] is the joint-space force that a unit torque at joint k >= i leaves at joint i.
  constexpr int NB = NQ;
  constexpr int kLc = 0;                       // Lc[i][k], k <= i, at kLc + i*(i+1)/2 + k (LDS mirror, after the pass)
  auto Lcs = [&](int i, int k) -> T& { return L(kLc + i * (i + 1) / 2 + k); };
  T Lc[NQ][NQ];   // lower triangle, also mirrored to LDS for the contact-row setup
  T Ldi[NQ];      // 1 / Lc[i][i] = sqrt(D_i)
  {
    T uk[NQ][NQ];            // uk[k][i]: joint-space force of column k at joint i (i <= k)
    V3<T> pn_[NQ], pf_[NQ];  // bias force of column k, expressed in the current body
    // inward: bodies NQ-1 .. 0; all columns are swept together, so every joint rotation, U_i and 1/D_i
    // is fetched once -- and a body ahead of its use (round 4): left to the compiler the seven LDS reads of a body sat
    // right in front of their first use, and a lone wave waited out every round trip (five times ~100 cycles per iteration)
    V3<T> mUa[2], mUl[2];
    T mdi[2];
    auto request_m = [&](int i, int b) { mUa[b] = ldv(kUa, i); mUl[b] = ldv(kUl, i); mdi[b] = L(kDi + i); };
    auto arrived_m = [&](int b) {
      asm volatile("" : "+v"(mdi[b]), "+v"(mUa[b].x), "+v"(mUa[b].y), "+v"(mUa[b].z), "+v"(mUl[b].x), "+v"(mUl[b].y), "+v"(mUl[b].z));
      asm volatile("" ::: "memory");
    };
    // (the last joint's U and 1/D are still in the registers of the outward pass above.  fp64 only: the fp32 kernels are
    // built for two waves per SIMD, which hide the round trips, and have no registers to spare)
    constexpr bool kAhead = sizeof(T) == 8;
    if constexpr (kAhead) { mUa[(NQ - 1) & 1] = hua[(NQ - 1) & 1]; mUl[(NQ - 1) & 1] = hul[(NQ - 1) & 1]; mdi[(NQ - 1) & 1] = hd[(NQ - 1) & 1]; }
#pragma unroll
    for (int i = NQ - 1; i >= 0; --i) {
      if constexpr (kAhead) {
        if (i < NQ - 1) arrived_m(i & 1);
        if (i > 0) { request_m(i - 1, (i - 1) & 1); __builtin_amdgcn_sched_barrier(kPinDs); }
      } else {
        request_m(i, i & 1);
      }
      const V3<T> Ua = mUa[i & 1], Ul = mUl[i & 1];
      const T di = mdi[i & 1];
      // column i starts here with a unit torque
      uk[i][i] = T(1);
      // columns k > i arrive from body i+1: express in body i, project on the joint
      if (i < NQ - 1) {
        const V3<T> r = mk(md.rpos(i + 1, 0), md.rpos(i + 1, 1), md.rpos(i + 1, 2));
        T Ri[9];
        joint_rotation<T>(md, i + 1, sn[i + 1], cs[i + 1], Ri);
#pragma unroll
        for (int k = i + 1; k < NQ; ++k) {
          const V3<T> f = rmul(Ri, pf_[k]);
          const V3<T> n = rmul(Ri, pn_[k]) + cross(r, f);
          uk[k][i] = -comp(n, md.axis(i));
          if (i > 0) {                         // nothing is propagated below the base joint
            const T s_ = uk[k][i] * di;
            pn_[k] = n + s_ * Ua;
            pf_[k] = f + s_ * Ul;
          }
        }
      }
      if (i > 0) {
        pn_[i] = di * Ua;
        pf_[i] = di * Ul;
      }
      // column i of the factor: A[i][r] / sqrt(D_i) for the rows r >= i
      Ldi[i] = rsqrt_t(di);                    // sqrt(D_i)
      const T sdi = di * Ldi[i];               // 1 / sqrt(D_i)
#pragma unroll
      for (int r = i; r < NQ; ++r) Lc[r][i] = r == i ? sdi : uk[r][i] * sdi;
    }
  }
  OS2R_STAMP(19);

  OS2R_STAMP(2);
  // ---- 4. whitening: y = Lc^-1 v ----
  // The constraint rows are solved in the coordinates y: for a row with Jacobian J_r,
  // J_r v = G_r y and the velocity response Minv J_r^T dl = Lc (G_r^T dl) with G_r = J_r Lc, so one
  // vector per row serves both the residual and the update, and because Lc is lower triangular
  // G_r of a contact on body b (and of the friction row of joint b) has only b+1 non-zeros.
  // This halves the per-row storage and work of the sweep compared with keeping J_r and
  // Minv J_r^T.  The rows are written straight into registers (Gr) and stay there through the sweeps;
  // only the factor is mirrored to LDS (the articulated-body slots are dead by now) for the row set-up.
  T fb[NQ];   // joint friction impulse bound
  T mub[NB];  // ground friction coefficient of the bodies that can touch
#pragma unroll
  for (int j = 0; j < NQ; ++j) fb[j] = par.friction(j);
#pragma unroll
  for (int b = 0; b < NB; ++b) mub[b] = ((CMASK >> b) & 1u) ? par.mu(b) : T(0);
  __builtin_amdgcn_sched_barrier(DR ? kPinDs : kPinVmem);   // requested here, needed by the solver
  T y[NQ];
  T idj[NQ];  // reciprocal of Minv[j][j] = |row j of Lc|^2 (joint friction rows)
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    T mjj = 0;
#pragma unroll
    for (int k = 0; k <= j; ++k) { mjj += Lc[j][k] * Lc[j][k]; Lcs(j, k) = Lc[j][k]; }
    idj[j] = mjj > T(0) ? rcp_t(mjj) : T(0);
  }
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    T acc_ = vs[i];
#pragma unroll
    for (int k = 0; k < i; ++k) acc_ -= Lc[i][k] * y[k];
    y[i] = opaque(acc_ * Ldi[i]);   // a plain value for the sweeps: its product is not to be fused into their updates
  }
  // only the change of y is mapped back, so an idle solve leaves v bit-identical.  The start value waits in registers in
  // the kernels built for the default solver settings, in per-lane LDS slots in the others (their register file is full
  // during the exact solves, see kMuInLds below)
  constexpr bool kStdExact = sizeof(T) == 8 && SOLVER == kSolverExact && MD::kStatic;   // the kernels that have registers to spare
  // (the generic fp64 kernels park it whatever their solver: with rows for every body their register file is full)
  constexpr bool kParkY0 = sizeof(T) == 8 && !kStdExact && (SOLVER != kSolverSweeps || !MD::kStatic);
  constexpr int kY0Slot = NQ * (NQ + 1) / 2 + 3 * NQ + NQ;   // behind the factor's mirror and the multipliers' slots
  static_assert(kY0Slot + NQ <= 8 * NQ, "per-lane LDS slots");
  T y0[NQ];
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    if constexpr (kParkY0) L(kY0Slot + i) = y[i];
    else y0[i] = y[i];
  }

  OS2R_STAMP(3);
  // ---- 5. ground contact candidates -> one point contact per body ----
  T dn[NB], dx[NB], dy[NB], erv[NB];  // dn/dx/dy: reciprocal of J Minv J^T = |G_r|^2 per row (0: row off)
  T Gr[NB][3][NQ];                    // the rows G = J Lc, held in registers through the sweeps
  bool act[NB];
  bool wave_act[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) { act[b] = false; wave_act[b] = false; }
  if (CMASK != 0u) {
    T Rw[9], ow[3];
    V3<T> aw[NQ], jo[NQ];  // world joint axes and joint origins
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      // first chunk of this body's candidate coordinates: requested ahead of the frame arithmetic that covers it
      constexpr int CHS = 4;
      T bA[3 * CHS], bB[3 * CHS];
      auto request = [&](int kk, int kend, T (&d)[3 * CHS]) {
#pragma unroll
        for (int j = 0; j < 3 * CHS; ++j) d[j] = (kk + j / 3 < kend) ? mconst->cand_p[kk + j / 3][j % 3] : T(0);   // scalar loads: wave-uniform operands
      };
      if constexpr (MD::kStatic) {
        if ((CMASK >> b) & 1u) {
          request(md.cand_begin(b), md.cand_begin(b + 1), bA);
          asm volatile("" ::: "memory");   // issued here, not where the scan starts
        }
      }
      const V3<T> r = mk(md.rpos(b, 0), md.rpos(b, 1), md.rpos(b, 2));
      T Rb[9];
      joint_rotation<T>(md, b, sn[b], cs[b], Rb);
      if (b == 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) Rw[k] = Rb[k];
        ow[0] = r.x; ow[1] = r.y; ow[2] = r.z;
      } else {
        const V3<T> t = rmul(Rw, r);
        ow[0] += t.x; ow[1] += t.y; ow[2] += t.z;
        T n[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) n[3 * i + j] = Rw[3 * i] * Rb[j] + Rw[3 * i + 1] * Rb[3 + j] + Rw[3 * i + 2] * Rb[6 + j];
#pragma unroll
        for (int k = 0; k < 9; ++k) Rw[k] = n[k];
      }
      const int ax = md.axis(b);
      aw[b] = mk(pick3(ax, Rw[0], Rw[1], Rw[2]), pick3(ax, Rw[3], Rw[4], Rw[5]), pick3(ax, Rw[6], Rw[7], Rw[8]));
      jo[b] = mk(ow[0], ow[1], ow[2]);
      if (!((CMASK >> b) & 1u)) continue;
      OS2R_STAMP(4);
      const int k0 = md.cand_begin(b), k1 = md.cand_begin(b + 1);
      T W = 0, sx = 0, sy = 0, sz = 0;
      // skip the scan when the body's candidate sphere is clear of the contact band for every
      // lane of the wave (no candidate can have z < margin then, so W stays 0 exactly)
      const T zc = Rw[6] * md.cand_center(b, 0) + Rw[7] * md.cand_center(b, 1) + Rw[8] * md.cand_center(b, 2) + ow[2];
      const bool near = zc - md.cand_radius(b) < margin;
      if constexpr (MD::kStatic) {   // the chunk requested ahead is taken here, on either side of the sphere test
#pragma unroll
        for (int j = 0; j < 3 * CHS; ++j) asm volatile("" : "+s"(bA[j]));
        asm volatile("" ::: "memory");
      }
      if constexpr (COUNT) wc.scanned += __ballot(near) != 0ull ? 1u : 0u;
      if (__ballot(near) != 0ull)
      // Candidate table: wave-shared LDS copy (broadcast reads).  The scan is software
      // pipelined by hand -- two register buffers of 4 candidates, the next chunk is requested
      // before the current one is consumed -- because at one wave per SIMD nothing else hides
      // the LDS latency.
      {
        constexpr int CH = 4;
        auto fetch = [&](int k, T (&d)[3 * CH]) {
#pragma unroll
          for (int j = 0; j < 3 * CH; ++j) d[j] = cand_lds[3 * k + j];
        };
        auto consume = [&](const T (&d)[3 * CH]) {
#pragma unroll
          for (int c = 0; c < CH; ++c) {
            const T px = d[3 * c], py = d[3 * c + 1], pz = d[3 * c + 2];
            const T z = Rw[6] * px + Rw[7] * py + Rw[8] * pz + ow[2];
            const T wgt = fmax_t(margin - z, T(0));
            W += wgt; sx += wgt * px; sy += wgt * py; sz += wgt * pz;
          }
        };
        if constexpr (MD::kStatic) {
          const T mo = margin - ow[2];
          // compile-time candidate counts: straight-line code.  Scalar loads return out of order, so the only
          // wait there is for them is "all arrived": the next chunk is requested right after the current one
          // has arrived (the asm uses pin that point) and the current chunk's arithmetic covers its flight.
          auto arrived = [&](T (&d)[3 * CH]) {
#pragma unroll
            for (int j = 0; j < 3 * CH; ++j) asm volatile("" : "+s"(d[j]));
            asm volatile("" ::: "memory");
          };
          // margin - z with the frame origin folded into the constant term (one add less), and the terms of the
          // coordinate the link's points share in runs (ga; a compile-time property of the table) taken once per
          // run: its product enters the height through `base`, its moment is the run's weight times the value
          const int ga = MD::cand_group_axis(b), a1 = (ga + 1) % 3, a2 = (ga + 2) % 3;   // folded: b is an unrolled loop index
          T S[3] = {T(0), T(0), T(0)};
          T Wg = T(0), pg = T(0), base = mo;
          auto weigh = [&](const T (&d)[3 * CH], int kk) {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
              if (kk + c < k1) {
                if (MD::cand_starts_run(b, kk + c, ga)) {
                  W += Wg; S[ga] = fma_t(pg, Wg, S[ga]);
                  Wg = T(0); pg = d[3 * c + ga];
                  base = fma_t(-Rw[6 + ga], pg, mo);
                }
                const T p1 = d[3 * c + a1], p2 = d[3 * c + a2];
                const T mz = fma_t(-Rw[6 + a1], p1, fma_t(-Rw[6 + a2], p2, base));
                const T wgt = fmax_t(mz, T(0));
                Wg += wgt; S[a1] = fma_t(wgt, p1, S[a1]); S[a2] = fma_t(wgt, p2, S[a2]);
              }
            }
          };
          static_assert(CH == CHS, "chunk size of the request issued ahead");
#pragma unroll
          for (int kk = k0; kk < k1; kk += 2 * CH) {
            arrived(bA);
            if (kk + CH < k1) request(kk + CH, k1, bB);
            weigh(bA, kk);
            if (kk + CH < k1) {
              arrived(bB);
              if (kk + 2 * CH < k1) request(kk + 2 * CH, k1, bA);
              weigh(bB, kk + CH);
            }
          }
          W += Wg; S[ga] = fma_t(pg, Wg, S[ga]);
          sx = S[0]; sy = S[1]; sz = S[2];
        } else {
        T bufA[3 * CH], bufB[3 * CH];
        int k = k0;
        const int kpair = k0 + ((k1 - k0) / (2 * CH)) * (2 * CH);   // whole pairs of chunks
        if (k < kpair) {
          fetch(k, bufA);
#pragma unroll 1
          for (; k < kpair; k += 2 * CH) {
            fetch(k + CH, bufB);
            consume(bufA);
            if (k + 2 * CH < kpair) fetch(k + 2 * CH, bufA);
            consume(bufB);
          }
        }
#pragma unroll 1
        for (; k < k1; ++k) {   // remainder (none for the reference's models: 8 | count)
          const T px = cand_lds[3 * k], py = cand_lds[3 * k + 1], pz = cand_lds[3 * k + 2];
          const T z = Rw[6] * px + Rw[7] * py + Rw[8] * pz + ow[2];
          const T wgt = fmax_t(margin - z, T(0));
          W += wgt; sx += wgt * px; sy += wgt * py; sz += wgt * pz;
        }
        }
      }
      OS2R_STAMP(5);
      act[b] = W > T(0);
      wave_act[b] = __ballot(act[b]) != 0ull;
      if constexpr (COUNT) { wc.row_bodies += wave_act[b] ? 1u : 0u; wc.lane_contacts += (unsigned)__popcll(__ballot(act[b])); }
      if (wave_act[b]) {
        // the factor rows this body needs are requested first; the Jacobian below covers the LDS trip
        T lcb[NQ][NQ];
#pragma unroll
        for (int j = 0; j < NQ; ++j)
#pragma unroll
          for (int k = 0; k < NQ; ++k)
            if (j <= b && k <= j) lcb[j][k] = Lcs(j, k);
        __builtin_amdgcn_sched_barrier(kPinDs);
        const T iw = act[b] ? rcp_t(W) : T(0);
        const V3<T> pc = mk(sx * iw, sy * iw, sz * iw);
        const V3<T> pw = rmul(Rw, pc) + jo[b];
        // gap-based non-penetration (Stewart-Trinkle): an open gap may close within the step,
        // a penetration is pushed out at the capped error-reduction velocity
        const T gap = pw.z;
        const T e = erp * (-gap) * inv_dt;
        erv[b] = gap >= T(0) ? -gap * inv_dt : (e > max_erv ? max_erv : e);
        T Jn[NQ], Jx[NQ], Jy[NQ];  // Jacobian rows of the contact point: normal z, tangents x, y
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
          if (j <= b) {
            const V3<T> c = cross(aw[j], pw - jo[j]);
            Jx[j] = c.x; Jy[j] = c.y; Jn[j] = c.z;
          }
        }
        // G = J Lc (b+1 non-zeros), d = |G|^2
        T sdn = 0, sdx = 0, sdy = 0;
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
          if (k <= b) {
            T gn = 0, gx = 0, gy = 0;
#pragma unroll
            for (int j = k; j < NQ; ++j)
              if (j <= b) { const T l = lcb[j][k]; gn += Jn[j] * l; gx += Jx[j] * l; gy += Jy[j] * l; }
            Gr[b][0][k] = gn; Gr[b][1][k] = gx; Gr[b][2][k] = gy;
            sdn += gn * gn; sdx += gx * gx; sdy += gy * gy;
          }
        }
        // reciprocals once per iteration of the physics, not once per row update
        dn[b] = (act[b] && sdn > T(0)) ? rcp_t(sdn) : T(0);
        dx[b] = (act[b] && sdx > T(0)) ? rcp_t(sdx) : T(0);
        dy[b] = (act[b] && sdy > T(0)) ? rcp_t(sdy) : T(0);
      }
    }
  }

  OS2R_STAMP(6);
  // ---- 6. projected Gauss-Seidel on the (whitened) velocities ----
  T ln[NB], lx[NB], ly[NB], lf[NQ];
#pragma unroll
  for (int b = 0; b < NB; ++b) { ln[b] = 0; lx[b] = 0; ly[b] = 0; }
#pragma unroll
  for (int j = 0; j < NQ; ++j) lf[j] = 0;
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    fb[j] = opaque(fb[j] * dt);
    idj[j] = fb[j] > T(0) ? idj[j] : T(0);
  }

  // Phase 1 (pgs_normal_iters sweeps): normal rows and joint-friction rows only; its normal
  // impulses fix the tangential box bounds +-mu*lambda_n.  Phase 2 (pgs_iters sweeps): all rows
  // with those fixed bounds -- a boxed LCP with a symmetric PSD matrix, i.e. a convex QP with a
  // unique velocity solution.  pgs_normal_iters == 0 selects the coupled pyramid (bounds follow
  // the current normal impulse inside the sweep), which is ill-posed for a slender leg sliding
  // at mu ~ 1 (Painleve) and is kept for experiments only.
  // A row that is switched off has reciprocal rd == 0: lam = l - res*0 = l, and l already lies
  // inside its box, so the row reproduces itself without any select.
  // The rows G are constant over the sweeps and live in registers (Gr, written by the row set-up): reading
  // them from LDS inside the sweep left every row waiting a full round trip, which at one wave per SIMD
  // nothing hides (that was half of the solver's time); parking them in LDS between set-up and sweeps
  // cost a wait too (+1.2 % without it).
  T moved = T(0);   // energy moved by the measuring sweep of a group: sum over the rows of |residual * impulse change|
  auto contact_row = [&](int b, int row, T target, T rd, T& l, T lo, T hi, bool upper, bool measure) {
    // Explicit fused operations, in source order: with contraction left to the compiler a sum of two products
    // (`g0*y0 + g1*y1` when the target is zero) has two fused forms, and which one it picks differs between the
    // specialised and the general sweep code -- a lane's result would depend on its company in the wave.
    T g[NQ];
    T res = -target;
#pragma unroll
    for (int k = 0; k < NQ; ++k)
      if (k <= b) { g[k] = Gr[b][row][k]; res = fma_t(g[k], y[k], res); }
    T lam = fma_t(-res, rd, l);
    lam = lam < lo ? lo : lam;
    if (upper) lam = lam > hi ? hi : lam;
    const T dl = lam - l;
    l = lam;
    if (measure) moved = fma_t(fabs_t(res), fabs_t(dl), moved);
#pragma unroll
    for (int k = 0; k < NQ; ++k)
      if (k <= b) y[k] = fma_t(g[k], dl, y[k]);
  };
  auto joint_rows = [&](bool measure) {
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      // joint Coulomb friction row: J = e_j, G = row j of Lc, d = Minv[j][j]
      T res = Lc[j][0] * y[0];
#pragma unroll
      for (int k = 1; k < NQ; ++k)
        if (k <= j) res = fma_t(Lc[j][k], y[k], res);
      T lam = fma_t(-res, idj[j], lf[j]);
      lam = lam < -fb[j] ? -fb[j] : lam;
      lam = lam > fb[j] ? fb[j] : lam;
      const T dl = lam - lf[j];
      lf[j] = lam;
      if (measure) moved = fma_t(fabs_t(res), fabs_t(dl), moved);
#pragma unroll
      for (int k = 0; k < NQ; ++k)
        if (k <= j) y[k] = fma_t(Lc[j][k], dl, y[k]);
    }
  };
  const bool fixed_box = pgs_normal_iters > 0;
  // every row fetched above has landed before the sweeps start (otherwise the waits sit inside them)
  __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
  T limfix[NB];
  // The bodies in contact are nearly always a suffix of the chain (the distal links reach the
  // ground first; a fallen robot lies on links 2..4), so both phases exist once per suffix as
  // straight-line code -- no per-body branch, no copies at the joins -- besides the general form.
  auto normal_sweep = [&](auto first) {
    constexpr int kFirst = decltype(first)::value;   // < 0: general form, test every body
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (!((CMASK >> b) & 1u)) continue;
      if (kFirst >= 0 ? b < kFirst : !wave_act[b]) continue;
      contact_row(b, 0, erv[b], dn[b], ln[b], T(0), T(0), false, false);
    }
    joint_rows(false);
  };
  // live_only: the sweep runs under the mask of a few lanes (the exact finish's loop): a body that none of them touches
  // has reciprocals 0 in all of them -- its rows would reproduce themselves -- and is skipped
  // (which lanes touch body b, as a lane mask found once: the vote of the lanes at work is then scalar arithmetic on it and the
  // execution mask -- a vote on `dn[b] > 0` inside the sweep goes through a 0 / 1 vector register and a vector compare whose
  // result the branch has to wait for)
  unsigned long long touch[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) touch[b] = ((CMASK >> b) & 1u) ? __builtin_amdgcn_ballot_w64(dn[b] > T(0)) : 0ull;
  auto sweep = [&](auto coupled, auto first, auto measure_, bool live_only = false) {
    constexpr bool measure = decltype(measure_)::value;
    constexpr int kFirst = decltype(first)::value;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (!((CMASK >> b) & 1u)) continue;
      if (kFirst >= 0 ? b < kFirst : !wave_act[b]) continue;
      if (live_only && (touch[b] & __builtin_amdgcn_ballot_w64(true)) == 0ull) continue;
      contact_row(b, 0, erv[b], dn[b], ln[b], T(0), T(0), false, measure);
      if (decltype(coupled)::value) limfix[b] = mub[b] * ln[b];   // the coupled pyramid, experiments only
      const T lim = limfix[b];
      contact_row(b, 1, T(0), dx[b], lx[b], -lim, lim, true, measure);
      contact_row(b, 2, T(0), dy[b], ly[b], -lim, lim, true, measure);
    }
    joint_rows(measure);
  };
  // Phase 2 in groups of kPgsGroup sweeps.  The last sweep of a group measures what it moved; an environment
  // that moved no more than pgs_tol is done and sits out the remaining groups (a divergent branch: the wave
  // skips a group once none of its lanes is live, and a lane's result does not depend on the other lanes).
  // With pgs_tol == 0 only an exact fixed point stops an environment, which changes nothing: every further
  // sweep would reproduce the state bit for bit.
  // the tolerance in a vector register for the whole solve: as a kernel argument it is re-fetched from the argument
  // segment (a scalar load and a wait of ~150 cycles that nothing covers) at every check
  T tol_v = pgs_tol;
  asm volatile("" : "+v"(tol_v));
  auto grouped_sweeps = [&](auto coupled, auto first) {
    constexpr int kFirst = decltype(first)::value;
    bool live = true;
    auto group = [&](int n, bool check) {
      if constexpr (COUNT) {
        const unsigned long long lv = __ballot(live && n > 0);   // the wave runs the group if any of its lanes is live
        if (lv != 0ull) {
          unsigned nb = 0;
#pragma unroll
          for (int b = 0; b < NB; ++b)
            if ((CMASK >> b) & 1u) nb += (kFirst >= 0 ? b >= kFirst : wave_act[b]) ? 1u : 0u;
          wc.sweeps += (unsigned)n; wc.body_sweeps += (unsigned)n * nb;
          wc.live_lane_sweeps += (unsigned)n * (unsigned)__popcll(lv);
        }
      }
      if (live && n > 0) {
        if (n == kPgsGroup) {
#pragma unroll
          for (int k = 0; k + 1 < kPgsGroup; ++k) sweep(coupled, first, std::false_type{});
        } else {
          for (int k = 0; k + 1 < n; ++k) sweep(coupled, first, std::false_type{});
        }
        if (check) {
          moved = T(0);
          sweep(coupled, first, std::true_type{});
          live = moved > tol_v;
        } else {
          sweep(coupled, first, std::false_type{});
        }
      }
    };
    int it = 0;
    for (; it + kPgsGroup < pgs_iters; it += kPgsGroup) group(kPgsGroup, true);   // a check after the last sweep would decide nothing
    group(pgs_iters - it, false);
  };
  // ---- exact finish (fp64; kExact* above).  The rows of phase 2 in sweep order: ----
  // f(slot, nz, g, target, rd, lambda&, lo, hi, upper): row `slot` has the non-zeros g[0..nz], the reciprocal rd of its squared norm
  // and an impulse in [lo, hi] (upper == false: no upper bound)
  auto each_row = [&](auto first, auto&& f) {
    constexpr int kFirst = decltype(first)::value;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (!((CMASK >> b) & 1u)) continue;
      if (kFirst >= 0 ? b < kFirst : !wave_act[b]) continue;
      const T lim = limfix[b];
      f(3 * b + 0, b, Gr[b][0], erv[b], dn[b], ln[b], T(0), T(0), false);
      f(3 * b + 1, b, Gr[b][1], T(0), dx[b], lx[b], -lim, lim, true);
      f(3 * b + 2, b, Gr[b][2], T(0), dy[b], ly[b], -lim, lim, true);
    }
#pragma unroll
    for (int j = 0; j < NQ; ++j) f(3 * NB + j, j, Lc[j], T(0), idj[j], lf[j], -fb[j], fb[j], true);
  };
  // The same, visiting only the rows whose bit is set in the wave-uniform mask U (bit 3b + t: row t of body b, bit 3 NB + j:
  // joint j) -- a body none of whose rows is in U, and the joint rows when none of them is, are passed over with ONE branch:
  // a wave that owns its SIMD pays ~45 cycles for every taken branch, more than for the test itself (round 4: the solve's
  // fourteen rows times four passes were 56 skip decisions, most of them taken)
  auto each_row_in = [&](auto first, unsigned U, auto&& f) {
    constexpr int kFirst = decltype(first)::value;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (!((CMASK >> b) & 1u)) continue;
      if (kFirst >= 0 ? b < kFirst : !wave_act[b]) continue;
      if (((U >> (3 * b)) & 7u) == 0u) continue;
      const T lim = limfix[b];
      if ((U >> (3 * b + 0)) & 1u) f(3 * b + 0, b, Gr[b][0], erv[b], dn[b], ln[b], T(0), T(0), false);
      if ((U >> (3 * b + 1)) & 1u) f(3 * b + 1, b, Gr[b][1], T(0), dx[b], lx[b], -lim, lim, true);
      if ((U >> (3 * b + 2)) & 1u) f(3 * b + 2, b, Gr[b][2], T(0), dy[b], ly[b], -lim, lim, true);
    }
    if (((U >> (3 * NB)) & ((1u << NQ) - 1u)) != 0u) {
#pragma unroll
      for (int j = 0; j < NQ; ++j)
        if ((U >> (3 * NB + j)) & 1u) f(3 * NB + j, j, Lc[j], T(0), idj[j], lf[j], -fb[j], fb[j], true);
    }
  };
  // One exact solve of the rows strictly inside their box, every other row held at its bound.  Returns whether a
  // bound cut the step short (that row then sits on its bound and the caller solves again with the smaller set).
  // Runs under the mask of the lanes that need it; nothing in it looks at another lane.
  auto exact_solve = [&](auto first, bool test_consistency) -> bool {
    constexpr int NT = NQ * (NQ + 1) / 2;
    auto tri = [](int i, int j) { return i * (i + 1) / 2 + j; };   // j <= i
    // Branch-free on purpose (bitwise logic on the predicates, selects, min / max): written with && / || and
    // conditional statements every row became a divergent branch of its own -- a hundred of them per solve, each a
    // handful of scalar instructions on the execution mask plus the jump, which a lone wave pays in full.
    auto is_free = [](T l, T lo, T hi, bool upper) { return (l > lo) & (upper ? (l < hi) : true); };
    T S[NT], h[NQ];
#pragma unroll
    for (int k = 0; k < NT; ++k) S[k] = T(0);
#pragma unroll
    for (int k = 0; k < NQ; ++k) h[k] = T(0);
    // The multipliers wait between the passes: in
    // registers in the kernels built for the default solver settings, in free per-lane LDS slots (behind the factor's
    // mirror) in the others, whose register file is full -- they also carry the sweeps-only solver and, for run-time
    // models, rows for every body (scratch otherwise: 12-370 B per lane)
    constexpr bool kMuInLds = !kStdExact;
    constexpr int kMuSlot = NQ * (NQ + 1) / 2;
    static_assert(kMuSlot + 3 * NB + NQ <= 8 * NQ, "per-lane LDS slots");
    T mu_reg[kMuInLds ? 1 : 3 * NB + NQ];
    auto mu_put = [&](int slot, T v) { if constexpr (kMuInLds) L(kMuSlot + slot) = v; else mu_reg[slot] = v; };
    auto mu_get = [&](int slot) -> T { if constexpr (kMuInLds) return L(kMuSlot + slot); else return mu_reg[slot]; };
    OS2R_ISA_MARK(9);
    OS2R_STAMP(8);    // (what precedes a solve inside the loop stays with the sweeps)
    // which rows are free for some lane at work here (two or three of the 64, typically, with two free rows each): found once
    // per solve -- the impulses do not move before the last pass -- branch-free, kept as one scalar bit mask
    unsigned U = 0u;
    const unsigned long long at_work = __builtin_amdgcn_ballot_w64(true);
    each_row(first, [&](int slot, int, const T (&)[NQ], T, T, T& l, T lo, T hi, bool upper) {
      // (a body that none of the lanes at work touches has its rows switched off in all of them -- impulse 0 in a box [0, 0], never
      // free --: three votes less per such body, typically one of the three a wave carries when a lone lane is in a burst)
      if (slot < 3 * NB && (touch[slot / 3] & at_work) == 0ull) return;
      // (a ballot per comparison: the ballot of their conjunction is lowered through a 0 / 1 register and a second compare)
      const unsigned long long fm = __builtin_amdgcn_ballot_w64(l > lo) & (upper ? __builtin_amdgcn_ballot_w64(l < hi) : ~0ull);
      U |= fm != 0ull ? (1u << slot) : 0u;
    });
    // pass 1: S = sum over the free rows of g g^T, h = -sum g w, w = g.y - target
    // A row that is free for none of the lanes at work here -- typically two or three of the 64 -- adds exact zeros to S
    // and h, has mu = 0 and keeps its impulse: the wave skips it in every pass (void for every lane, so a lane's result
    // does not depend on its company).
    each_row_in(first, U, [&](int, int nz, const T (&g)[NQ], T target, T rd, T& l, T lo, T hi, bool upper) {
      const bool fr = is_free(l, lo, hi, upper);
      // the row's weight as a number the optimiser cannot see through: it would turn the products below back into
      // selects of every entry (two v_cndmask per double) or into a branch around the row.  A free row weighs 1 / |g|^2 (round 5):
      // rows of very different mobility -- the tangential row along the boom has a thousandth of the others' squared norm -- would
      // otherwise put a direction of S below the regularisation, where the proximal iterations converge at 0.87 per iteration,
      // the consistency test takes the set for inconsistent and the solves zigzag between two bounds of that row
      const T f = opaque(fr ? rd : T(0));
      T w = -target;
#pragma unroll
      for (int k = 0; k < NQ; ++k)
        if (k <= nz) w = fma_t(g[k], y[k], w);
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        if (i <= nz) {
          const T gs = f * g[i];
          h[i] = fma_t(-gs, w, h[i]);
#pragma unroll
          for (int j = 0; j < NQ; ++j)
            if (j <= i) S[tri(i, j)] = fma_t(gs, g[j], S[tri(i, j)]);
        }
      }
    });
    OS2R_ISA_MARK(10);
    OS2R_STAMP(26);   // pass 1: S, h
    T tr = T(0);
#pragma unroll
    for (int i = 0; i < NQ; ++i) tr += S[tri(i, i)];
    // (values that are products and feed sums below are made opaque: left to the compiler, `a + b * c` is fused or not
    // depending on the instantiation -- the specialised and the general form of a wave -- and on the branch taken,
    // and a lane's result would depend on its company)
    const T eps = opaque(tr > T(0) ? T(kExactEps) * tr : T(1));   // no free row: S = 0 and h = 0, the step is zero
    // S + eps I = Lf D Lf^T, natural order (symmetric positive definite: no pivoting).  Lf overwrites S (strict lower
    // part), Di holds the reciprocal pivots.
    T Di[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      T u[NQ];   // u[k] = Lf[j][k] D[k]: the entries of row j before their division by the pivot
#pragma unroll
      for (int k = 0; k < NQ; ++k) {
        if (k < j) {
          T t = S[tri(j, k)];
#pragma unroll
          for (int m = 0; m < NQ; ++m)
            if (m < k) t = fma_t(-S[tri(k, m)], u[m], t);
          u[k] = t;
        }
      }
      T dj = S[tri(j, j)] + eps;
#pragma unroll
      for (int k = 0; k < NQ; ++k) {
        if (k < j) {
          const T ljk = opaque(u[k] * Di[k]);
          dj = fma_t(-ljk, u[k], dj);
          S[tri(j, k)] = ljk;
        }
      }
      Di[j] = rcp_t(dj);
    }
    // kExactProx proximal iterations: d_1 from h, d_k from h + eps d_(k-1); ds their sum
    T d[NQ], ds[NQ];
#pragma unroll
    for (int it = 0; it < kExactProx; ++it) {
      T z[NQ];
#pragma unroll
      for (int i = 0; i < NQ; ++i) z[i] = it == 0 ? h[i] : fma_t(eps, d[i], h[i]);
#pragma unroll
      for (int i = 0; i < NQ; ++i)
#pragma unroll
        for (int k = 0; k < NQ; ++k)
          if (k < i) z[i] = fma_t(-S[tri(i, k)], z[k], z[i]);
#pragma unroll
      for (int i = 0; i < NQ; ++i) z[i] = opaque(z[i] * Di[i]);
#pragma unroll
      for (int i = NQ - 1; i >= 0; --i)
#pragma unroll
        for (int k = 0; k < NQ; ++k)
          if (k > i) z[i] = fma_t(-S[tri(k, i)], z[k], z[i]);
#pragma unroll
      for (int i = 0; i < NQ; ++i) { d[i] = z[i]; ds[i] = it == 0 ? z[i] : ds[i] + z[i]; }
    }
    OS2R_ISA_MARK(11);
    OS2R_STAMP(27);   // factorisation + proximal solves
    // pass 2: impulses of the free rows from the residuals, mu = -(K w + g.ds) / eps, and whether the full step
    // would take a row out of its box
    const T ieps = -rcp_t(eps);
    // (which lanes' full step leaves a box: gathered as a lane mask in scalar registers -- a bool that is or-ed across the rows'
    // scalar branches lives in a vector register as 0 / 1, a select and an or per row)
    unsigned long long cut_lanes = 0ull;
    // (the variant that also measures the residuals before and after the step runs when a lane of the wave asks for it)
    T found = T(0), left = T(0);
    auto pass2 = [&](auto measure_) {
      each_row_in(first, U, [&](int slot, int nz, const T (&g)[NQ], T target, T rd, T& l, T lo, T hi, bool upper) {
        const bool fr = is_free(l, lo, hi, upper);
        const T f = opaque(fr ? opaque(ieps * rd) : T(0));
        T w = -target;
#pragma unroll
        for (int k = 0; k < NQ; ++k)
          if (k <= nz) w = fma_t(g[k], y[k], w);
        T r = T(kExactProx) * w;
#pragma unroll
        for (int k = 0; k < NQ; ++k)
          if (k <= nz) r = fma_t(g[k], ds[k], r);
        const T m = opaque(r * f);
        mu_put(slot, m);
        const T full = l + m;
        cut_lanes |= __builtin_amdgcn_ballot_w64(full < lo) | (upper ? __builtin_amdgcn_ballot_w64(full > hi) : 0ull);
        if constexpr (decltype(measure_)::value) {
          T wl = w;
#pragma unroll
          for (int k = 0; k < NQ; ++k)
            if (k <= nz) wl = fma_t(g[k], d[k], wl);
          const T wf = opaque(fr ? w : T(0)), wlf = opaque(fr ? wl : T(0));
          found = fma_t(wf, wf, found);
          left = fma_t(wlf, wlf, left);
        }
      });
    };
    if (__ballot(test_consistency) != 0ull) pass2(std::true_type{});
    else pass2(std::false_type{});
    const unsigned long long incons_lanes = __builtin_amdgcn_ballot_w64(test_consistency) & ~cut_lanes & __builtin_amdgcn_ballot_w64(left > T(kExactIncons) * found);
    const bool incons = __builtin_amdgcn_inverse_ballot_w64(incons_lanes);
    OS2R_ISA_MARK(12);
    OS2R_STAMP(28);   // pass 2: impulses, cut test
    // the largest feasible fraction of the step: the wave computes it when one of its lanes needs it, and only the lanes
    // whose full step leaves a box take it
    T alpha = T(1);
    if ((cut_lanes | incons_lanes) != 0ull) {
      T a = incons ? T(kExactNoBound) : T(1);
      each_row_in(first, U, [&](int slot, int, const T (&)[NQ], T, T, T& l, T lo, T hi, bool upper) {
        const T m = mu_get(slot);
        const bool up = m > T(0);
        const bool bounded = (m < T(0)) | (upper ? up : false);
        const T room = (up ? hi : lo) - l;
        const T lim = opaque(room * rcp_t(bounded ? m : T(1)));   // same sign as m, so lim >= 0
        a = (bounded & (lim < a)) ? lim : a;
      });
      cut_lanes |= incons_lanes & __builtin_amdgcn_ballot_w64(a < T(kExactNoBound));   // the long step ends on a bound: the same as a cut
      alpha = __builtin_amdgcn_inverse_ballot_w64(cut_lanes) ? a : T(1);
    }
    const bool cut = __builtin_amdgcn_inverse_ballot_w64(cut_lanes);
    OS2R_ISA_MARK(13);
    OS2R_STAMP(29);   // step length
    // the velocity takes the last proximal iterate (exact on the free rows), the impulses their multipliers; a row
    // that the cut step has taken to its bound (the room left is below kExactSnap of what it had) is set on it
#pragma unroll
    for (int i = 0; i < NQ; ++i) y[i] = fma_t(alpha, d[i], y[i]);
    if (cut_lanes != 0ull) {
      each_row_in(first, U, [&](int slot, int, const T (&)[NQ], T, T, T& l, T lo, T hi, bool upper) {
        const T m = mu_get(slot);
        T nl = fma_t(alpha, m, l);
        const bool at_hi = cut & (upper ? ((m > T(0)) & ((hi - nl) <= T(kExactSnap) * (hi - l))) : false);
        const bool at_lo = cut & (m < T(0)) & ((nl - lo) <= T(kExactSnap) * (l - lo));
        nl = at_hi ? hi : nl;
        nl = at_lo ? lo : nl;
        nl = fmax_t(nl, lo);
        if (upper) nl = fmin_t(nl, hi);
        l = nl;
      });
    } else {
      // (a row of U that is not free for this lane has the multiplier 0 and sits inside its box)
      each_row_in(first, U, [&](int slot, int, const T (&)[NQ], T, T, T& l, T lo, T hi, bool upper) {
        T nl = l + mu_get(slot);
        nl = fmax_t(nl, lo);
        if (upper) nl = fmin_t(nl, hi);
        l = nl;
      });
    }
    OS2R_ISA_MARK(14);
    OS2R_STAMP(30);   // apply
    return cut;
  };
  // Phase 2 with the exact finish: warm_first(NQ) sweeps, the last of them measured; from then on an environment that is
  // still live solves (again while a bound cuts its step short, pgs_exact solves at most per physics iteration) and
  // takes one measured sweep, until the sweep moves no more than pgs_tol or pgs_iters sweeps are spent.
  auto exact_sweeps = [&](auto first) {
    constexpr int kFirst = decltype(first)::value;
    // (a run-time value on purpose: with a compile-time count the first sweeps are unrolled into one scheduling region whose
    // hoisted operand reads cost the 5-dof kernels thirty registers more)
    int kExactFirst = warm_first(NQ);
    asm volatile("" : "+s"(kExactFirst));
    const int nfirst = pgs_iters < kExactFirst ? pgs_iters : kExactFirst;
    for (int k = 0; k + 1 < nfirst; ++k) sweep(std::false_type{}, first, std::false_type{});
    if (nfirst > 0) { moved = T(0); sweep(std::false_type{}, first, std::true_type{}); }
    int sweeps = nfirst, solves = 0;
    bool live = nfirst > 0 && moved > tol_v && sweeps < pgs_iters;
    if constexpr (COUNT) {
      unsigned nb = 0;
#pragma unroll
      for (int b = 0; b < NB; ++b)
        if ((CMASK >> b) & 1u) nb += (kFirst >= 0 ? b >= kFirst : wave_act[b]) ? 1u : 0u;
      wc.sweeps += (unsigned)nfirst; wc.body_sweeps += (unsigned)nfirst * nb; wc.live_lane_sweeps += (unsigned)nfirst * 64u;
    }
    // the loop is wave-uniform, what a lane does in it is its own business: a lane whose step was cut solves again
    // while another takes its sweep
    // (rotated by hand -- the vote at the bottom, a guard in front: the compiler does not rotate a loop whose header votes, and
    // with the exit in the header it copied the 23 loop-carried values -- impulses, velocities -- to a second set of registers
    // and back at the top of every round)
    if (__ballot(live) != 0ull) do {
      const bool do_solve = live && solves < pgs_exact;
      if constexpr (COUNT) {
        const unsigned long long sv = __ballot(do_solve);
        if (sv != 0ull) { wc.exact_solves += 1u; wc.lane_exact_solves += (unsigned)__popcll(sv); }
      }
      bool again = false;
      OS2R_STAMP(8);
      if (do_solve) {
        again = exact_solve(first, solves > 0);
        ++solves;
        again = again && solves < pgs_exact;
      }
      OS2R_STAMP(18);   // the exact solves of phase 2 (its sweeps stay on stamp 8)
      const bool do_sweep = live && !again;
      if constexpr (COUNT) {
        const unsigned long long wv = __ballot(do_sweep);
        if (wv != 0ull) {
          unsigned nb = 0;
#pragma unroll
          for (int b = 0; b < NB; ++b)
            if ((CMASK >> b) & 1u) nb += (kFirst >= 0 ? b >= kFirst : wave_act[b]) ? 1u : 0u;
          wc.sweeps += 1u; wc.body_sweeps += nb; wc.live_lane_sweeps += (unsigned)__popcll(wv);
        }
      }
      if (do_sweep) {
        moved = T(0);
        sweep(std::false_type{}, first, std::true_type{}, true);
        ++sweeps;
        live = moved > tol_v && sweeps < pgs_iters;
      }
    } while (__ballot(live) != 0ull);
  };
  int first_act = NB;
  bool is_suffix = true;
#pragma unroll
  for (int b = NB - 1; b >= 0; --b) {
    if (!((CMASK >> b) & 1u)) continue;
    if (wave_act[b]) { is_suffix = is_suffix && first_act == next_cand_body<CMASK, NB>(b); first_act = b; }
  }
  // The exact finish between its warm start and the impulses it leaves for the environment's next physics iteration.
  auto exact_carried = [&](auto first) {
    constexpr int kFirst = decltype(first)::value;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (!((CMASK >> b) & 1u)) continue;
      if (kFirst >= 0 ? b < kFirst : !wave_act[b]) continue;
      const unsigned long long ap_lanes = touch[b] & __builtin_amdgcn_ballot_w64(((carry.act >> b) & 1u) != 0u);
      if (ap_lanes == 0ull) continue;
      const bool ap = __builtin_amdgcn_inverse_ballot_w64(ap_lanes);   // a contact now and in the last iteration
      const T lim = limfix[b];
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        T& l = t == 0 ? ln[b] : (t == 1 ? lx[b] : ly[b]);
        const T prev = t == 0 ? carry.ln[b] : (t == 1 ? carry.lx[b] : carry.ly[b]);
        T nl = t == 0 ? fmax_t(prev, T(0)) : fmin_t(fmax_t(prev, -lim), lim);
        nl = ap ? nl : l;
        const T dl = nl - l;
        l = nl;
#pragma unroll
        for (int k = 0; k < NQ; ++k)
          if (k <= b) y[k] = fma_t(Gr[b][t][k], dl, y[k]);
      }
    }
    {
      const bool jp = (carry.act & kCarryJoints) != 0u;   // (false in the first iteration after a reset only)
#pragma unroll
      for (int j = 0; j < NQ; ++j) {
        T nl = fmin_t(fmax_t(carry.lf[j], -fb[j]), fb[j]);
        nl = jp ? nl : lf[j];
        const T dl = nl - lf[j];
        lf[j] = nl;
#pragma unroll
        for (int k = 0; k < NQ; ++k)
          if (k <= j) y[k] = fma_t(Lc[j][k], dl, y[k]);
      }
    }
    exact_sweeps(first);
    unsigned act = kCarryJoints;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (!((CMASK >> b) & 1u)) continue;
      if (kFirst >= 0 ? b < kFirst : !wave_act[b]) continue;
      carry.ln[b] = ln[b]; carry.lx[b] = lx[b]; carry.ly[b] = ly[b];
      act |= (dn[b] > T(0) ? 1u : 0u) << b;
    }
    carry.act = act;
#pragma unroll
    for (int j = 0; j < NQ; ++j) carry.lf[j] = lf[j];
  };
  auto solve_fixed_box = [&](auto first) {
    for (int it = 0; it < pgs_normal_iters; ++it) normal_sweep(first);
    OS2R_STAMP(7);
#pragma unroll
    for (int b = 0; b < NB; ++b) limfix[b] = opaque(mub[b] * ln[b]);   // (a plain value: the exact finish subtracts from it)
    if constexpr (sizeof(T) == 8 && SOLVER == kSolverExact) {
      exact_carried(first);
    } else if constexpr (sizeof(T) == 8 && SOLVER == kSolverBoth) {
      if (pgs_exact > 0) exact_carried(first);
      else grouped_sweeps(std::false_type{}, first);
    } else {
      grouped_sweeps(std::false_type{}, first);
    }
  };
  if (!fixed_box) {
#pragma unroll
    for (int b = 0; b < NB; ++b) limfix[b] = 0;
    grouped_sweeps(std::true_type{}, std::integral_constant<int, -1>{});
  } else if (!(is_suffix && first_act < NB && for_body<0, NB, CMASK>(first_act, solve_fixed_box))) {
    solve_fixed_box(std::integral_constant<int, -1>{});
  }
  OS2R_STAMP(8);
  // back to joint velocities: v += Lc (y - y0)
  if constexpr (kParkY0) {
#pragma unroll
    for (int i = 0; i < NQ; ++i) y0[i] = L(kY0Slot + i);
  }
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    T acc_ = 0;
#pragma unroll
    for (int k = 0; k < NQ; ++k)
      if (k <= i) acc_ += Lc[i][k] * (y[k] - y0[k]);
    vs[i] += acc_;
  }

  // ---- 7. semi-implicit Euler ----
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    qd[i] = vs[i];
    q[i] += dt * vs[i];
  }
  OS2R_STAMP(9);
}

}  // namespace os2r
