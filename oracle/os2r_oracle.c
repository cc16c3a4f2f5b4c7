/*
 * os2r_oracle.c — CPU restatement (scalar fp64, plain C) of the gym-os2r env-step path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under gym-os2r_amd/ may include, link or
 * call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / reported CPU baseline.
 *
 * What is restated, and from where (paths relative to the reference checkout):
 *
 *   env-step loop      gym_os2r/runtimes/gazebo_runtime.py:65-97 (10 substeps, action held :70-77)
 *   set_action         gym_os2r/tasks/monopod.py:202-236
 *   observation        gym_os2r/tasks/monopod.py:238-272, tasks/monopod_no_norm.py:222-246
 *   done               gym_os2r/tasks/monopod.py:274-298 (reset_space :198)
 *   rewards            gym_os2r/rewards/__init__.py:66-207, rewards/rewards_utils.py:10-122
 *   reset / IK         gym_os2r/randomizers/monopod_no_rand.py:59-98,
 *                      gym_os2r/randomizers/monopod.py:89-128, gym_os2r/utils/reset.py:4-40
 *   randomisation      gym_os2r/randomizers/monopod.py:56-61,182-215
 *   vec-env semantics  gym_os2r/common/vec_env/subproc_vec_env.py:15-21 (auto-reset)
 *
 * PARITY PINNING.  The epilogue (observation / reward / done), tolerance() and the
 * reset IK are pinned by golden vectors generated from the reference's own Python
 * (tests/golden/, tools/gen_golden.py).  The physics of one iteration
 * (`gazebo.run()`, gazebo_runtime.py:76) lives in third-party code that is not in
 * the reference checkout (gym-ignition -> ScenarIO -> Ignition Gazebo ->
 * ign-physics-dartsim -> DART, all unpinned in setup.py:22-26) and cannot be built
 * or imported here: for the dynamics this oracle is PARITY UNPINNED.  It restates
 * the published scheme of that stack — Featherstone's articulated-body algorithm
 * with joint damping taken implicitly in the articulated-inertia projection,
 * semi-implicit Euler, then velocity-level constraint impulses (frictional ground
 * contact + per-dof Coulomb joint friction as a boxed LCP), then position update —
 * and is checked by independent known answers and by
 * invariants (tests/test_oracle_dynamics.py).
 *
 * The dynamics here use dense 6x6 spatial algebra (Featherstone, "Rigid Body
 * Dynamics Algorithms", 2008, Table 7.1 and ch.2 notation) on purpose: the HIP
 * kernels use hand-specialised sparse forms, so agreement is a meaningful check.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/os2r.h"
#include "os2r_oracle.h"

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------- *
 * Counter RNG: Philox4x32-10 (Salmon et al., SC'11, "Parallel random numbers:
 * as easy as 1, 2, 3"; Random123).  Known answers are checked in
 * tests/test_oracle_rng.py.
 * ------------------------------------------------------------------------- */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  const uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)M0 * c0;
    uint64_t p1 = (uint64_t)M1 * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* 53-bit uniform in [0,1) from two 32-bit words */
static double u53(uint32_t a, uint32_t b) {
  return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}

/* two uniforms of block `blk` of (env, stream, ctr) */
void orc_uniform2(uint64_t seed, uint32_t env, uint32_t stream, uint32_t ctr, uint32_t blk,
                  double u[2]) {
  uint32_t c[4] = {env, stream, ctr, blk};
  uint32_t k[2] = {(uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32)};
  uint32_t o[4];
  orc_philox4x32_10(c, k, o);
  u[0] = u53(o[0], o[1]);
  u[1] = u53(o[2], o[3]);
}

/* Box-Muller pair of standard normals from one block */
static void normal2(uint64_t seed, uint32_t env, uint32_t stream, uint32_t ctr, uint32_t blk,
                    double z[2]) {
  double u[2];
  orc_uniform2(seed, env, stream, ctr, blk, u);
  double r = sqrt(-2.0 * log(1.0 - u[0]));
  double th = 6.283185307179586476925286766559 * u[1];
  z[0] = r * cos(th);
  z[1] = r * sin(th);
}

enum { STREAM_ACTION = 1, STREAM_RESET = 2, STREAM_PARAMS = 3, STREAM_GRAVITY = 4 };

/* ------------------------------------------------------------------------- *
 * rewards_utils.py:10-73  _sigmoids ; :76-122 tolerance
 * sigmoid ids: 0 gaussian 1 hyperbolic 2 long_tail 3 reciprocal 4 cosine
 *              5 linear 6 quadratic 7 tanh_squared
 * ------------------------------------------------------------------------- */
static double sigmoid_value(double x, double value_at_1, int sigmoid) {
  switch (sigmoid) {
    case 0: { double scale = sqrt(-2.0 * log(value_at_1)); double t = x * scale; return exp(-0.5 * (t * t)); }
    case 1: { double scale = acosh(1.0 / value_at_1); return 1.0 / cosh(x * scale); }
    case 2: { double scale = sqrt(1.0 / value_at_1 - 1.0); double t = x * scale; return 1.0 / (t * t + 1.0); }
    case 3: { double scale = 1.0 / value_at_1 - 1.0; return 1.0 / (fabs(x) * scale + 1.0); }
    case 4: { double scale = acos(2.0 * value_at_1 - 1.0) / M_PI; double sx = x * scale;
              return fabs(sx) < 1.0 ? (1.0 + cos(M_PI * sx)) / 2.0 : 0.0; }
    case 5: { double scale = 1.0 - value_at_1; double sx = x * scale; return fabs(sx) < 1.0 ? 1.0 - sx : 0.0; }
    case 6: { double scale = sqrt(1.0 - value_at_1); double sx = x * scale; return fabs(sx) < 1.0 ? 1.0 - sx * sx : 0.0; }
    case 7: { double scale = atanh(sqrt(1.0 - value_at_1)); double t = tanh(x * scale); return 1.0 - t * t; }
    default: return NAN;
  }
}

double orc_tolerance(double x, double lower, double upper, double margin, int sigmoid,
                     double value_at_margin) {
  int in_bounds = (lower <= x) && (x <= upper);
  if (margin == 0.0) return in_bounds ? 1.0 : 0.0;
  double d = (x < lower ? lower - x : x - upper) / margin;
  return in_bounds ? 1.0 : sigmoid_value(d, value_at_margin, sigmoid);
}

/* utils/reset.py:4-40 — (hip, knee) that put the foot on the ground for a boom pitch.
 * def6 = upper_leg_length, lower_leg_length, central_pivot_height, length_boom,
 *        hip_offset, clipping_adjust (settings.yaml task_modes/<mode>/definition, mm) */
void orc_leg_joint_angles(const double def6[6], double pitch, double out[2]) {
  double ul = def6[0], ll = def6[1], cph = def6[2], lb = def6[3];
  double lh = (lb * sin(pitch) + cph) / cos(pitch);
  double lleg = lh - def6[4] - def6[5];
  if (lleg > ul + ll) { out[0] = 0.0; out[1] = 0.0; return; }
  double ua = acos((ul * ul + lleg * lleg - ll * ll) / (2.0 * ul * lleg));
  double la = asin(ul * sin(ua) / ll) + ua;
  out[0] = ua;
  out[1] = -la;
}

/* numpy.mod(x + pi, 2*pi) - pi   (tasks/monopod.py:260-261) */
static double py_mod(double a, double b) {
  double m = fmod(a, b);
  if (m != 0.0) { if ((b < 0.0) != (m < 0.0)) m += b; }
  else m = copysign(0.0, b);
  return m;
}
double orc_wrap(double x) { return py_mod(x + M_PI, 2.0 * M_PI) - M_PI; }

/* ------------------------------------------------------------------------- *
 * Observation (tasks/monopod.py:238-272) for one environment.
 * hist1 = action_history[1] at the time of the call.
 * ------------------------------------------------------------------------- */
void orc_observe(const Os2rTaskSpec* ts, const double* q, const double* qd, const double hist1[2],
                 double* obs) {
  for (int d = 0; d < ts->obs_dim; ++d) {
    int s = ts->obs_src[d];
    double lo = ts->obs_low[d], hi = ts->obs_high[d];
    double y;
    switch (ts->obs_kind[d]) {
      case OS2R_OBS_POS_NORM:          y = q[s]; obs[d] = 2.0 * (y - lo) / (hi - lo) - 1.0; break;
      case OS2R_OBS_POS_PERIODIC_NORM: y = orc_wrap(q[s]); obs[d] = 2.0 * (y - lo) / (hi - lo) - 1.0; break;
      case OS2R_OBS_VEL_TANH:          obs[d] = tanh(0.05 * qd[s]); break;
      case OS2R_OBS_TORQUE_NORM:       y = hist1[s]; obs[d] = 2.0 * (y - lo) / (hi - lo) - 1.0; break;
      case OS2R_OBS_POS_RAW:           obs[d] = q[s]; break;
      case OS2R_OBS_POS_PERIODIC_RAW:  obs[d] = orc_wrap(q[s]); break;
      case OS2R_OBS_VEL_RAW:           obs[d] = qd[s]; break;
      case OS2R_OBS_TORQUE_RAW:        obs[d] = hist1[s]; break;
      default: obs[d] = NAN;
    }
  }
}

/* tasks/monopod.py:274-298: done = observation outside reset_space, evaluated the
 * reference's way on the observation itself: reset_space = Box(low+eps, high-eps)
 * with low/high = -1/+1 (normalised task, :183-184,198) or the raw limits
 * (monopod_no_norm.py).  gym Box.contains is all(x>=low) and all(x<=high). */
int orc_done(const Os2rTaskSpec* ts, const double* obs) {
  const double eps = 2.220446049250313e-16;
  int done = 0;
  for (int d = 0; d < ts->obs_dim; ++d) {
    double lo, hi;
    if (ts->normalized) { lo = -1.0 + eps; hi = 1.0 - eps; }
    else { lo = ts->obs_low[d] + eps; hi = ts->obs_high[d] - eps; }
    if (!(obs[d] >= lo) || !(obs[d] <= hi)) done = 1;
  }
  return done;
}

/* rewards/__init__.py.  a0 = actions[0] (current), a1 = actions[1] (previous). */
double orc_reward(const Os2rTaskSpec* ts, const double* obs, const double a0[2], const double a1[2]) {
  double nrm = ts->normalized ? 1.0 : 0.0;
  double H = 0.11 / 1.57 * nrm + 0.11 * (1.0 - nrm);   /* _BALANCE_HEIGHT :77 */
  double bp = ts->idx_pitch_pos >= 0 ? obs[ts->idx_pitch_pos] : NAN;
  switch (ts->reward_id) {
    case OS2R_REWARD_BALANCING_V1:
    case OS2R_REWARD_STANDING_V1:
      return orc_tolerance(bp, H, 4.0 * H, 0.0, 0, 0.1);
    case OS2R_REWARD_BALANCING_V2: {
      double bal = orc_tolerance(bp, H, 4.0 * H, 0.0, 0, 0.1);
      double s0 = orc_tolerance(a0[0], 0.0, 0.0, 1.0, 6, 0.4);
      double s1 = orc_tolerance(a0[1], 0.0, 0.0, 1.0, 6, 0.4);
      return bal * (s0 * s1);
    }
    case OS2R_REWARD_BALANCING_V3: {
      double bal = orc_tolerance(bp, H, 4.0 * H, 0.01, 2, 0.1);
      double s0 = orc_tolerance(a0[0] - a1[0], 0.0, 0.0, 1.0, 6, 0.1);
      double s1 = orc_tolerance(a0[1] - a1[1], 0.0, 0.0, 1.0, 6, 0.1);
      return bal * (s0 * s1);
    }
    case OS2R_REWARD_HOPPING_V1: {
      double bal = orc_tolerance(bp, H, 4.0 * H, 0.0, 0, 0.1);
      double s0 = orc_tolerance(a0[0] - a1[0], 0.0, 0.0, 0.1, 6, 0.0);
      double s1 = orc_tolerance(a0[1] - a1[1], 0.0, 0.0, 0.1, 6, 0.0);
      double hv = ts->idx_yaw_vel >= 0 ? obs[ts->idx_yaw_vel] : NAN;
      double move = orc_tolerance(hv, 0.25, 0.3, 0.15, 7, 0.1);
      return bal * (s0 * s1) * move;
    }
    case OS2R_REWARD_STRAIGHT_V1: {
      double s0 = orc_tolerance(a0[0] / 20.0, 0.0, 0.0, 1.0, 6, 0.0);
      double s1 = orc_tolerance(a0[1] / 20.0, 0.0, 0.0, 1.0, 6, 0.0);
      double sc = (s0 + s1) / 2.0;            /* ndarray.mean() of two elements */
      sc = (4.0 + sc) / 5.0;
      double hip = obs[ts->idx_hip_pos], knee = obs[ts->idx_knee_pos];
      double hr = orc_tolerance(hip, 0.0, 0.0, 1.0, 5, 0.1);
      double kr = orc_tolerance(knee, 0.0, 0.0, 1.0, 5, 0.1);
      return hr * kr * sc;
    }
    default: return NAN;
  }
}

/* ------------------------------------------------------------------------- *
 * Dense spatial algebra (6-vectors [angular; linear], 6x6 row-major matrices).
 * ------------------------------------------------------------------------- */
typedef double V6[6];
typedef double M6[36];
typedef double M3[9];

static void m3_mul(const M3 a, const M3 b, M3 c) {
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
    double s = 0; for (int k = 0; k < 3; ++k) s += a[3 * i + k] * b[3 * k + j]; c[3 * i + j] = s; }
}
static void m3_transpose(const M3 a, M3 t) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) t[3 * i + j] = a[3 * j + i]; }
static void m3_vec(const M3 a, const double v[3], double o[3]) {
  for (int i = 0; i < 3; ++i) o[i] = a[3 * i] * v[0] + a[3 * i + 1] * v[1] + a[3 * i + 2] * v[2];
}
static void skew(const double v[3], M3 s) {
  s[0] = 0; s[1] = -v[2]; s[2] = v[1]; s[3] = v[2]; s[4] = 0; s[5] = -v[0]; s[6] = -v[1]; s[7] = v[0]; s[8] = 0;
}
static void cross3(const double a[3], const double b[3], double c[3]) {
  c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
}
static void axis_rot(int axis, double q, M3 r) {
  double c = cos(q), s = sin(q);
  for (int i = 0; i < 9; ++i) r[i] = 0;
  if (axis == 0) { r[0] = 1; r[4] = c; r[5] = -s; r[7] = s; r[8] = c; }
  else if (axis == 1) { r[4] = 1; r[0] = c; r[2] = s; r[6] = -s; r[8] = c; }
  else { r[8] = 1; r[0] = c; r[1] = -s; r[3] = s; r[4] = c; }
}
static void m6_set_blocks(M6 m, const M3 a, const M3 b, const M3 c, const M3 d) {
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
    m[6 * i + j] = a[3 * i + j]; m[6 * i + 3 + j] = b[3 * i + j];
    m[6 * (i + 3) + j] = c[3 * i + j]; m[6 * (i + 3) + 3 + j] = d[3 * i + j]; }
}
static void m6_vec(const M6 m, const V6 v, V6 o) {
  for (int i = 0; i < 6; ++i) { double s = 0; for (int k = 0; k < 6; ++k) s += m[6 * i + k] * v[k]; o[i] = s; }
}
static void m6t_vec(const M6 m, const V6 v, V6 o) {
  for (int i = 0; i < 6; ++i) { double s = 0; for (int k = 0; k < 6; ++k) s += m[6 * k + i] * v[k]; o[i] = s; }
}
static void m6_mul(const M6 a, const M6 b, M6 c) {
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) {
    double s = 0; for (int k = 0; k < 6; ++k) s += a[6 * i + k] * b[6 * k + j]; c[6 * i + j] = s; }
}
static void m6_transpose(const M6 a, M6 t) { for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) t[6 * i + j] = a[6 * j + i]; }
static double dot6(const V6 a, const V6 b) { double s = 0; for (int i = 0; i < 6; ++i) s += a[i] * b[i]; return s; }

/* motion cross product operator crm(v) and force operator crf(v) = -crm(v)^T (RBDA eq. 2.31-2.33) */
static void crm(const V6 v, M6 m) {
  M3 w, l, z = {0};
  skew(v, w); skew(v + 3, l);
  m6_set_blocks(m, w, z, l, w);
}
static void crf(const V6 v, M6 m) {
  M6 c; crm(v, c);
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) m[6 * i + j] = -c[6 * j + i];
}

/* Plücker motion transform parent->child for rotation E (child<-parent) and child origin r
 * in parent coordinates: X = [E 0; -E r^ , E]  (RBDA eq. 2.24-2.26) */
static void plux(const M3 E, const double r[3], M6 X) {
  M3 rs, Ers, mErs, z = {0};
  skew(r, rs); m3_mul(E, rs, Ers);
  for (int i = 0; i < 9; ++i) mErs[i] = -Ers[i];
  m6_set_blocks(X, E, z, mErs, E);
}

/* spatial inertia about the body-frame origin (RBDA eq. 2.63) */
static void spatial_inertia(double m, const double c[3], const double ic6[6], M6 I) {
  M3 ic = {ic6[0], ic6[1], ic6[2], ic6[1], ic6[3], ic6[4], ic6[2], ic6[4], ic6[5]};
  M3 cs, cst, ccT, a, b, bt, d = {0};
  skew(c, cs); m3_transpose(cs, cst); m3_mul(cs, cst, ccT);
  for (int i = 0; i < 9; ++i) { a[i] = ic[i] + m * ccT[i]; b[i] = m * cs[i]; bt[i] = m * cst[i]; }
  d[0] = d[4] = d[8] = m;
  m6_set_blocks(I, a, b, bt, d);
}

/* per-environment physical parameters */
typedef struct {
  double mass_scale[OS2R_MAX_DOF], damping[OS2R_MAX_DOF], friction[OS2R_MAX_DOF], mu[OS2R_MAX_DOF];
  double gravity_z;
} EnvParams;

static void nominal_params(const Os2rModel* m, EnvParams* p) {
  for (int i = 0; i < OS2R_MAX_DOF; ++i) {
    p->mass_scale[i] = 1.0; p->damping[i] = m->damping[i]; p->friction[i] = m->friction[i]; p->mu[i] = m->mu[i];
  }
  p->gravity_z = m->gravity_z;
}

/* ------------------------------------------------------------------------- *
 * Articulated-body forward dynamics + inverse of the damping-augmented mass
 * matrix, for one environment.  Outputs may be NULL.
 *   qdd   [nq]       unconstrained acceleration
 *   minv  [nq*nq]    (M + dt*diag(damping))^-1
 *   rw    [nq][9]    world orientation of each body
 *   ow    [nq][3]    world position of each body origin
 * ------------------------------------------------------------------------- */
static void dynamics(const Os2rModel* md, const EnvParams* ep, double dt, const double* q,
                     const double* qd, const double* tau_full, double* qdd, double* minv,
                     double (*rw)[9], double (*ow)[3]) {
  const int n = md->nq;
  M6 X[OS2R_MAX_DOF], IA[OS2R_MAX_DOF];
  V6 S[OS2R_MAX_DOF], v[OS2R_MAX_DOF], c[OS2R_MAX_DOF], pA[OS2R_MAX_DOF], U[OS2R_MAX_DOF];
  double D[OS2R_MAX_DOF], u[OS2R_MAX_DOF];

  /* pass 1: kinematics, velocity-product terms, rigid-body inertias and bias forces */
  for (int i = 0; i < n; ++i) {
    M3 rq, rj, E;
    M3 rf; memcpy(rf, md->rfix[i], sizeof(M3));
    axis_rot(md->axis[i], q[i], rq);
    m3_mul(rf, rq, rj);        /* child orientation in parent */
    m3_transpose(rj, E);
    plux(E, md->rpos[i], X[i]);
    for (int k = 0; k < 6; ++k) S[i][k] = 0;
    S[i][md->axis[i]] = 1.0;
    V6 vj; for (int k = 0; k < 6; ++k) vj[k] = S[i][k] * qd[i];
    if (i == 0) { for (int k = 0; k < 6; ++k) v[i][k] = vj[k]; }
    else { V6 t; m6_vec(X[i], v[i - 1], t); for (int k = 0; k < 6; ++k) v[i][k] = t[k] + vj[k]; }
    M6 cm; crm(v[i], cm); m6_vec(cm, vj, c[i]);
    spatial_inertia(md->mass[i] * ep->mass_scale[i], md->com[i], md->icom[i], IA[i]);
    V6 Iv; m6_vec(IA[i], v[i], Iv);
    M6 cf; crf(v[i], cf); m6_vec(cf, Iv, pA[i]);
    if (rw) {
      if (i == 0) { memcpy(rw[0], rj, sizeof(M3)); memcpy(ow[0], md->rpos[0], 3 * sizeof(double)); }
      else { m3_mul(rw[i - 1], rj, rw[i]); double t[3]; m3_vec(rw[i - 1], md->rpos[i], t);
             for (int k = 0; k < 3; ++k) ow[i][k] = ow[i - 1][k] + t[k]; }
    }
  }
  /* pass 2: articulated inertias, inward */
  for (int i = n - 1; i >= 0; --i) {
    m6_vec(IA[i], S[i], U[i]);
    D[i] = dot6(S[i], U[i]) + dt * ep->damping[i];
    u[i] = tau_full[i] - ep->damping[i] * qd[i] - dot6(S[i], pA[i]);
    if (i > 0) {
      M6 Ia, Xt, t1, t2; V6 pa, t;
      for (int r = 0; r < 6; ++r) for (int s = 0; s < 6; ++s) Ia[6 * r + s] = IA[i][6 * r + s] - U[i][r] * U[i][s] / D[i];
      m6_vec(Ia, c[i], t);
      for (int k = 0; k < 6; ++k) pa[k] = pA[i][k] + t[k] + U[i][k] * u[i] / D[i];
      m6_transpose(X[i], Xt); m6_mul(Xt, Ia, t1); m6_mul(t1, X[i], t2);
      for (int k = 0; k < 36; ++k) IA[i - 1][k] += t2[k];
      m6t_vec(X[i], pa, t);
      for (int k = 0; k < 6; ++k) pA[i - 1][k] += t[k];
    }
  }
  /* pass 3: accelerations, outward; the base accelerates upward by -g (RBDA 7.3) */
  if (qdd) {
    V6 a_prev = {0, 0, 0, 0, 0, -ep->gravity_z};
    for (int i = 0; i < n; ++i) {
      V6 ap; m6_vec(X[i], a_prev, ap);
      for (int k = 0; k < 6; ++k) ap[k] += c[i][k];
      qdd[i] = (u[i] - dot6(U[i], ap)) / D[i];
      for (int k = 0; k < 6; ++k) a_prev[k] = ap[k] + S[i][k] * qdd[i];
    }
  }
  /* unit-torque responses with the same factorisation: column k of the inverse */
  if (minv) {
    for (int k = 0; k < n; ++k) {
      double uk[OS2R_MAX_DOF]; V6 p = {0};
      for (int i = n - 1; i >= 0; --i) {
        uk[i] = (i == k ? 1.0 : 0.0) - dot6(S[i], p);
        if (i > 0) { V6 pa, t; for (int r = 0; r < 6; ++r) pa[r] = p[r] + U[i][r] * uk[i] / D[i];
                     m6t_vec(X[i], pa, t); memcpy(p, t, sizeof(V6)); }
      }
      V6 a_prev = {0};
      for (int i = 0; i < n; ++i) {
        V6 ap; m6_vec(X[i], a_prev, ap);
        double x = (uk[i] - dot6(U[i], ap)) / D[i];
        minv[i * n + k] = x;
        for (int r = 0; r < 6; ++r) a_prev[r] = ap[r] + S[i][r] * x;
      }
    }
  }
}

void orc_dynamics(const Os2rModel* md, double dt, const double* mass_scale, const double* damping,
                  double gravity_z, const double* q, const double* qd, const double* tau_full,
                  double* qdd, double* minv, double* rw, double* ow) {
  EnvParams ep; nominal_params(md, &ep);
  if (mass_scale) memcpy(ep.mass_scale, mass_scale, md->nq * sizeof(double));
  if (damping) memcpy(ep.damping, damping, md->nq * sizeof(double));
  ep.gravity_z = gravity_z;
  dynamics(md, &ep, dt, q, qd, tau_full, qdd, minv, (double(*)[9])rw, (double(*)[3])ow);
}

/* ------------------------------------------------------------------------- *
 * One physics iteration (the work of one `gazebo.run()`, gazebo_runtime.py:76).
 *
 * Two contact models:
 *   ORC_CONTACT_CENTROID    THE SPECIFICATION (what the HIP kernels implement):
 *                           per body one point contact at the weighted centroid of the candidates
 *                           inside the 1 mm band, gap-based normal target, two-phase fixed-box PGS.
 *   ORC_CONTACT_PER_VERTEX  oracle-only comparison model after the scheme SURVEY.md Appendix B
 *                           attributes to the reference's backend [recollection of upstream
 *                           behaviour, not checkable here]: one contact per penetrating candidate
 *                           (z < 0), error-reduction target only (no open-gap term), friction
 *                           pyramid coupled to the current normal impulse, cfg->pgs_iters sweeps.
 *                           Used by tests/test_oracle_contact.py to measure how far the
 *                           specification sits from that scheme; never on a product path.
 * ------------------------------------------------------------------------- */
typedef struct { double J[OS2R_MAX_DOF], T[OS2R_MAX_DOF], d, target, lambda; int kind, normal_row, body; double bound, point[3]; } Row;
/* kind: 0 normal (lambda>=0), 1 tangential (|lambda|<=mu*lambda_normal), 2 joint friction (|lambda|<=bound) */
#define ORC_MAX_ROWS (3 * OS2R_MAX_CAND + OS2R_MAX_DOF)

void orc_contact_points(const Os2rModel* md, double margin, const double (*rw)[9], const double (*ow)[3],
                        int* active, double (*pw)[3], double* gap) {
  /* Per body: candidates closer than `margin` to the ground plane z=0 (or below it) are weighted
   * by (margin - z); their weighted centroid is the body's single contact point, `gap` its signed
   * height (negative = penetration). */
  int k = 0;
  for (int b = 0; b < md->nq; ++b) {
    double W = 0, pl[3] = {0, 0, 0};
    active[b] = 0;
    for (; k < md->ncand && md->cand_body[k] == b; ++k) {
      const double* p = md->cand_p[k];
      double z = rw[b][6] * p[0] + rw[b][7] * p[1] + rw[b][8] * p[2] + ow[b][2];
      double w = z < margin ? margin - z : 0.0;
      W += w; pl[0] += w * p[0]; pl[1] += w * p[1]; pl[2] += w * p[2];
    }
    if (W > 0.0) {
      double pc[3] = {pl[0] / W, pl[1] / W, pl[2] / W}, t[3];
      m3_vec(rw[b], pc, t);
      for (int i = 0; i < 3; ++i) pw[b][i] = t[i] + ow[b][i];
      gap[b] = pw[b][2];
      active[b] = 1;
    }
  }
}

/* three rows (normal z, tangents x, y) of a point contact at world point pw on body b */
static int add_contact_rows(const Os2rModel* md, const double (*rw)[9], const double (*ow)[3], int b,
                            const double pw[3], double target_n, double mu, Row* rows, int nr) {
  const int n = md->nq;
  double Jp[3][OS2R_MAX_DOF];
  for (int j = 0; j < n; ++j) {
    if (j <= b) {
      double aw[3] = {rw[j][md->axis[j]], rw[j][3 + md->axis[j]], rw[j][6 + md->axis[j]]};
      double dlt[3] = {pw[0] - ow[j][0], pw[1] - ow[j][1], pw[2] - ow[j][2]}, cr[3];
      cross3(aw, dlt, cr);
      Jp[0][j] = cr[0]; Jp[1][j] = cr[1]; Jp[2][j] = cr[2];
    } else { Jp[0][j] = Jp[1][j] = Jp[2][j] = 0.0; }
  }
  const int dirs[3] = {2, 0, 1};  /* normal z, then tangents x, y */
  const int nrow = nr;
  for (int t = 0; t < 3; ++t) {
    Row* r = &rows[nr++];
    memset(r, 0, sizeof(Row));
    for (int j = 0; j < n; ++j) r->J[j] = Jp[dirs[t]][j];
    r->kind = t == 0 ? 0 : 1; r->normal_row = nrow; r->bound = mu; r->body = b;
    r->target = t == 0 ? target_n : 0.0;
    memcpy(r->point, pw, 3 * sizeof(double));
  }
  return nr;
}

/* Unconstrained velocity v* = qd + dt*qdd, the inverse of the damping-augmented mass matrix and
 * the constraint rows of the boxed LCP of this physics iteration.  Returns the number of rows. */
static int build_problem(const Os2rConfig* cfg, int contact_model, const EnvParams* ep, const double* q, const double* qd,
                         const double tau2[2], double* v, double* minv, Row* rows) {
  const Os2rModel* md = &cfg->model;
  const int n = md->nq;
  const double dt = cfg->dt;
  double tau[OS2R_MAX_DOF] = {0}, qdd[OS2R_MAX_DOF];
  double rw[OS2R_MAX_DOF][9], ow[OS2R_MAX_DOF][3];
  tau[md->act_dof[0]] = tau2[0];
  tau[md->act_dof[1]] = tau2[1];
  dynamics(md, ep, dt, q, qd, tau, qdd, minv, rw, ow);
  for (int i = 0; i < n; ++i) v[i] = qd[i] + dt * qdd[i];

  int nr = 0;
  if (cfg->contact && contact_model == ORC_CONTACT_CENTROID) {
    int active[OS2R_MAX_DOF]; double pw[OS2R_MAX_DOF][3], gap[OS2R_MAX_DOF];
    orc_contact_points(md, cfg->contact_margin, rw, ow, active, pw, gap);
    for (int b = 0; b < n; ++b) {
      if (!active[b]) continue;
      /* gap-based non-penetration (Stewart-Trinkle): an open gap may close within the step,
       * a penetration is pushed out at the capped error-reduction velocity */
      double erv;
      if (gap[b] >= 0.0) erv = -gap[b] / dt;
      else { erv = cfg->erp * (-gap[b]) / dt; if (erv > cfg->max_erv) erv = cfg->max_erv; }
      nr = add_contact_rows(md, rw, ow, b, pw[b], erv, ep->mu[b], rows, nr);
    }
  } else if (cfg->contact) {
    /* comparison model: every candidate below the plane is a contact of its own */
    for (int k = 0; k < md->ncand; ++k) {
      const int b = md->cand_body[k];
      const double* p = md->cand_p[k];
      double t[3], pw[3];
      m3_vec(rw[b], p, t);
      for (int i = 0; i < 3; ++i) pw[i] = t[i] + ow[b][i];
      if (!(pw[2] < 0.0)) continue;
      double erv = cfg->erp * (-pw[2]) / dt; if (erv > cfg->max_erv) erv = cfg->max_erv;
      nr = add_contact_rows(md, rw, ow, b, pw, erv, ep->mu[b], rows, nr);
    }
  }
  for (int j = 0; j < n; ++j) {
    if (!(ep->friction[j] > 0.0)) continue;
    Row* r = &rows[nr++];
    memset(r, 0, sizeof(Row));
    r->J[j] = 1.0; r->kind = 2; r->bound = ep->friction[j] * dt; r->target = 0.0; r->body = j; r->normal_row = -1;
  }
  for (int r = 0; r < nr; ++r) {
    Row* R = &rows[r];
    for (int i = 0; i < n; ++i) { double s = 0; for (int j = 0; j < n; ++j) s += minv[i * n + j] * R->J[j]; R->T[i] = s; }
    double s = 0; for (int j = 0; j < n; ++j) s += R->J[j] * R->T[j];
    R->d = s;
  }
  return nr;
}

/* Projected Gauss-Seidel on the velocities, fixed sweep counts, cold start.
 * Phase 1 (normal_iters sweeps): normal rows and joint-friction rows only; its normal
 * impulses fix the tangential box bounds +-mu*lambda_n.  Phase 2 (iters sweeps): all rows
 * with those fixed bounds -- a boxed LCP with a symmetric PSD matrix, i.e. a convex QP with a
 * unique velocity solution.  (With normal_iters == 0 the bounds follow the current normal
 * impulse inside the sweep, the classical coupled pyramid, which is ill-posed for a slender
 * leg sliding at mu ~ 1: Painleve's paradox.)  After the solve rows[r].bound of a tangential
 * row holds the box actually used (fixed-box form) or mu (coupled form). */
/* Stopping rule of phase 2: every 4th sweep (not the last) measures the energy it moved, the sum over the
 * rows of |residual * impulse change| (the decrease of the QP objective up to a factor <= 2); the
 * environment stops sweeping when that is <= tol.  tol == 0 stops at exact fixed points only, which
 * changes nothing (every further sweep would reproduce the state bit for bit). */
#define ORC_PGS_GROUP 4
/* EXPERIMENT (oracle only, off by default; DESIGN.md 3.2 "convergence study"): in phase 2 the three rows of a
 * contact are minimised over exactly as a block (enumeration of the active sets of the 3-variable box QP, first
 * assignment that meets the optimality conditions) instead of row by row.  Measures what an exact per-contact
 * block would buy in closed loop before anything of the kind is built for the GPU. */
#define ORC_EXACT_SMALL_SPEC 0   /* (laboratory: free sets of at most that many rows are solved in the dual; the specification: none, see below) */
/* ---- the laboratory (ORC_EXPERIMENTS) ----
 * The solver studies of rounds 3-5 (docs/studies/) run on switches that change what the solver below does.  They exist ONLY in
 * the laboratory build (make lab -> liboracle_lab.so, -DORC_EXPERIMENTS; used by tests/diag/ and the studies): in the checker
 * (libos2r_oracle.so: what tests/, smoke() and bench.py's cpu_baseline load) every one of them is the compile-time constant of
 * the specification, no setter is exported and the experimental branches are dead code. */
#ifdef ORC_EXPERIMENTS
static int g_block_solve = 0, g_block_kind = 0, g_row_order = 0, g_prox = 2, g_clamp_all = 0, g_small = ORC_EXACT_SMALL_SPEC, g_incons_once = 0, g_lag_box = 0;
static int g_warm = 1, g_first = 3 /* ORC_WARM_FIRST */, g_solve_always = 0, g_stall_incons_only = 0, g_sweep_after_cut = 0, g_max_rounds = 0, g_stop_at_cap = 0;
static double g_incons = 1e-4, g_stall = 0.0;
static int g_pivot = 0;                  /* 1: an inconsistent-set step that would pin a row it has pinned before in this iteration ends phase 2 instead (round 5) */
void orc_set_experimental_pivot(int on) { g_pivot = on; }
static int g_snap = 0;                   /* 1: the warm start puts a tangential row that ended the last iteration ON a bound on the same bound of this iteration's box (round 5; within an env-step only) */
void orc_set_experimental_snap(int on) { g_snap = on; }
static double g_margin = 0.0;            /* > 0: free rows keep that fraction of their box width away from the bounds (round 5 study) */
void orc_set_experimental_margin(double m) { g_margin = m; }
static int g_box_probe = 0;
static double g_box_stat[4];
static _Thread_local double tl_boxn[64];
void orc_set_experimental_box_probe(int on) { g_box_probe = on; for (int i = 0; i < 4; ++i) g_box_stat[i] = 0.0; }
void orc_debug_box_stat(double* out4) { for (int i = 0; i < 4; ++i) out4[i] = g_box_stat[i]; }
static int g_warm_p0 = 0;               /* 1: phase 1 (the normal sweeps that fix the friction box) starts from the remembered impulses (round 5 study) */
void orc_set_experimental_warm_p0(int on) { g_warm_p0 = on; }
static int g_solve_first = 0;            /* k > 0: a physics iteration whose predecessor IN THE SAME env-step took >= k exact solves skips the first sweeps and opens with a solve (round 5) */
void orc_set_experimental_solve_first(int k) { g_solve_first = k; }
static int g_multicut = 0;               /* k > 0: a step that is cut below 1e-k of its length pins every row that would reach its bound within 10 x that fraction, no step taken (round 5) */
void orc_set_experimental_multicut(int k) { g_multicut = k; }
static int g_repin = 0;                  /* 1: a cut step puts every row back that the last sweep released from a bound and that violates the same bound again, at once (round 5) */
void orc_set_experimental_repin(int on) { g_repin = on; }
static int g_equil = 1;                  /* the regularised solve takes every free row with the weight 1 / |g_r|^2 (the specification since round 5; 0: round 4) */
void orc_set_experimental_equil(int on) { g_equil = on; }
static int g_prox_later = 0;             /* proximal iterations of an environment's second and later solves of an iteration (0: g_prox) */
void orc_set_experimental_prox_later(int k) { g_prox_later = k; }
static int g_trace = -1;                 /* ORC_TRACE_SOLVES=k in the environment: iterations with k or more solves are printed */
static long long g_dbg_counter[4];       /* orc_debug_counter: [0] warm-started iterations, [1] of those with an active row that has no remembered impulse */
static int g_dbg_on = 0;
void orc_set_experimental_block_solve(int on) { g_block_solve = on; }
void orc_set_experimental_block_kind(int kind) { g_block_kind = kind; }   /* 0: enumeration of the block's active sets, 1: Gauss-Seidel pass + one exact solve of the rows left free, 2: the pair update */
void orc_set_experimental_row_order(int order) { g_row_order = order; }
void orc_set_experimental_prox(int k) { g_prox = k; }
void orc_set_experimental_incons(double v) { g_incons = v; }
void orc_set_experimental_clamp_all(int on) { g_clamp_all = on; }
void orc_set_experimental_small(int on) { g_small = on; }
void orc_set_experimental_incons_once(int n) { g_incons_once = n; }
void orc_set_experimental_lag_box(int on) { g_lag_box = on; }
void orc_set_experimental_stall(double factor) { g_stall = factor < 0 ? -factor : factor; g_stall_incons_only = factor < 0; }   /* < 0: only rounds with an inconsistent-set step */
void orc_set_experimental_sweep_after_cut(int on) { g_sweep_after_cut = on; }
void orc_set_experimental_rounds(int max_rounds, int stop_at_cap) { g_max_rounds = max_rounds; g_stop_at_cap = stop_at_cap; }
/* first > 0: that many sweeps before the first check; first = -k: k - 1 sweeps, then EVERY environment solves once before its first check */
void orc_set_experimental_warm(int mode, int first) {
  g_warm = mode; g_first = first > 0 ? first : (first < 0 ? -first - 1 : 3); g_solve_always = first < 0;
}
long long orc_debug_counter(int which, int reset) { g_dbg_on = 1; long long v = g_dbg_counter[which & 3]; if (reset) g_dbg_counter[which & 3] = 0; return v; }
#else
enum { g_block_solve = 0, g_block_kind = 0, g_row_order = 0, g_prox = 2, g_clamp_all = 0, g_small = ORC_EXACT_SMALL_SPEC, g_incons_once = 0, g_lag_box = 0,
       g_warm = 1, g_first = 3, g_solve_always = 0, g_stall_incons_only = 0, g_sweep_after_cut = 0, g_max_rounds = 0, g_stop_at_cap = 0,
       g_trace = 0, g_dbg_on = 0, g_prox_later = 0, g_pivot = 0, g_equil = 1, g_repin = 0, g_multicut = 0, g_snap = 0, g_solve_first = 0, g_warm_p0 = 0 };
static const double g_incons = 1e-4, g_stall = 0.0, g_margin = 0.0;
#endif


static int solve_sub(const double A[3][3], const double c[3], const int* idx, int m, double* mu) {
  /* A_FF mu_F = -c_F for the m free variables idx[0..m) by Cramer's rule; 0 if singular */
  if (m == 0) return 1;
  if (m == 1) { const int i = idx[0]; if (!(A[i][i] > 0.0)) return 0; mu[i] = -c[i] / A[i][i]; return 1; }
  if (m == 2) {
    const int i = idx[0], j = idx[1];
    const double det = A[i][i] * A[j][j] - A[i][j] * A[i][j];
    if (!(det > 1e-14 * A[i][i] * A[j][j])) return 0;
    mu[i] = -(A[j][j] * c[i] - A[i][j] * c[j]) / det;
    mu[j] = -(A[i][i] * c[j] - A[i][j] * c[i]) / det;
    return 1;
  }
  const double a = A[0][0], b = A[0][1], cc = A[0][2], d = A[1][1], e = A[1][2], f = A[2][2];
  const double co00 = d * f - e * e, co01 = cc * e - b * f, co02 = b * e - cc * d;
  const double det = a * co00 + b * co01 + cc * co02;
  if (!(det > 1e-14 * a * d * f)) return 0;
  const double co11 = a * f - cc * cc, co12 = b * cc - a * e, co22 = a * d - b * b;
  mu[0] = -(co00 * c[0] + co01 * c[1] + co02 * c[2]) / det;
  mu[1] = -(co01 * c[0] + co11 * c[1] + co12 * c[2]) / det;
  mu[2] = -(co02 * c[0] + co12 * c[1] + co22 * c[2]) / det;
  return 1;
}

/* exact block update of rows r0 (normal), r0+1, r0+2 (tangential, fixed box); returns 0 if it declined */
static int block_update(int n, Row* rows, int r0, double* v, double* moved) {
  Row* R[3] = {&rows[r0], &rows[r0 + 1], &rows[r0 + 2]};
  for (int i = 0; i < 3; ++i) if (!(R[i]->d > 0.0)) return 0;
  double A[3][3], w[3], lam[3], c[3], lo[3], hi[3];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < n; ++k) s += R[i]->J[k] * R[j]->T[k]; A[i][j] = s; }
    double s = -R[i]->target; for (int k = 0; k < n; ++k) s += R[i]->J[k] * v[k];
    w[i] = s; lam[i] = R[i]->lambda;
  }
  lo[0] = 0.0; hi[0] = INFINITY;
  for (int i = 1; i < 3; ++i) { hi[i] = R[i]->bound; lo[i] = -hi[i]; }
  for (int i = 0; i < 3; ++i) { c[i] = w[i]; for (int j = 0; j < 3; ++j) c[i] -= A[i][j] * lam[j]; }
  double best_mu[3] = {lam[0], lam[1], lam[2]}; int found = 0;
  for (int code = 0; code < 27 && !found; ++code) {
    int st[3] = {code % 3, (code / 3) % 3, code / 9};            /* 0 free, 1 lower, 2 upper */
    int ok = 1, idx[3], m = 0; double mu[3], cf[3];
    for (int i = 0; i < 3; ++i) {
      if (st[i] == 2 && !isfinite(hi[i])) ok = 0;
      if (hi[i] - lo[i] <= 0.0 && st[i] != 1) ok = 0;             /* pinned variable: one state only */
      if (st[i] == 0) idx[m++] = i; else mu[i] = st[i] == 1 ? lo[i] : hi[i];
    }
    if (!ok) continue;
    for (int i = 0; i < 3; ++i) { cf[i] = c[i]; for (int j = 0; j < 3; ++j) if (st[j] != 0) cf[i] += A[i][j] * mu[j]; }
    if (!solve_sub(A, cf, idx, m, mu)) continue;
    for (int i = 0; i < 3 && ok; ++i) {
      double g = c[i]; for (int j = 0; j < 3; ++j) g += A[i][j] * mu[j];
      const double tolg = 1e-12 * sqrt(A[i][i]) * (fabs(w[0]) / sqrt(A[0][0]) + fabs(w[1]) / sqrt(A[1][1]) + fabs(w[2]) / sqrt(A[2][2]) + 1e-300);
      if (st[i] == 0) { if (mu[i] < lo[i] || mu[i] > hi[i]) ok = 0; }
      else if (hi[i] - lo[i] > 0.0) { if (st[i] == 1 ? g < -tolg : g > tolg) ok = 0; }
    }
    if (ok) { found = 1; for (int i = 0; i < 3; ++i) best_mu[i] = mu[i]; }
  }
  if (!found) return 0;
  for (int i = 0; i < 3; ++i) {
    const double dl = best_mu[i] - lam[i];
    R[i]->lambda = best_mu[i];
    *moved += fabs(w[i]) * fabs(dl);
    for (int k = 0; k < n; ++k) v[k] += R[i]->T[k] * dl;
  }
  return 1;
}

/* The block update a kernel can afford (experiment; g_block_kind == 1): one Gauss-Seidel pass over the three rows in
 * impulse space (the block's Gram matrix A = G G^T, residuals w), then ONE exact solve of the rows that pass left strictly
 * inside their box, the others held where the pass put them; the result is taken if it stays inside the box, else the
 * pass stands.  Exact whenever the pass identifies the block's active set, which a warm start nearly always does. */
static int block_update_cheap(int n, Row* rows, int r0, double* v, double* moved) {
  Row* R[3] = {&rows[r0], &rows[r0 + 1], &rows[r0 + 2]};
  for (int i = 0; i < 3; ++i) if (!(R[i]->d > 0.0)) return 0;
  double A[3][3], w[3], w0[3], lam[3], l0[3], lo[3], hi[3];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < n; ++k) s += R[i]->J[k] * R[j]->T[k]; A[i][j] = s; }
    double s = -R[i]->target; for (int k = 0; k < n; ++k) s += R[i]->J[k] * v[k];
    w[i] = s; w0[i] = s; lam[i] = R[i]->lambda; l0[i] = lam[i];
  }
  lo[0] = 0.0; hi[0] = INFINITY;
  for (int i = 1; i < 3; ++i) { hi[i] = R[i]->bound; lo[i] = -hi[i]; }
  int fr[3];
  for (int i = 0; i < 3; ++i) {
    double nl = lam[i] - w[i] / A[i][i];
    if (nl < lo[i]) nl = lo[i];
    if (nl > hi[i]) nl = hi[i];
    const double dl = nl - lam[i];
    lam[i] = nl;
    for (int j = 0; j < 3; ++j) w[j] += A[j][i] * dl;
  }
  int idx[3], m = 0;
  for (int i = 0; i < 3; ++i) { fr[i] = lam[i] > lo[i] && lam[i] < hi[i]; if (fr[i]) idx[m++] = i; }
  double mu[3] = {0, 0, 0};
  if (m > 0 && solve_sub(A, w, idx, m, mu)) {
    int ok = 1;
    for (int i = 0; i < 3; ++i) if (fr[i]) { const double t = lam[i] + mu[i]; if (t < lo[i] || t > hi[i]) ok = 0; }
    if (ok) for (int i = 0; i < 3; ++i) if (fr[i]) lam[i] += mu[i];
  }
  for (int i = 0; i < 3; ++i) {
    const double dl = lam[i] - l0[i];
    R[i]->lambda = lam[i];
    *moved += fabs(w0[i]) * fabs(dl);
    for (int k = 0; k < n; ++k) v[k] += R[i]->T[k] * dl;
  }
  return 1;
}

/* g_block_kind == 2: the pair update.  The environments that need an exact solve have, as a rule, the normal and ONE tangential
 * row of one contact free (docs/studies/round4_solver.md): after the Gauss-Seidel pass over the three rows in impulse space,
 * if the normal row and exactly one tangential row are strictly inside their boxes, that pair is solved exactly (2 x 2); if the
 * result leaves the box, the row that violates is set on the bound it violates and the other one re-solved alone (taken if it
 * stays inside); anything else keeps the pass. */
static int block_update_pair(int n, Row* rows, int r0, double* v, double* moved) {
  Row* R[3] = {&rows[r0], &rows[r0 + 1], &rows[r0 + 2]};
  for (int i = 0; i < 3; ++i) if (!(R[i]->d > 0.0)) return 0;
  double A[3][3], w[3], w0[3], lam[3], l0[3], lo[3], hi[3];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < n; ++k) s += R[i]->J[k] * R[j]->T[k]; A[i][j] = s; }
    double s = -R[i]->target; for (int k = 0; k < n; ++k) s += R[i]->J[k] * v[k];
    w[i] = s; w0[i] = s; lam[i] = R[i]->lambda; l0[i] = lam[i];
  }
  lo[0] = 0.0; hi[0] = INFINITY;
  for (int i = 1; i < 3; ++i) { hi[i] = R[i]->bound; lo[i] = -hi[i]; }
  for (int i = 0; i < 3; ++i) {
    double nl = lam[i] - w[i] / A[i][i];
    if (nl < lo[i]) nl = lo[i];
    if (nl > hi[i]) nl = hi[i];
    const double dl = nl - lam[i];
    lam[i] = nl;
    for (int j = 0; j < 3; ++j) w[j] += A[j][i] * dl;
  }
  const int fn = lam[0] > lo[0], fx = lam[1] > lo[1] && lam[1] < hi[1], fy = lam[2] > lo[2] && lam[2] < hi[2];
  if (fn && (fx != fy)) {
    const int t = fx ? 1 : 2;
    const double det = A[0][0] * A[t][t] - A[0][t] * A[0][t];
    if (det > 1e-14 * A[0][0] * A[t][t]) {
      double dn = -(A[t][t] * w[0] - A[0][t] * w[t]) / det, dt_ = -(A[0][0] * w[t] - A[0][t] * w[0]) / det;
      double nn = lam[0] + dn, nt = lam[t] + dt_;
      int ok = nn >= lo[0] && nt >= lo[t] && nt <= hi[t];
      if (!ok && g_block_kind == 2) {
        /* one violator: set it on its bound, re-solve the other alone */
        if (nn < lo[0]) { dn = lo[0] - lam[0]; dt_ = -(w[t] + A[t][0] * dn) / A[t][t]; }
        else { dt_ = (nt < lo[t] ? lo[t] : hi[t]) - lam[t]; dn = -(w[0] + A[0][t] * dt_) / A[0][0]; }
        nn = lam[0] + dn; nt = lam[t] + dt_;
        ok = nn >= lo[0] && nt >= lo[t] && nt <= hi[t];
      }
      if (ok) {
        lam[0] = nn; lam[t] = nt;
      }
    }
  }
  for (int i = 0; i < 3; ++i) {
    const double dl = lam[i] - l0[i];
    R[i]->lambda = lam[i];
    *moved += fabs(w0[i]) * fabs(dl);
    for (int k = 0; k < n; ++k) v[k] += R[i]->T[k] * dl;
  }
  return 1;
}

/* EXPERIMENT (oracle only, off by default): order of the three rows of a contact inside a phase-2 sweep.
 * 0: normal, x, y (the specification)   1: y, x, normal   2: the less mobile tangential row, the other, normal */

/* ---- exact finish of the fixed-box problem (cfg->pgs_exact) ----
 * Gauss-Seidel identifies the active set of most problems within a few sweeps and then crawls on the few whose rows
 * are nearly collinear in the whitened metric (contraction 0.9 per sweep) or degenerate (a sticking foot plus sticking
 * joints: eight rows in five dof).  An environment that has not converged after ORC_EXACT_FIRST sweeps therefore
 * solves its FREE rows (strictly inside their box) exactly, all other rows held at their bounds:
 *     whitened coordinates  Minv = Lc Lc^T,  y = Lc^-1 v,  g_r = Lc^T J_r   (so J_r v = g_r . y,  Minv J_r^T = Lc g_r)
 *     minimum-norm change d of y with  g_r . (y + d) = t_r  for r in F:
 *         (S + eps I) d = -sum_F c_r g_r w_r,   S = sum_F c_r g_r g_r^T (5 x 5, whatever |F| is),  w_r = g_r . y - t_r,
 *         c_r = 1 / |g_r|^2: every free row is taken at unit length (round 5.  Rows differ in mobility by three orders of
 *         magnitude -- a contact on a leg link can hardly move ALONG the boom: |g|^2 = 5e-3 against 2 .. 9 for the other two rows
 *         of the contact -- and unweighted such a row put a direction of S at 1.5e-7 of its trace, below the regularisation:
 *         the proximal iterations converge at 0.87 per iteration there, the consistency test below takes the set for
 *         inconsistent and the solves zigzag between the two bounds of that row until they are spent, leaving up to 3 % of the
 *         velocity wrong: docs/studies/round5_solver.md),
 *         eps = ORC_EXACT_EPS * trace S (= ORC_EXACT_EPS * |F|),  ORC_EXACT_PROX = 2 proximal iterations (d_k from h + eps d_{k-1}) sharpen it
 *         (three up to round 5: with every row at unit length the third changes neither the solves an environment needs nor the
 *         closed loop -- docs/studies/round5_solver.md 6);
 *         the impulses of the free rows follow from the residuals:  mu_r = -c_r (K w_r + g_r . sum_k d_k) / eps
 *     (the regularisation keeps d in the range of the free rows, and 8 sticking rows in 5 dof are no special case).
 * If the full step would take a free row out of its box, the step is cut at the first bound it meets (that row is set
 * on its bound and the solve repeated with the smaller free set); after a full step one ordinary sweep re-tests every
 * row and measures what it moved.  At most `exact` solves per physics iteration; the sweep cap `iters` still holds. */
#define ORC_EXACT_FIRST_COLD(n) ((n) >= 5 ? 6 : 4)   /* (studies only: sweeps before the first check of a cold start, as in round 3) */
#define ORC_EXACT_FIRST(n) (cold_now ? ORC_EXACT_FIRST_COLD(n) : g_first)   /* sweeps before the first check */
#define ORC_EXACT_EPS 1e-6
#define ORC_EXACT_PROX g_prox
#define ORC_EXACT_SNAP 1e-12
/* A free set can be INCONSISTENT: more sticking rows than the degrees of freedom they act on (four or five of them on
 * the three dof of `monopod-fixed`, two sticking contacts in five dof).  The solve then ends at the least-squares point
 * with a residual left on the free rows, and the multipliers do not converge: every round moves them on by the same
 * amount until, tens of rounds later, one of them reaches its bound.  From an environment's second solve of a physics
 * iteration on, a solve that is not cut therefore measures what it leaves on its free rows; if that is more than
 * ORC_EXACT_INCONS of what it found (squared norms), it goes on in the direction of its multipliers -- past the full
 * step -- to the first bound it meets, sets that row on it and solves again, as after a cut. */
#define ORC_EXACT_INCONS g_incons
/* EXPERIMENT (laboratory only, orc_set_experimental_small(1 .. 3); docs/studies/round4_solver.md, round5_solver.md): small free sets
 * solved in the DUAL.  In the regimes the stepper lives in, an environment that needs a solve has two free rows -- the normal and
 * one tangential row of a contact that sticks along one world axis and slides along the other (77 % of the solves of C4, 89 % of
 * C3), or the normals of two contacts (14 %) -- or one (4 %), seldom three and hardly ever more.  With the switch on, a free set of
 * at most that many rows is solved as A mu = -w with A = G_F G_F^T (L D L^T in sweep order, no pivoting): the impulses move by mu
 * and the velocity by Minv J_F^T mu -- the same equality-constrained minimum as the regularised 5 x 5 solve below -- with the same
 * cut at the first bound; a set whose rows are (nearly) dependent -- a pivot below ORC_EXACT_SMALL_PIVOT of its diagonal entry -- or
 * larger takes the regularised solve.  It passes every exactness test.  Built into the kernels twice and measured slower both
 * times: round 4 gathered the rows through per-lane LDS slots; round 5 gathered a lane's two rows with selects while the rows of the
 * wave's free-row mask go by (no LDS) -- as many instructions as the regularised solve it replaces, because that solve already
 * visits only the rows that are free for some lane at work (three or four), and a longer tail (profiles/r05_ab/).  The
 * specification stays with the one solve. */
#define ORC_EXACT_SMALL 3
#define ORC_EXACT_SMALL_PIVOT 1e-8
static _Thread_local int tl_last_small = 0;   /* diagnostics: dual solves among the solves of the last iteration */

static void chol_lower(int n, const double* a, double* l) {   /* a = l l^T, row-major n x n */
  for (int i = 0; i < n * n; ++i) l[i] = 0.0;
  for (int j = 0; j < n; ++j) {
    double s = a[j * n + j];
    for (int k = 0; k < j; ++k) s -= l[j * n + k] * l[j * n + k];
    l[j * n + j] = sqrt(s);
    for (int i = j + 1; i < n; ++i) {
      double t = a[i * n + j];
      for (int k = 0; k < j; ++k) t -= l[i * n + k] * l[j * n + k];
      l[i * n + j] = t / l[j * n + j];
    }
  }
}

static void row_box(const Row* rows, const Row* R, int fixed_box, double* lo, double* hi) {
  if (R->kind == 0) { *lo = 0.0; *hi = INFINITY; }
  else if (R->kind == 1) { *hi = fixed_box ? R->bound : R->bound * rows[R->normal_row].lambda; *lo = -*hi; }
  else { *hi = R->bound; *lo = -*hi; }
}

/* diagnostics (ORC_TRACE_SOLVES=k in the environment: iterations with k or more solves are printed, solve by solve:
 * the free set by row kind -- n normal, t tangential, j joint, lower case; upper case = the row the step ended on) */
static _Thread_local char tl_trace[4096];
static _Thread_local int tl_trace_len = 0;
static void trace_sweep(double moved) {
  if (g_trace <= 0) return;
  char* b = tl_trace + tl_trace_len; char* e = tl_trace + sizeof(tl_trace) - 8;
  if (b >= e) return;
  b += snprintf(b, (size_t)(e - b), " E=%.2g", moved);
  tl_trace_len = (int)(b - tl_trace);
}
static void trace_solve(const Row* rows, int nr, const int* fr, const double* mu, double alpha, int cut, int incons, int nviol, double found, double left) {
  if (g_trace <= 0) return;
  char* b = tl_trace + tl_trace_len; char* e = tl_trace + sizeof(tl_trace) - 8;
  if (b >= e) return;
  *b++ = ' '; *b++ = '[';
  for (int r = 0; r < nr && b < e; ++r) {
    if (!(rows[r].d > 0.0)) { *b++ = '.'; continue; }
    char c = rows[r].kind == 0 ? 'n' : (rows[r].kind == 1 ? 't' : 'j');
    if (!fr[r]) { double lo, hi; row_box(rows, &rows[r], 1, &lo, &hi); *b++ = rows[r].lambda <= lo ? '-' : '+'; continue; }
    *b++ = c;
  }
  b += snprintf(b, (size_t)(e - b), "] a=%.3g%s%s v%d f=%.1e l=%.1e", alpha, cut ? " cut" : "", incons ? " incons" : "", nviol, found, left);
  if (getenv("ORC_TRACE_ROOM")) {   /* how far inside its box every free boxed row sits, as a fraction of the box's width */
    b += snprintf(b, (size_t)(e - b), " room{");
    for (int r = 0; r < nr && b < e; ++r) {
      if (!fr[r] || rows[r].kind == 0) continue;
      double lo, hi; row_box(rows, &rows[r], 1, &lo, &hi);
      const double room = fmin(rows[r].lambda - lo, hi - rows[r].lambda) / (hi - lo);
      b += snprintf(b, (size_t)(e - b), "%d:%.1e ", r, room);
    }
    b += snprintf(b, (size_t)(e - b), "}");
  }
  tl_trace_len = (int)(b - tl_trace);
}

static _Thread_local int tl_incons_steps = 0;   /* steps to the first bound of an inconsistent free set taken so far in this iteration */
static int warm_slot(const Row* rows, int r);
static _Thread_local unsigned tl_was_lo = 0u, tl_was_hi = 0u;   /* rows (by index) that sat on their lower / upper bound before the last sweep */
static _Thread_local unsigned tl_incons_pinned = 0u;   /* rows (by warm slot) that such steps have set on a bound in this iteration */
#ifdef ORC_EXPERIMENTS
/* laboratory diagnostics (orc_debug_free_set_hist): the shape of the free set of every exact solve, by the solve's index within its
 * physics iteration (0 .. 15) -- class 0: the normal and ONE tangential row of one contact, 1: two normals, 2: one row, 3: the three
 * rows of one contact, 4: any other set of at most three rows, 5: four rows or more; +6 when a joint row is among them */
static long long g_fs_hist[16][12];
static _Thread_local int tl_solve_index = 0;
void orc_debug_free_set_hist(long long* out, int reset) {
  memcpy(out, g_fs_hist, sizeof(g_fs_hist));
  if (reset) memset(g_fs_hist, 0, sizeof(g_fs_hist));
}
static void note_free_set(const Row* rows, int nr, const int* fr) {
  int m = 0, nn = 0, nt = 0, nj = 0, bodies = 0, last_body = -1;
  for (int r = 0; r < nr; ++r) {
    if (!fr[r]) continue;
    ++m;
    if (rows[r].kind == 0) ++nn; else if (rows[r].kind == 1) ++nt; else ++nj;
    if (rows[r].kind != 2 && rows[r].body != last_body) { ++bodies; last_body = rows[r].body; }
  }
  int c;
  if (m == 2 && nn == 1 && nt == 1 && bodies == 1) c = 0;
  else if (m == 2 && nn == 2) c = 1;
  else if (m == 1) c = 2;
  else if (m == 3 && nn == 1 && nt == 2 && bodies == 1) c = 3;
  else if (m <= 3) c = 4;
  else c = 5;
  if (nj) c += 6;
  __atomic_fetch_add(&g_fs_hist[tl_solve_index < 15 ? tl_solve_index : 15][c], 1, __ATOMIC_RELAXED);
}
#endif
/* one exact solve of the free rows; returns 1 if the step was cut short by a bound */
static int exact_step(int n, Row* rows, int nr, const double* lc, double* v, int test_consistency) {
  double g[3 * OS2R_MAX_DOF + OS2R_MAX_DOF][OS2R_MAX_DOF], w[3 * OS2R_MAX_DOF + OS2R_MAX_DOF], wgt[3 * OS2R_MAX_DOF + OS2R_MAX_DOF];
  int fr[3 * OS2R_MAX_DOF + OS2R_MAX_DOF];
  double S[OS2R_MAX_DOF][OS2R_MAX_DOF] = {{0}}, h[OS2R_MAX_DOF] = {0};
  double tr = 0.0;
  for (int r = 0; r < nr; ++r) {
    const Row* R = &rows[r];
    double lo, hi; row_box(rows, R, 1, &lo, &hi);
    fr[r] = R->d > 0.0 && R->lambda > lo && R->lambda < hi;
    if (g_margin > 0.0 && fr[r] && isfinite(hi)) {   /* (laboratory) a boxed row within g_margin of its box width from a bound is held where it is */
      const double m = g_margin * (hi - lo);
      fr[r] = R->lambda > lo + m && R->lambda < hi - m;
    }
    if (!fr[r]) continue;
    for (int k = 0; k < n; ++k) { double s = 0; for (int j = k; j < n; ++j) s += R->J[j] * lc[j * n + k]; g[r][k] = s; }
    double res = -R->target; for (int j = 0; j < n; ++j) res += R->J[j] * v[j];
    w[r] = res;
    wgt[r] = g_equil ? 1.0 / R->d : 1.0;
    for (int i = 0; i < n; ++i) { h[i] -= wgt[r] * g[r][i] * res; for (int j = 0; j < n; ++j) S[i][j] += wgt[r] * g[r][i] * g[r][j]; }
  }
  for (int i = 0; i < n; ++i) tr += S[i][i];
  if (!(tr > 0.0)) return 0;                       /* no free row: nothing to solve */
#ifdef ORC_EXPERIMENTS
  if (g_dbg_on) note_free_set(rows, nr, fr);
#endif
  /* ---- a small, well-conditioned free set: the dual solve ---- */
  {
    int F[ORC_EXACT_SMALL], m = 0, small = g_small;   /* (laboratory: g_small = 0: off, 1 .. 3: sets of up to that many rows) */
    const int small_max = g_small < ORC_EXACT_SMALL ? g_small : ORC_EXACT_SMALL;
    for (int r = 0; r < nr && small; ++r) if (fr[r]) { if (m < small_max) F[m++] = r; else small = 0; }
    if (small && m > 0) {
      double A[ORC_EXACT_SMALL][ORC_EXACT_SMALL], L[ORC_EXACT_SMALL][ORC_EXACT_SMALL] = {{0}}, D[ORC_EXACT_SMALL], mu_[ORC_EXACT_SMALL];
      for (int a = 0; a < m; ++a) for (int b = 0; b <= a; ++b) {
        double t = 0; for (int k = 0; k < n; ++k) t += g[F[a]][k] * g[F[b]][k];
        A[a][b] = t;
      }
      for (int a = 0; a < m && small; ++a) {       /* L D L^T, rows in sweep order */
        double d = A[a][a];
        for (int b = 0; b < a; ++b) {
          double t = A[a][b];
          for (int c = 0; c < b; ++c) t -= L[a][c] * L[b][c] * D[c];
          L[a][b] = t / D[b];
          d -= L[a][b] * L[a][b] * D[b];
        }
        D[a] = d;
        if (!(d > ORC_EXACT_SMALL_PIVOT * A[a][a])) small = 0;   /* dependent rows: the regularised solve below */
      }
      if (small) {
        for (int a = 0; a < m; ++a) { double t = -w[F[a]]; for (int b = 0; b < a; ++b) t -= L[a][b] * mu_[b]; mu_[a] = t; }
        for (int a = 0; a < m; ++a) mu_[a] /= D[a];
        for (int a = m - 1; a >= 0; --a) for (int b = a + 1; b < m; ++b) mu_[a] -= L[b][a] * mu_[b];
        int cut = 0;
        for (int a = 0; a < m; ++a) {
          double lo, hi; row_box(rows, &rows[F[a]], 1, &lo, &hi);
          const double full = rows[F[a]].lambda + mu_[a];
          if (full < lo || full > hi) cut = 1;
        }
        double alpha = 1.0;
        if (cut)
          for (int a = 0; a < m; ++a) {
            if (mu_[a] == 0.0) continue;
            double lo, hi; row_box(rows, &rows[F[a]], 1, &lo, &hi);
            if (mu_[a] > 0.0 && !isfinite(hi)) continue;
            const double lim = ((mu_[a] > 0.0 ? hi : lo) - rows[F[a]].lambda) / mu_[a];
            if (lim < alpha) alpha = lim;
          }
        for (int a = 0; a < m; ++a) {
          Row* R = &rows[F[a]];
          double lo, hi; row_box(rows, R, 1, &lo, &hi);
          const double l = R->lambda;
          double nl = l + alpha * mu_[a];
          if (cut) {
            if (mu_[a] > 0.0 && isfinite(hi) && hi - nl <= ORC_EXACT_SNAP * (hi - l)) nl = hi;
            if (mu_[a] < 0.0 && nl - lo <= ORC_EXACT_SNAP * (l - lo)) nl = lo;
          }
          if (nl < lo) nl = lo;
          if (nl > hi) nl = hi;
          R->lambda = nl;
          for (int j = 0; j < n; ++j) v[j] += R->T[j] * (alpha * mu_[a]);
        }
        tl_last_small += 1;
        return cut;
      }
    }
  }
  const double eps = ORC_EXACT_EPS * tr;
  /* L D L^T of S + eps I (symmetric positive definite), natural order */
  double Lf[OS2R_MAX_DOF][OS2R_MAX_DOF] = {{0}}, D[OS2R_MAX_DOF];
  for (int j = 0; j < n; ++j) {
    double dj = S[j][j] + eps;
    for (int k = 0; k < j; ++k) dj -= Lf[j][k] * Lf[j][k] * D[k];
    D[j] = dj;
    for (int i = j + 1; i < n; ++i) {
      double t = S[i][j];
      for (int k = 0; k < j; ++k) t -= Lf[i][k] * Lf[j][k] * D[k];
      Lf[i][j] = t / dj;
    }
  }
  double d[OS2R_MAX_DOF] = {0}, ds[OS2R_MAX_DOF] = {0};
  const int nprox = test_consistency && g_prox_later > 0 ? g_prox_later : ORC_EXACT_PROX;
  for (int it = 0; it < nprox; ++it) {
    double z[OS2R_MAX_DOF];
    for (int i = 0; i < n; ++i) z[i] = h[i] + (it > 0 ? eps * d[i] : 0.0);
    for (int i = 0; i < n; ++i) for (int k = 0; k < i; ++k) z[i] -= Lf[i][k] * z[k];
    for (int i = 0; i < n; ++i) z[i] /= D[i];
    for (int i = n - 1; i >= 0; --i) for (int k = i + 1; k < n; ++k) z[i] -= Lf[k][i] * z[k];
    for (int i = 0; i < n; ++i) { d[i] = z[i]; ds[i] += z[i]; }
  }
  /* impulses of the free rows; does the full step take one of them out of its box? */
  double mu[3 * OS2R_MAX_DOF + OS2R_MAX_DOF];
  int cut = 0;
  double found = 0.0, left = 0.0;   /* squared residuals of the free rows before and after the step */
  for (int r = 0; r < nr; ++r) {
    if (!fr[r]) continue;
    double s = nprox * w[r];
    for (int k = 0; k < n; ++k) s += g[r][k] * ds[k];
    mu[r] = -wgt[r] * s / eps;
    double wl = w[r];
    for (int k = 0; k < n; ++k) wl += g[r][k] * d[k];
    found += w[r] * w[r]; left += wl * wl;
    double lo, hi; row_box(rows, &rows[r], 1, &lo, &hi);
    const double full = rows[r].lambda + mu[r];
    if (full < lo || full > hi) cut = 1;
  }
  /* if so, the largest feasible fraction of the step; an inconsistent free set: the step to the first bound, however long */
  /* EXPERIMENT g_incons_once: an environment takes at most that many inconsistent-set steps per physics iteration (the period-2
   * cycling of docs/studies/round4_solver.md is a second, third, ... such step undoing the first) */
  const int on = test_consistency && !cut && left > ORC_EXACT_INCONS * found && (g_incons_once <= 0 || tl_incons_steps < g_incons_once);
  double alpha = on ? INFINITY : 1.0;
  int first_hit = -1;
  if (cut || on)
    for (int r = 0; r < nr; ++r) {
      if (!fr[r] || mu[r] == 0.0) continue;
      double lo, hi; row_box(rows, &rows[r], 1, &lo, &hi);
      if (mu[r] > 0.0 && !isfinite(hi)) continue;
      const double lim = ((mu[r] > 0.0 ? hi : lo) - rows[r].lambda) / mu[r];
      if (lim < alpha) { alpha = lim; first_hit = r; }
    }
  if (on) { if (isfinite(alpha)) cut = 1; else alpha = 1.0; }
  if (g_pivot && on && cut && first_hit >= 0) {
    /* the step of an inconsistent set ends on row first_hit.  If such a step has set that row on a bound before in this
     * iteration -- the re-test sweep released it in between -- the environment is cycling (docs/studies/round5_solver.md): it keeps
     * the state the last sweep left and ends phase 2 */
    const unsigned bit = 1u << warm_slot(rows, first_hit);
    if (tl_incons_pinned & bit) return 2;
    tl_incons_pinned |= bit;
  }
  if (on && cut) tl_incons_steps += 1;
  if (g_trace > 0) {
    int nviol = 0;
    for (int r = 0; r < nr; ++r) if (fr[r]) { double lo, hi; row_box(rows, &rows[r], 1, &lo, &hi); const double full = rows[r].lambda + mu[r]; nviol += full < lo || full > hi; }
    trace_solve(rows, nr, fr, mu, alpha, cut, on, nviol, found, left);
  }
  if (g_multicut > 0 && cut && !on && alpha < pow(10.0, -g_multicut)) {
    /* EXPERIMENT (round 5): a step that a bound cuts at a ten-thousandth of its length moves nothing; the rows that cut it sit a hair
     * inside boxes that the full step would leave by hundreds of their widths (a robot at rest: sliding contacts whose friction box
     * moved by a hair, joints on the edge of sticking).  All of them -- every free row whose own bound is reached within 10 x the
     * cut fraction -- are set on their bounds at once, the velocity following row by row, and the solve repeats */
    for (int r = 0; r < nr; ++r) {
      if (!fr[r] || mu[r] == 0.0) continue;
      double lo, hi; row_box(rows, &rows[r], 1, &lo, &hi);
      if (mu[r] > 0.0 && !isfinite(hi)) continue;
      const double lim = ((mu[r] > 0.0 ? hi : lo) - rows[r].lambda) / mu[r];
      if (lim > 10.0 * alpha) continue;
      const double nl = mu[r] > 0.0 ? hi : lo, dl = nl - rows[r].lambda;
      rows[r].lambda = nl;
      for (int j = 0; j < n; ++j) v[j] += rows[r].T[j] * dl;
    }
    return 1;
  }
  if (g_repin && cut && !on) {
    /* EXPERIMENT (round 5): the re-test sweep released rows from their bounds that the very next full step sends back beyond the
     * same bound: instead of cutting the step at the first of them (a solve per row, at step lengths of 1e-5 .. 1e-2), all of them
     * are put back on their bounds at once -- the velocity following row by row, no step taken -- and the solve repeats */
    int n_back = 0;
    for (int r = 0; r < nr; ++r) {
      if (!fr[r]) continue;
      double lo, hi; row_box(rows, &rows[r], 1, &lo, &hi);
      const double full = rows[r].lambda + mu[r];
      const int back_lo = full < lo && ((tl_was_lo >> r) & 1u), back_hi = full > hi && ((tl_was_hi >> r) & 1u);
      if (!back_lo && !back_hi) continue;
      const double nl = back_lo ? lo : hi, dl = nl - rows[r].lambda;
      rows[r].lambda = nl;
      for (int j = 0; j < n; ++j) v[j] += rows[r].T[j] * dl;
      ++n_back;
    }
    if (n_back > 0) return 1;
  }
  if (g_clamp_all && cut && !on) {
    /* EXPERIMENT: a step that a bound cuts short is taken in full, every row clamped into its own box (all violators are
     * set on their bounds at once, a primal-dual active-set step), the velocity following row by row */
    for (int r = 0; r < nr; ++r) {
      if (!fr[r]) continue;
      double lo, hi; row_box(rows, &rows[r], 1, &lo, &hi);
      double nl = rows[r].lambda + mu[r];
      if (nl < lo) nl = lo;
      if (nl > hi) nl = hi;
      const double dl = nl - rows[r].lambda;
      rows[r].lambda = nl;
      for (int j = 0; j < n; ++j) v[j] += rows[r].T[j] * dl;
    }
    return 1;
  }
  /* the velocity takes the last proximal iterate (exact on the free rows), the impulses their multipliers; a row
   * that the cut step has taken to its bound (the room left is below ORC_EXACT_SNAP of what it had) is set on it */
  for (int i = 0; i < n; ++i) { double s = 0; for (int k = 0; k <= i; ++k) s += lc[i * n + k] * d[k]; v[i] += alpha * s; }
  for (int r = 0; r < nr; ++r) {
    if (!fr[r]) continue;
    double lo, hi; row_box(rows, &rows[r], 1, &lo, &hi);
    const double l = rows[r].lambda;
    double nl = l + alpha * mu[r];
    if (cut) {
      if (mu[r] > 0.0 && isfinite(hi) && hi - nl <= ORC_EXACT_SNAP * (hi - l)) nl = hi;
      if (mu[r] < 0.0 && nl - lo <= ORC_EXACT_SNAP * (l - lo)) nl = lo;
    }
    if (nl < lo) nl = lo;
    if (nl > hi) nl = hi;
    rows[r].lambda = nl;
  }
  return cut;
}

/* diagnostics: phase-2 sweeps and exact solves of the calling thread's last solve (orc_get_solver_counts) */
static _Thread_local int tl_last_sweeps = 0, tl_last_solves = 0;

/* Warm start of phase 2: the solver state of an environment.  Phase 2 of every physics iteration starts from the impulses
 * that ended the environment's previous iteration -- of this env-step or of the one before: the state is part of the
 * environment (OrcSim.warm, orc_get/set_solver_state; the reference's backend keeps one persistent constraint solver per
 * world behind gym_os2r/runtimes/gazebo_runtime.py:76,111-114) -- clamped into this iteration's box; a row without a
 * remembered impulse (a body that had no contact then; every row after a reset) keeps what phase 1 gave it.  The solution
 * moves by a thousandth per iteration, so ORC_WARM_FIRST sweeps identify the active set before the first check.  A single
 * iteration (orc_substep, orc_contact_problem) starts like the first one after a reset: nothing remembered.
 * Per thread: an environment's iterations run back to back on one thread; NAN marks "nothing remembered".
 * (orc_set_experimental_warm, studies: mode 0 -- no warm start, every iteration cold with ORC_EXACT_FIRST_COLD sweeps;
 *  mode 2 -- round 3: the first iteration of every env-step cold, the state forgotten between env-steps;
 *  `first` > 0 -- another number of sweeps before the first check of a warm iteration; < 0 -- see the function) */
#define ORC_WARM_FIRST 3
#define ORC_WARM_ROWS (3 * OS2R_MAX_DOF + OS2R_MAX_DOF)
#define ORC_WARM_SLOTS (ORC_WARM_ROWS + 1)   /* the last slot: 1.0 once an iteration has left its impulses (the joint rows are remembered) */
/* first > 0: that many sweeps before the first check; first = -k: k - 1 sweeps, then EVERY environment solves once before its first check */
static _Thread_local double tl_warm[ORC_WARM_SLOTS];
static _Thread_local int tl_prev_opened = 0;
static _Thread_local int tl_prev_solves = 0;   /* (g_solve_first) exact solves of the previous physics iteration of this env-step */
static _Thread_local signed char tl_warm_side[ORC_WARM_SLOTS];   /* (g_snap) -1 / +1: the row ended the last iteration on its lower / upper bound */
static _Thread_local int tl_cold = 0;   /* modes 0 and 2 only */
static void warm_forget(double* w) { for (int k = 0; k < ORC_WARM_SLOTS; ++k) w[k] = NAN; }
static int warm_slot(const Row* rows, int r) {
  const Row* R = &rows[r];
  if (R->kind == 0) return 3 * R->body;
  if (R->kind == 1) return 3 * R->body + (r - R->normal_row);
  return 3 * OS2R_MAX_DOF + R->body;
}

static void solve_rows(int n, Row* rows, int nr, int normal_iters, int iters, double tol, int exact, const double* minv, double* v) {
#ifdef ORC_EXPERIMENTS
  /* laboratory: ORC_DUMP_SOLVES=<file> with ORC_TRACE_SOLVES=k appends every problem that took k or more solves (its rows as they
   * come in, the unconstrained velocity, Minv) as one JSON object per line (tests/diag/r5_dump_replay.py) */
  Row rows_in[3 * OS2R_MAX_DOF + OS2R_MAX_DOF]; double v_in[OS2R_MAX_DOF];
  const int dumping = g_trace > 0 && nr <= 3 * OS2R_MAX_DOF + OS2R_MAX_DOF;
  if (dumping) { memcpy(rows_in, rows, (size_t)nr * sizeof(Row)); memcpy(v_in, v, (size_t)n * sizeof(double)); }
#endif
  int order[ORC_MAX_ROWS];
  for (int r = 0; r < nr; ++r) order[r] = r;
  if (g_row_order != 0)
    for (int r = 0; r + 2 < nr; ++r)
      if (rows[r].kind == 0 && rows[r + 1].kind == 1 && rows[r + 2].kind == 1) {
        const int y_first = g_row_order == 1 || rows[r + 2].d < rows[r + 1].d;
        order[r] = y_first ? r + 2 : r + 1; order[r + 1] = y_first ? r + 1 : r + 2; order[r + 2] = r;
        r += 2;
      }
  if (normal_iters <= 0) exact = 0;               /* the coupled pyramid has no fixed box to pivot on */
  const int cold_now = g_warm == 0 || tl_cold;
  double lc[OS2R_MAX_DOF * OS2R_MAX_DOF];
  if (exact > 0) chol_lower(n, minv, lc);
  /* EXPERIMENT (oracle only, off by default; built into the kernels in round 4, measured and backed out:
   * docs/studies/round4_solver.md): the lagged friction box.  An environment whose solver state covers every row that is
   * active now -- the joints, and every body in contact had a contact in the last iteration too -- runs no phase 1: the box
   * of a remembered contact is mu times the normal impulse that ended the environment's last iteration, and phase 2 starts
   * from the remembered impulses.  An environment with a FRESH row (a body that has just touched down, the iteration
   * after a reset) runs the normal sweeps for all its rows; its fresh contacts take their box from them. */
  const int remembers = exact > 0 && normal_iters > 0 && g_warm && !tl_cold && nr <= ORC_WARM_ROWS && g_lag_box;
  int fresh = 1;
  if (remembers) {
    fresh = isnan(tl_warm[ORC_WARM_ROWS]);
    for (int r = 0; r < nr; ++r) if (rows[r].kind == 0 && rows[r].d > 0.0 && isnan(tl_warm[warm_slot(rows, r)])) fresh = 1;
  }
  for (int phase = 0; phase < 2; ++phase) {
    const int sweeps = phase == 0 ? (fresh ? normal_iters : 0) : iters;
    if (phase == 0 && g_warm_p0 && sweeps > 0 && exact > 0 && g_warm && !tl_cold && nr <= ORC_WARM_ROWS)
      for (int r = 0; r < nr; ++r) {   /* (laboratory) phase 1 starts from the remembered normal and joint-friction impulses, not from zero */
        Row* R = &rows[r];
        if (R->kind == 1 || !(R->d > 0.0)) continue;
        const double w0 = tl_warm[warm_slot(rows, r)];
        if (isnan(w0)) continue;
        double lo, hi; row_box(rows, R, 1, &lo, &hi);
        const double nl = w0 < lo ? lo : (w0 > hi ? hi : w0), dl = nl - R->lambda;
        R->lambda = nl;
        for (int j = 0; j < n; ++j) v[j] += R->T[j] * dl;
      }
#ifdef ORC_EXPERIMENTS
    if (phase == 1 && normal_iters > 0 && g_box_probe && nr <= 64) {
      /* (laboratory) how far the normal impulses that fix the friction box are from the converged normal-only solve: 200 more
       * sweeps of phase 1 on a copy */
      double l2[64], v2[OS2R_MAX_DOF];
      for (int r = 0; r < nr; ++r) { l2[r] = rows[r].lambda; tl_boxn[r] = rows[r].lambda; }
      for (int j = 0; j < n; ++j) v2[j] = v[j];
      for (int it = 0; it < 200; ++it)
        for (int r = 0; r < nr; ++r) {
          const Row* R = &rows[r];
          if (!(R->d > 0.0) || R->kind == 1) continue;
          double res = -R->target; for (int j = 0; j < n; ++j) res += R->J[j] * v2[j];
          double lam = l2[r] - res / R->d, lo, hi;
          row_box(rows, R, 1, &lo, &hi);
          if (lam < lo) lam = lo;
          if (lam > hi) lam = hi;
          const double dl = lam - l2[r];
          l2[r] = lam;
          for (int j = 0; j < n; ++j) v2[j] += R->T[j] * dl;
        }
      double num = 0.0, den = 0.0;
      for (int r = 0; r < nr; ++r) if (rows[r].kind == 0 && rows[r].d > 0.0) { num += fabs(rows[r].lambda - l2[r]); den += l2[r]; }
      #pragma omp atomic
      g_box_stat[0] += num;
      #pragma omp atomic
      g_box_stat[1] += den;
    }
#endif
    if (phase == 1 && normal_iters > 0)
      for (int r = 0; r < nr; ++r) if (rows[r].kind == 1) {
        const int nrow = rows[r].normal_row;
        const double mem = remembers ? tl_warm[warm_slot(rows, nrow)] : NAN;
        rows[r].bound *= isnan(mem) ? rows[nrow].lambda : mem;
      }
    if (phase == 1 && exact > 0 && g_warm && !tl_cold && nr <= ORC_WARM_ROWS) {
      int fresh = 0;   /* diagnostics: an active row without a remembered impulse (a new contact, the iteration after a reset) */
      if (g_dbg_on) for (int r = 0; r < nr; ++r) if (rows[r].d > 0.0 && isnan(tl_warm[warm_slot(rows, r)])) fresh = 1;
#ifdef ORC_EXPERIMENTS
      if (g_dbg_on) {   /* (shared counters: switched on by the first orc_debug_counter call only -- sixteen threads on one cache line halve the oracle's speed) */
        __atomic_fetch_add(&g_dbg_counter[0], 1, __ATOMIC_RELAXED);
        if (fresh) __atomic_fetch_add(&g_dbg_counter[1], 1, __ATOMIC_RELAXED);
      }
#else
      (void)fresh;
#endif
    }
    if (phase == 1 && exact > 0 && g_warm && !tl_cold && nr <= ORC_WARM_ROWS)
      for (int r = 0; r < nr; ++r) {
        Row* R = &rows[r];
        const double w0 = tl_warm[warm_slot(rows, r)];
        if (!(R->d > 0.0) || isnan(w0)) continue;
        double lo, hi; row_box(rows, R, 1, &lo, &hi);
        double nl = w0 < lo ? lo : (w0 > hi ? hi : w0);
        if (g_snap && R->kind == 1 && tl_warm_side[warm_slot(rows, r)] != 0) nl = tl_warm_side[warm_slot(rows, r)] > 0 ? hi : lo;
        const double dl = nl - R->lambda;
        R->lambda = nl;
        for (int j = 0; j < n; ++j) v[j] += R->T[j] * dl;
      }
    int solves = 0, rounds = 0, solves_at_measure = 0, incons_at_measure = 0;
    if (phase == 1) { tl_incons_steps = 0; tl_incons_pinned = 0u; }
    int cycling = 0;
    double e_prev = -1.0;
    if (phase == 1) { tl_last_sweeps = 0; tl_last_solves = 0; tl_last_small = 0; tl_trace_len = 0; }
#ifdef ORC_EXPERIMENTS
    if (g_trace < 0) { const char* t = getenv("ORC_TRACE_SOLVES"); g_trace = t ? atoi(t) : 0; }
#endif
    /* (laboratory) an iteration opens with a solve when its predecessor took >= g_solve_first solves, and goes on doing so while that takes exactly one */
    const int first_now = (g_solve_first && phase == 1 && exact > 0 && (tl_prev_solves >= g_solve_first || (tl_prev_opened && tl_prev_solves == 1))) ? 0 : ORC_EXACT_FIRST(n);
    if (phase == 1) tl_prev_opened = first_now == 0 && g_solve_first;
    for (int it = 0; it < sweeps; ++it) {
      /* exact finish: from the check after the first ORC_EXACT_FIRST sweeps on, solves (repeated while a bound cuts
       * the step short) precede every sweep until the budget `exact` is spent */
      if (phase == 1 && exact > 0 && it >= first_now && solves < exact) {
        int blocked = 1;
        while (blocked && solves < exact) {
#ifdef ORC_EXPERIMENTS
          tl_solve_index = solves;
#endif
          blocked = exact_step(n, rows, nr, lc, v, solves > 0); ++solves; if (g_sweep_after_cut) break;
          if (blocked == 2) { cycling = 1; break; }
        }
        if (cycling) { tl_last_solves = solves; break; }
        tl_last_solves = solves;
        ++rounds;
      }
      if (phase == 1) tl_last_sweeps = it + 1;
      if (phase == 1 && g_repin) {
        tl_was_lo = tl_was_hi = 0u;
        for (int r = 0; r < nr && r < 32; ++r) {
          double lo, hi; row_box(rows, &rows[r], 1, &lo, &hi);
          if (rows[r].d > 0.0 && rows[r].lambda <= lo) tl_was_lo |= 1u << r;
          if (rows[r].d > 0.0 && rows[r].lambda >= hi) tl_was_hi |= 1u << r;
        }
      }
      double moved = 0.0;
      for (int ri = 0; ri < nr; ++ri) {
        const int r = (phase == 1 && !g_block_solve) ? order[ri] : ri;
        Row* R = &rows[r];
        if (g_block_solve && (g_block_solve == 1 || it < g_block_solve - 1) && phase == 1 && normal_iters > 0 && R->kind == 0 && r + 2 < nr && rows[r + 1].kind == 1 &&
            rows[r + 2].kind == 1 && (g_block_kind >= 2 ? block_update_pair(n, rows, r, v, &moved) : g_block_kind ? block_update_cheap(n, rows, r, v, &moved) : block_update(n, rows, r, v, &moved))) { ri += 2; continue; }
        if (!(R->d > 0.0)) continue;
        if (phase == 0 && R->kind == 1) continue;
        double res = -R->target; for (int j = 0; j < n; ++j) res += R->J[j] * v[j];
        double lam = R->lambda - res / R->d, lo, hi;
        row_box(rows, R, normal_iters > 0, &lo, &hi);
        if (lam < lo) lam = lo;
        if (lam > hi) lam = hi;
        double dl = lam - R->lambda;
        R->lambda = lam;
        moved += fabs(res) * fabs(dl);
        for (int j = 0; j < n; ++j) v[j] += R->T[j] * dl;
      }
      if (phase == 1) trace_sweep(moved);
      /* EXPERIMENT: a round (solves + re-test sweep) that does not bring the measure below g_stall of the last one ends phase 2 */
      if (phase == 1 && exact > 0 && g_stall > 0.0 && solves > 0 && it + 1 > first_now) {
        if (solves > solves_at_measure && e_prev >= 0.0 && moved > g_stall * e_prev && (!g_stall_incons_only || tl_incons_steps > incons_at_measure)) { break; }
      }
      if (phase == 1) { e_prev = moved; solves_at_measure = solves; incons_at_measure = tl_incons_steps; }
      if (phase == 1 && exact > 0 && g_max_rounds > 0 && (rounds >= g_max_rounds || (g_stop_at_cap && solves >= exact))) break;   /* experiment: bounded rounds */
      if (phase == 1 && it + 1 < sweeps && moved <= tol) {
        if (exact > 0 ? it + 1 >= first_now && !(g_solve_always && solves == 0) : (it + 1) % ORC_PGS_GROUP == 0) break;
      }
    }
    if (phase == 1 && g_trace > 0 && solves >= g_trace) { tl_trace[tl_trace_len] = 0; fprintf(stderr, "solves %d sweeps %d:%s\n", solves, tl_last_sweeps, tl_trace); }
#ifdef ORC_EXPERIMENTS
    if (phase == 1 && dumping && solves >= g_trace && getenv("ORC_DUMP_SOLVES")) {
      char buf[16384]; int o = 0;
      o += snprintf(buf + o, sizeof(buf) - o, "{\"n\": %d, \"nr\": %d, \"solves\": %d, \"vstar\": [", n, nr, solves);
      for (int i = 0; i < n; ++i) o += snprintf(buf + o, sizeof(buf) - o, "%s%.17g", i ? ", " : "", v_in[i]);
      o += snprintf(buf + o, sizeof(buf) - o, "], \"minv\": [");
      for (int i = 0; i < n * n; ++i) o += snprintf(buf + o, sizeof(buf) - o, "%s%.17g", i ? ", " : "", minv[i]);
      o += snprintf(buf + o, sizeof(buf) - o, "], \"rows\": [");
      for (int r = 0; r < nr; ++r) {
        o += snprintf(buf + o, sizeof(buf) - o, "%s{\"kind\": %d, \"body\": %d, \"normal_row\": %d, \"target\": %.17g, \"bound\": %.17g, \"J\": [", r ? ", " : "",
                      rows_in[r].kind, rows_in[r].body, rows_in[r].normal_row, rows_in[r].target, rows_in[r].bound);
        for (int i = 0; i < n; ++i) o += snprintf(buf + o, sizeof(buf) - o, "%s%.17g", i ? ", " : "", rows_in[r].J[i]);
        o += snprintf(buf + o, sizeof(buf) - o, "]}");
      }
      o += snprintf(buf + o, sizeof(buf) - o, "]}\n");
      FILE* f = fopen(getenv("ORC_DUMP_SOLVES"), "a");
      if (f) { fwrite(buf, 1, (size_t)o, f); fclose(f); }
    }
#endif
#ifdef ORC_EXPERIMENTS
    if (phase == 1 && normal_iters > 0 && g_box_probe && nr <= 64) {
      double num = 0.0, den = 0.0;
      for (int r = 0; r < nr; ++r) if (rows[r].kind == 0 && rows[r].d > 0.0) { num += fabs(tl_boxn[r] - rows[r].lambda); den += rows[r].lambda; }
      #pragma omp atomic
      g_box_stat[2] += num;
      #pragma omp atomic
      g_box_stat[3] += den;
    }
#endif
    if (phase == 1) tl_prev_solves = solves;
    if (phase == 1 && exact > 0 && g_warm && nr <= ORC_WARM_ROWS) {
      warm_forget(tl_warm);
      for (int r = 0; r < nr; ++r) if (rows[r].d > 0.0) tl_warm[warm_slot(rows, r)] = rows[r].lambda;
      for (int k = 0; k < ORC_WARM_SLOTS; ++k) tl_warm_side[k] = 0;
      if (g_snap) for (int r = 0; r < nr; ++r) if (rows[r].d > 0.0 && rows[r].kind == 1) {
        double lo, hi; row_box(rows, &rows[r], 1, &lo, &hi);
        tl_warm_side[warm_slot(rows, r)] = rows[r].lambda >= hi ? 1 : (rows[r].lambda <= lo ? -1 : 0);
      }
      tl_warm[ORC_WARM_ROWS] = (g_solve_first && tl_prev_opened && solves == 1) ? 2.0 : 1.0;   /* (2: laboratory, the solve-first mode carried to the next env-step) */
      tl_cold = 0;
    }
  }
}

static void substep_model(const Os2rConfig* cfg, int contact_model, const EnvParams* ep, double* q, double* qd, const double tau2[2]) {
  const int n = cfg->model.nq;
  double v[OS2R_MAX_DOF], minv[OS2R_MAX_DOF * OS2R_MAX_DOF];
  if (contact_model == ORC_CONTACT_CENTROID) {
    Row rows[3 * OS2R_MAX_DOF + OS2R_MAX_DOF];
    const int nr = build_problem(cfg, contact_model, ep, q, qd, tau2, v, minv, rows);
    solve_rows(n, rows, nr, cfg->pgs_normal_iters, cfg->pgs_iters, cfg->pgs_tol, cfg->pgs_exact, minv, v);
  } else {
    Row* rows = (Row*)malloc(sizeof(Row) * ORC_MAX_ROWS);
    const int nr = build_problem(cfg, contact_model, ep, q, qd, tau2, v, minv, rows);
    solve_rows(n, rows, nr, 0, cfg->pgs_iters, cfg->pgs_tol, 0, minv, v);
    free(rows);
  }
  for (int i = 0; i < n; ++i) { qd[i] = v[i]; q[i] += cfg->dt * v[i]; }
}

static void substep(const Os2rConfig* cfg, const EnvParams* ep, double* q, double* qd, const double tau2[2]) {
  warm_forget(tl_warm); tl_cold = g_warm != 1;   /* a single iteration: nothing remembered */
  substep_model(cfg, ORC_CONTACT_CENTROID, ep, q, qd, tau2);
}

static void params_from(const Os2rConfig* cfg, const double* mass_scale, const double* damping, const double* friction,
                        const double* mu, double gravity_z, EnvParams* ep) {
  nominal_params(&cfg->model, ep);
  int n = cfg->model.nq;
  if (mass_scale) memcpy(ep->mass_scale, mass_scale, n * sizeof(double));
  if (damping) memcpy(ep->damping, damping, n * sizeof(double));
  if (friction) memcpy(ep->friction, friction, n * sizeof(double));
  if (mu) memcpy(ep->mu, mu, n * sizeof(double));
  ep->gravity_z = gravity_z;
}

void orc_substep(const Os2rConfig* cfg, const double* mass_scale, const double* damping, const double* friction,
                 const double* mu, double gravity_z, double* q, double* qd, const double tau2[2]) {
  EnvParams ep; params_from(cfg, mass_scale, damping, friction, mu, gravity_z, &ep);
  substep(cfg, &ep, q, qd, tau2);
}

void orc_substep_model(const Os2rConfig* cfg, int contact_model, const double* mass_scale, const double* damping,
                       const double* friction, const double* mu, double gravity_z, double* q, double* qd,
                       const double tau2[2]) {
  EnvParams ep; params_from(cfg, mass_scale, damping, friction, mu, gravity_z, &ep);
  warm_forget(tl_warm); tl_cold = g_warm != 1;
  substep_model(cfg, contact_model, &ep, q, qd, tau2);
}

/* The boxed LCP of one physics iteration, laid open for independent checks (tests/test_oracle_contact.py):
 * builds the rows for the state (q, qd), solves them with the configuration's sweep counts and returns
 * the problem AND the solution.  Arrays are sized for max_rows rows; returns the number of rows (or -1
 * if max_rows is too small).  vstar: unconstrained velocity; v_out: velocity after the solve; bound: mu
 * (tangential) or friction*dt (joint) as built; box: the half-width actually used for the row in the
 * last sweep; lambda: impulses after the solve; point: world contact point of the row. */
int orc_contact_problem(const Os2rConfig* cfg, int contact_model, const double* mass_scale, const double* damping,
                        const double* friction, const double* mu, double gravity_z, const double* q,
                        const double* qd, const double tau2[2], int max_rows, double* vstar, double* minv,
                        double* J, double* target, int32_t* kind, int32_t* normal_row, int32_t* body,
                        double* bound, double* box, double* lambda, double* point, double* v_out) {
  EnvParams ep; params_from(cfg, mass_scale, damping, friction, mu, gravity_z, &ep);
  const int n = cfg->model.nq;
  Row* rows = (Row*)malloc(sizeof(Row) * ORC_MAX_ROWS);
  double v[OS2R_MAX_DOF];
  const int nr = build_problem(cfg, contact_model, &ep, q, qd, tau2, v, minv, rows);
  if (nr > max_rows) { free(rows); return -1; }
  for (int i = 0; i < n; ++i) vstar[i] = v[i];
  for (int r = 0; r < nr; ++r) {
    for (int j = 0; j < n; ++j) J[r * n + j] = rows[r].J[j];
    target[r] = rows[r].target; kind[r] = rows[r].kind; normal_row[r] = rows[r].normal_row; body[r] = rows[r].body;
    bound[r] = rows[r].bound;
    for (int i = 0; i < 3; ++i) point[3 * r + i] = rows[r].point[i];
  }
  const int coupled = contact_model != ORC_CONTACT_CENTROID || cfg->pgs_normal_iters == 0;
  warm_forget(tl_warm); tl_cold = g_warm != 1;
  solve_rows(n, rows, nr, coupled ? 0 : cfg->pgs_normal_iters, cfg->pgs_iters, cfg->pgs_tol, coupled ? 0 : cfg->pgs_exact, minv, v);
  for (int r = 0; r < nr; ++r) {
    lambda[r] = rows[r].lambda;
    if (rows[r].kind == 0) box[r] = INFINITY;
    else if (rows[r].kind == 1) box[r] = coupled ? rows[r].bound * rows[rows[r].normal_row].lambda : rows[r].bound;
    else box[r] = rows[r].bound;
  }
  for (int i = 0; i < n; ++i) v_out[i] = v[i];
  free(rows);
  return nr;
}

/* ------------------------------------------------------------------------- *
 * Batched simulator with the same semantics as the C-ABI in include/os2r.h.
 * Host arrays, SoA [count][N] like the device layout.
 * ------------------------------------------------------------------------- */
struct OrcSim {
  Os2rConfig cfg;
  int64_t N;
  double *q, *qd;          /* [nq][N] */
  double *hist;            /* [2][2][N]: hist[which][j][e] */
  double *mass_scale, *damping, *friction, *mu, *gravity; /* [nq][N] ..., [N] */
  int32_t* steps; uint32_t* episode; uint8_t* pose;
  uint64_t step_count;
  int nthreads;
  int contact_model;   /* ORC_CONTACT_*: the specification unless a test asks for the comparison model */
  double* warm;            /* [N][ORC_WARM_SLOTS]: the solver state (impulses that ended the last physics iteration; NAN: none) */
  int8_t* solver_counts; /* diagnostics, [2][substeps][N]: phase-2 sweeps / exact solves of every physics iteration of the last step */
};

static void load_params(const OrcSim* s, int64_t e, EnvParams* ep) {
  int n = s->cfg.model.nq; int64_t N = s->N;
  for (int i = 0; i < n; ++i) {
    ep->mass_scale[i] = s->mass_scale[i * N + e]; ep->damping[i] = s->damping[i * N + e];
    ep->friction[i] = s->friction[i * N + e]; ep->mu[i] = s->mu[i * N + e];
  }
  ep->gravity_z = s->gravity[e];
}

/* randomizers/monopod_no_rand.py:59-98 and randomizers/monopod.py:89-128,182-215 */
static void reset_env(OrcSim* s, int64_t e) {
  const Os2rConfig* cfg = &s->cfg; const Os2rTaskSpec* ts = &cfg->task;
  const int n = cfg->model.nq; const int64_t N = s->N;
  const uint32_t genv = (uint32_t)(cfg->env_offset + e);
  const uint32_t epi = s->episode[e];
  double u[2], qn[OS2R_MAX_DOF] = {0};
  orc_uniform2(cfg->seed, genv, STREAM_RESET, epi, 0, u);
  int pi = (int)(u[0] * ts->n_reset_poses);
  if (pi >= ts->n_reset_poses) pi = ts->n_reset_poses - 1;
  double pitch = ts->reset_pitch[pi], hip, knee, yaw = 0.0;
  if (ts->reset_mode == OS2R_RESET_FIXED) {
    if (ts->reset_simple) {
      /* observation_space.sample() of the two joint angles: U(-1,1) (monopod_no_rand.py:84) */
      double w[2]; orc_uniform2(cfg->seed, genv, STREAM_RESET, epi, 1, w);
      hip = 2.0 * w[0] - 1.0; knee = 2.0 * w[1] - 1.0;
    } else { hip = ts->reset_hip[pi]; knee = ts->reset_knee[pi]; }
  } else {
    pitch *= 0.8 + 0.4 * u[1];                                   /* monopod.py:94 */
    double z[2], w[2], w2[2];
    normal2(cfg->seed, genv, STREAM_RESET, epi, 1, z);
    orc_uniform2(cfg->seed, genv, STREAM_RESET, epi, 2, w);
    orc_uniform2(cfg->seed, genv, STREAM_RESET, epi, 3, w2);
    double r0 = fabs(0.2 * z[0]), r1 = fabs(0.2 * z[1]);
    double rmax = r0 > r1 ? r0 : r1, rmin = r0 > r1 ? r1 : r0;
    if (!ts->reset_laying[pi]) { double a[2]; orc_leg_joint_angles(ts->leg_def, pitch, a); hip = a[0]; knee = a[1]; }
    else { hip = 1.57 - (w[0] < 0.5 ? 3.14 : 0.0); knee = 0.0; }      /* :106 */
    /* `(x>0 - x<0)` is the chained comparison x>0 (monopod.py:102-103,109-110) */
    hip = hip + (hip > 0.0 ? 1.0 : 0.0) * rmax;
    knee = knee - (knee > 0.0 ? 1.0 : 0.0) * rmin;
    double dir = 1.0 - (w[1] < 0.5 ? 2.0 : 0.0);                  /* :111 */
    hip *= dir; knee *= dir;
    yaw = -0.2 + 0.4 * w2[0];                                     /* :113 */
  }
  if (ts->dof_pitch >= 0) qn[ts->dof_pitch] = pitch;
  if (ts->dof_yaw >= 0) qn[ts->dof_yaw] = yaw;
  if (ts->dof_hip >= 0) qn[ts->dof_hip] = hip;
  if (ts->dof_knee >= 0) qn[ts->dof_knee] = knee;
  for (int i = 0; i < n; ++i) { s->q[i * N + e] = qn[i]; s->qd[i * N + e] = 0.0; }
  s->pose[e] = (uint8_t)ts->reset_pose_id[pi];
  s->steps[e] = 0;
  warm_forget(s->warm + (size_t)e * ORC_WARM_SLOTS);   /* a new episode: the solver remembers nothing */
  if (ts->reset_mode == OS2R_RESET_RANDOM && ts->randomize_params) {
    const Os2rModel* md = &cfg->model;
    for (int i = 0; i < n; ++i) {
      double a[2], b[2];
      orc_uniform2(cfg->seed, genv, STREAM_PARAMS, epi, 2 * i, a);
      orc_uniform2(cfg->seed, genv, STREAM_PARAMS, epi, 2 * i + 1, b);
      s->mass_scale[i * N + e] = ts->dr_mass_lo + (ts->dr_mass_hi - ts->dr_mass_lo) * a[0];
      s->friction[i * N + e] = ts->dr_friction_lo + (ts->dr_friction_hi - ts->dr_friction_lo) * a[1];
      s->damping[i * N + e] = md->damping[i] * (ts->dr_damping_lo + (ts->dr_damping_hi - ts->dr_damping_lo) * b[0]);
      s->mu[i * N + e] = ts->dr_mu_base * (ts->dr_mu_lo + (ts->dr_mu_hi - ts->dr_mu_lo) * b[1]);
    }
  }
  /* randomize_physics runs when the reference re-creates its simulator, every num_physics_rollouts rollouts
   * (randomizers/monopod.py:36-41,56-61,371): `epi` rollouts of this environment are over now */
  if (ts->reset_mode == OS2R_RESET_RANDOM && ts->gravity_rollouts > 0 && epi > 0u && epi % (uint32_t)ts->gravity_rollouts == 0u) {
    double z[2]; normal2(cfg->seed, genv, STREAM_GRAVITY, epi / (uint32_t)ts->gravity_rollouts, 0, z);
    s->gravity[e] = ts->dr_gravity_mean + ts->dr_gravity_std * z[0];
  }
  s->episode[e] = epi + 1;
}

int orc_create(const Os2rConfig* cfg, OrcSim** out) {
  if (!cfg || !out || cfg->abi_version != OS2R_ABI_VERSION || cfg->num_envs <= 0) return OS2R_ERR_INVALID;
  OrcSim* s = (OrcSim*)calloc(1, sizeof(OrcSim));
  s->cfg = *cfg; s->N = cfg->num_envs; s->nthreads = 1;
  const int n = cfg->model.nq; const int64_t N = s->N;
  s->q = calloc(n * N, 8); s->qd = calloc(n * N, 8); s->hist = calloc(4 * N, 8);
  s->mass_scale = calloc(n * N, 8); s->damping = calloc(n * N, 8); s->friction = calloc(n * N, 8);
  s->mu = calloc(n * N, 8); s->gravity = calloc(N, 8);
  s->steps = calloc(N, 4); s->episode = calloc(N, 4); s->pose = calloc(N, 1);
  s->warm = malloc((size_t)N * ORC_WARM_SLOTS * 8);
  for (int64_t e = 0; e < N; ++e) {
    for (int i = 0; i < n; ++i) {
      s->mass_scale[i * N + e] = 1.0; s->damping[i * N + e] = cfg->model.damping[i];
      s->friction[i * N + e] = cfg->model.friction[i]; s->mu[i * N + e] = cfg->model.mu[i];
    }
    s->gravity[e] = cfg->model.gravity_z;
    if (cfg->task.reset_mode == OS2R_RESET_RANDOM && cfg->task.dr_gravity_std > 0.0) {
      double z[2]; normal2(cfg->seed, (uint32_t)(cfg->env_offset + e), STREAM_GRAVITY, 0, 0, z);
      s->gravity[e] = cfg->task.dr_gravity_mean + cfg->task.dr_gravity_std * z[0];   /* monopod.py:58 */
    }
    reset_env(s, e);
  }
  *out = s;
  return OS2R_OK;
}

void orc_destroy(OrcSim* s) {
  if (!s) return;
  free(s->q); free(s->qd); free(s->hist); free(s->mass_scale); free(s->damping); free(s->friction);
  free(s->mu); free(s->gravity); free(s->steps); free(s->episode); free(s->pose); free(s->warm); free(s->solver_counts); free(s);
}

void orc_set_threads(OrcSim* s, int n) { s->nthreads = n < 1 ? 1 : n; }
void orc_set_contact_model(OrcSim* s, int model) { s->contact_model = model; }
/* diagnostics: record, from now on, the phase-2 sweeps and exact solves of every physics iteration of a step */
int orc_get_solver_counts(OrcSim* s, int8_t* sweeps, int8_t* solves) {
  const size_t n = (size_t)s->cfg.substeps * s->N;
  if (!s->solver_counts) { s->solver_counts = (int8_t*)calloc(3 * n, 1); return 1; }   /* first call switches recording on */
  if (sweeps) memcpy(sweeps, s->solver_counts, n);
  if (solves) memcpy(solves, s->solver_counts + n, n);
  return 0;
}
/* (recording on:) how many of those solves were dual solves of a small free set */
int orc_get_small_solve_counts(OrcSim* s, int8_t* small) {
  const size_t n = (size_t)s->cfg.substeps * s->N;
  if (!s->solver_counts) return 1;
  memcpy(small, s->solver_counts + 2 * n, n);
  return 0;
}

static void observe_env(const OrcSim* s, int64_t e, double* obs) {
  const int n = s->cfg.model.nq; const int64_t N = s->N;
  double q[OS2R_MAX_DOF], qd[OS2R_MAX_DOF], h1[2];
  for (int i = 0; i < n; ++i) { q[i] = s->q[i * N + e]; qd[i] = s->qd[i * N + e]; }
  h1[0] = s->hist[(2 + 0) * N + e]; h1[1] = s->hist[(2 + 1) * N + e];
  orc_observe(&s->cfg.task, q, qd, h1, obs);
}

int orc_reset(OrcSim* s, const uint8_t* mask, double* obs) {
  const int D = s->cfg.task.obs_dim;
  for (int64_t e = 0; e < s->N; ++e) {
    if (!mask || mask[e]) reset_env(s, e);
    if (obs) observe_env(s, e, obs + e * D);
  }
  return OS2R_OK;
}

int orc_step(OrcSim* s, const double* actions, double* obs, double* reward, uint8_t* done, double* term_obs) {
  const Os2rConfig* cfg = &s->cfg; const Os2rTaskSpec* ts = &cfg->task;
  const int n = cfg->model.nq, D = ts->obs_dim; const int64_t N = s->N;
  const uint64_t t = s->step_count;
#ifdef _OPENMP
#pragma omp parallel for num_threads(s->nthreads) schedule(static)
#endif
  for (int64_t e = 0; e < N; ++e) {
    double a[2];
    if (actions) { a[0] = actions[2 * e]; a[1] = actions[2 * e + 1]; }
    else {
      double u[2]; orc_uniform2(cfg->seed, (uint32_t)(cfg->env_offset + e), STREAM_ACTION, (uint32_t)t, (uint32_t)(t >> 32), u);
      a[0] = 2.0 * u[0] - 1.0; a[1] = 2.0 * u[1] - 1.0;
    }
    for (int j = 0; j < 2; ++j) { if (a[j] < -1.0) a[j] = -1.0; if (a[j] > 1.0) a[j] = 1.0; }
    double tau[2] = {cfg->model.max_torque[0] * a[0], cfg->model.max_torque[1] * a[1]};  /* monopod.py:223 */
    double q[OS2R_MAX_DOF], qd[OS2R_MAX_DOF];
    EnvParams ep; load_params(s, e, &ep);
    for (int i = 0; i < n; ++i) { q[i] = s->q[i * N + e]; qd[i] = s->qd[i * N + e]; }
    tl_prev_solves = 0; tl_prev_opened = 0;
    if (g_warm == 1) { memcpy(tl_warm, s->warm + (size_t)e * ORC_WARM_SLOTS, sizeof tl_warm); tl_cold = 0; if (g_solve_first && tl_warm[ORC_WARM_ROWS] == 2.0) { tl_prev_opened = 1; tl_prev_solves = 1; } }
    else { warm_forget(tl_warm); tl_cold = 1; }
    for (int k = 0; k < cfg->substeps; ++k) {                                                          /* gazebo_runtime.py:70-77 */
      substep_model(cfg, s->contact_model, &ep, q, qd, tau);
      if (s->solver_counts) {
        s->solver_counts[(size_t)k * s->N + e] = (int8_t)(tl_last_sweeps > 127 ? 127 : tl_last_sweeps);
        s->solver_counts[((size_t)cfg->substeps + k) * s->N + e] = (int8_t)(tl_last_solves > 127 ? 127 : tl_last_solves);
        s->solver_counts[((size_t)2 * cfg->substeps + k) * s->N + e] = (int8_t)(tl_last_small > 127 ? 127 : tl_last_small);
      }
    }
    if (g_warm == 1) memcpy(s->warm + (size_t)e * ORC_WARM_SLOTS, tl_warm, sizeof tl_warm);
    int bad = 0;
    for (int i = 0; i < n; ++i) { if (!isfinite(q[i]) || !isfinite(qd[i])) bad = 1; s->q[i * N + e] = q[i]; s->qd[i * N + e] = qd[i]; }
    /* action_history.appendleft (monopod.py:232-235) */
    double h1[2] = {s->hist[0 * N + e], s->hist[1 * N + e]};
    s->hist[2 * N + e] = h1[0]; s->hist[3 * N + e] = h1[1];
    /* the history stores the read-back force target divided by max_torque: (2.5*a)/2.5 */
    double as[2] = {tau[0] / cfg->model.max_torque[0], tau[1] / cfg->model.max_torque[1]};
    s->hist[0 * N + e] = as[0]; s->hist[1 * N + e] = as[1];
    double ob[OS2R_MAX_OBS];
    orc_observe(ts, q, qd, h1, ob);
    double rew = orc_reward(ts, ob, as, h1);
    int dn = orc_done(ts, ob);
    s->steps[e] += 1;
    int trunc = ts->max_episode_steps > 0 && s->steps[e] >= ts->max_episode_steps;
    uint8_t flag = (uint8_t)((dn ? 1 : 0) | (trunc ? 2 : 0) | (bad ? 4 : 0));
    if (term_obs) memcpy(term_obs + e * D, ob, D * sizeof(double));
    if (flag && cfg->auto_reset) { reset_env(s, e); observe_env(s, e, ob); }    /* subproc_vec_env.py:17-20 */
    if (obs) memcpy(obs + e * D, ob, D * sizeof(double));
    if (reward) reward[e] = rew;
    if (done) done[e] = flag;
  }
  s->step_count = t + 1;
  return OS2R_OK;
}

int orc_get_state(OrcSim* s, double* q, double* qd) {
  size_t b = (size_t)s->cfg.model.nq * s->N * 8;
  if (q) memcpy(q, s->q, b);
  if (qd) memcpy(qd, s->qd, b);
  return OS2R_OK;
}
int orc_set_state(OrcSim* s, const double* q, const double* qd) {
  size_t b = (size_t)s->cfg.model.nq * s->N * 8;
  if (q) memcpy(s->q, q, b);
  if (qd) memcpy(s->qd, qd, b);
  for (int64_t e = 0; e < s->N; ++e) warm_forget(s->warm + (size_t)e * ORC_WARM_SLOTS);   /* as os2r_set_state: starts like a reset */
  return OS2R_OK;
}
/* The solver state in the layout of include/os2r.h (os2r_get_solver_state): lam [4*nq][N] -- rows b, nq + b, 2nq + b: normal
 * and the two tangential impulses of body b's contact, row 3nq + j: friction impulse of joint j --, flags [N]: bit b: body b
 * had a contact in the environment's last physics iteration, bit 31: an iteration has run since the reset. */
int orc_get_solver_state(OrcSim* s, double* lam, uint32_t* flags) {
  const int n = s->cfg.model.nq; const int64_t N = s->N;
  for (int64_t e = 0; e < N; ++e) {
    const double* w = s->warm + (size_t)e * ORC_WARM_SLOTS;
    uint32_t f = isnan(w[ORC_WARM_ROWS]) ? 0u : 0x80000000u;
    for (int b = 0; b < n; ++b) {
      if (!isnan(w[3 * b])) f |= 1u << b;
      if (lam) for (int t = 0; t < 3; ++t) lam[(size_t)(t * n + b) * N + e] = isnan(w[3 * b + t]) ? 0.0 : w[3 * b + t];
      if (lam) lam[(size_t)(3 * n + b) * N + e] = isnan(w[3 * OS2R_MAX_DOF + b]) ? 0.0 : w[3 * OS2R_MAX_DOF + b];
    }
    if (flags) flags[e] = f;
  }
  return OS2R_OK;
}
int orc_set_solver_state(OrcSim* s, const double* lam, const uint32_t* flags) {
  const int n = s->cfg.model.nq; const int64_t N = s->N;
  if (!lam || !flags) return OS2R_ERR_INVALID;
  for (int64_t e = 0; e < N; ++e) {
    double* w = s->warm + (size_t)e * ORC_WARM_SLOTS;
    warm_forget(w);
    const uint32_t f = flags[e];
    if (f & 0x80000000u) w[ORC_WARM_ROWS] = 1.0;
    for (int b = 0; b < n; ++b) {
      if ((f >> b) & 1u) for (int t = 0; t < 3; ++t) w[3 * b + t] = lam[(size_t)(t * n + b) * N + e];
      if (f & 0x80000000u) w[3 * OS2R_MAX_DOF + b] = lam[(size_t)(3 * n + b) * N + e];
    }
  }
  return OS2R_OK;
}
int orc_get_action_history(OrcSim* s, int which, double* out) { memcpy(out, s->hist + (size_t)which * 2 * s->N, 2 * s->N * 8); return OS2R_OK; }
int orc_set_action_history(OrcSim* s, int which, const double* in) { memcpy(s->hist + (size_t)which * 2 * s->N, in, 2 * s->N * 8); return OS2R_OK; }

static double* param_ptr(OrcSim* s, int field, int* count) {
  int n = s->cfg.model.nq;
  switch (field) {
    case OS2R_PARAM_MASS_SCALE: *count = n; return s->mass_scale;
    case OS2R_PARAM_DAMPING: *count = n; return s->damping;
    case OS2R_PARAM_FRICTION: *count = n; return s->friction;
    case OS2R_PARAM_MU: *count = n; return s->mu;
    case OS2R_PARAM_GRAVITY: *count = 1; return s->gravity;
    default: *count = 0; return NULL;
  }
}
int orc_set_params(OrcSim* s, int field, const double* src) {
  int c; double* p = param_ptr(s, field, &c); if (!p) return OS2R_ERR_INVALID;
  memcpy(p, src, (size_t)c * s->N * 8); return OS2R_OK;
}
int orc_get_params(OrcSim* s, int field, double* dst) {
  int c; double* p = param_ptr(s, field, &c); if (!p) return OS2R_ERR_INVALID;
  memcpy(dst, p, (size_t)c * s->N * 8); return OS2R_OK;
}
int orc_get_episode_info(OrcSim* s, int32_t* steps, uint32_t* episode, uint8_t* pose) {
  if (steps) memcpy(steps, s->steps, s->N * 4);
  if (episode) memcpy(episode, s->episode, s->N * 4);
  if (pose) memcpy(pose, s->pose, s->N);
  return OS2R_OK;
}
int orc_set_episode_info(OrcSim* s, const int32_t* steps, const uint32_t* episode, const uint8_t* pose) {
  if (steps) memcpy(s->steps, steps, s->N * 4);
  if (episode) memcpy(s->episode, episode, s->N * 4);
  if (pose) memcpy(s->pose, pose, s->N);
  return OS2R_OK;
}
uint64_t orc_get_step_count(OrcSim* s) { return s->step_count; }
void orc_set_step_count(OrcSim* s, uint64_t v) { s->step_count = v; }
