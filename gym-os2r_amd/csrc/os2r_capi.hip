// os2r_capi.hip — implementation of the C-ABI declared in include/os2r.h.
//
// Host-side only: owns the device buffers of one simulator handle, converts the
// host config into the uniform device structs, and launches the kernels of
// os2r_kernels.hpp on the caller's stream.  Nothing here computes physics on the
// CPU; a missing GPU is an error (OS2R_ERR_NO_DEVICE), never a fallback.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <deque>
#include <mutex>
#include <string>
#include <vector>

#include "os2r_kernels.hpp"

using namespace os2r;

namespace os2r {
int static_model_id(const Os2rModel& m);
bool same_model(const Os2rModel& a, const Os2rModel& b);
}

// model-specialised code objects registered by the host binding (include/os2r.h)
struct JitEntry {
  Os2rModel model;
  int dtype = 0, device = 0;
  hipModule_t module = nullptr;
  hipFunction_t fn[2][2][2] = {};   // [contact][per-env parameters][default sweep counts compiled in]
  // optional: contact + default sweep counts + the observation layout below folded in (os2r_jit_step_c1_d*_l)
  hipFunction_t fn_layout[2] = {};
  unsigned long long layout_kinds = 0, layout_srcs = 0;
  int layout_dim = -1;
};

static void task_layout(const Os2rTaskSpec& t, unsigned long long& kinds, unsigned long long& srcs, int& dim) {
  kinds = 0; srcs = 0; dim = t.obs_dim;
  for (int d = 0; d < t.obs_dim && d < OS2R_MAX_OBS; ++d) {
    kinds |= (unsigned long long)(t.obs_kind[d] & 15) << (4 * d);
    srcs |= (unsigned long long)(t.obs_src[d] & 15) << (4 * d);
  }
}

static std::mutex g_jit_mutex;
static std::deque<JitEntry> g_jit;   // entries are never removed: handles keep pointers into it

// newest registration of this robot that exports kernels for the handle's contact flag (a robot may have been
// registered once with and once without ground contact: two code objects)
static const JitEntry* find_jit(const Os2rModel& m, int dtype, int device, bool contact, const Os2rTaskSpec& task) {
  unsigned long long kinds, srcs;
  int dim;
  task_layout(task, kinds, srcs, dim);
  std::lock_guard<std::mutex> lock(g_jit_mutex);
  const JitEntry* any = nullptr;
  for (auto it = g_jit.rbegin(); it != g_jit.rend(); ++it)
    if (it->dtype == dtype && it->device == device && (it->fn[contact][0][0] || it->fn[contact][1][0]) &&
        os2r::same_model(it->model, m)) {
      // a code object built for this handle's observation layout is preferred over a newer one built for another
      if (contact && it->layout_dim == dim && it->layout_kinds == kinds && it->layout_srcs == srcs) return &*it;
      if (!any) any = &*it;
    }
  return any;
}

namespace {

thread_local std::string g_create_error;

// Every entry point works on the device of its handle and leaves the caller's current device as it found it
// (a process may hold handles on several GPUs; torch keeps its own idea of the current device).
struct DeviceGuard {
  int prev = -1, dev = -1;
  explicit DeviceGuard(int device) : dev(device) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) (void)hipSetDevice(dev);
  }
  ~DeviceGuard() {
    if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
  }
};

struct SimBase {
  Os2rConfig cfg;
  std::string err;
  int nq = 0, D = 0;
  unsigned cmask = 0;
  int model_id = -1;  // matching constexpr model table, -1: run-time model kernels
  const JitEntry* jit = nullptr;  // registered model-specialised code object (os2r_register_model_kernels)
  bool dr = false;
  size_t esz = 8;
  unsigned long long step_count = 0;
  std::vector<void*> allocs;
  // device buffers (typed views below)
  void *model_d = nullptr, *task_d = nullptr;
  void *q = nullptr, *qd = nullptr, *hist = nullptr;
  void *mass_scale = nullptr, *damping = nullptr, *friction = nullptr, *mu = nullptr, *gravity = nullptr;
  int32_t* steps = nullptr;
  uint32_t* episode = nullptr;
  uint8_t* pose = nullptr;
  void* solver_l = nullptr;          // [4*nq][N]: the contact solver's state (os2r_get/set_solver_state)
  uint32_t* solver_flags = nullptr;  // [N]
  unsigned int* violations = nullptr;
  // scratch outputs for os2r_bench_steps
  void *b_obs = nullptr, *b_rew = nullptr, *b_term = nullptr;
  unsigned long long* counters = nullptr;   // caller-owned work-counter buffer (os2r_set_work_counters)
  uint8_t* b_done = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  uint16_t* reason = nullptr;           // caller-owned done-reason buffer (os2r_set_done_reasons)
  uint8_t* done_mask = nullptr;         // caller-owned done-mask buffer (os2r_set_done_mask)
  uint32_t* mirror_host = nullptr;      // two words of mapped, coherent host memory (os2r_get_violation_mirror) ...
  uint32_t* mirror_dev = nullptr;       // ... and the address the kernels write them through
  unsigned long long* debug = nullptr;  // diagnostic stamp builds only
};

}  // namespace

struct Os2rSim : SimBase {};

namespace {

#define HIP_TRY(sim, expr)                                                                     \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) {                                                                    \
      (sim)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                          \
      return OS2R_ERR_HIP;                                                                     \
    }                                                                                          \
  } while (0)

template <typename T>
void fill_model(const Os2rModel& m_in, bool contact, DevModel<T>& d) {
  Os2rModel m = m_in;
  if (!contact) m.ncand = 0;  // contact off: no candidates reach the kernels, whichever instantiation runs
  std::memset(&d, 0, sizeof(d));
  d.nq = m.nq;
  for (int i = 0; i < OS2R_MAX_DOF; ++i) {
    d.axis[i] = m.axis[i];
    for (int k = 0; k < 9; ++k) d.rfix[i][k] = (T)m.rfix[i][k];
    for (int k = 0; k < 3; ++k) { d.rpos[i][k] = (T)m.rpos[i][k]; d.com[i][k] = (T)m.com[i][k]; }
    for (int k = 0; k < 6; ++k) d.icom[i][k] = (T)m.icom[i][k];
    d.mass[i] = (T)m.mass[i];
    d.damping[i] = (T)m.damping[i];
    d.friction[i] = (T)m.friction[i];
    d.mu[i] = (T)m.mu[i];
  }
  for (int k = 0; k < 2; ++k) { d.act_dof[k] = m.act_dof[k]; d.max_torque[k] = (T)m.max_torque[k]; }
  d.gravity_z = (T)m.gravity_z;
  int k = 0;
  for (int b = 0; b < OS2R_MAX_DOF; ++b) {
    d.cand_begin[b] = k;
    while (k < m.ncand && m.cand_body[k] == b) ++k;
  }
  d.cand_begin[OS2R_MAX_DOF] = k;
  for (int c = 0; c < m.ncand; ++c)
    for (int j = 0; j < 3; ++j) d.cand_p[c][j] = (T)m.cand_p[c][j];
  for (int b = 0; b < OS2R_MAX_DOF; ++b) {
    for (int j = 0; j < 3; ++j) d.cand_center[b][j] = (T)m.cand_center[b][j];
    // a zero radius (model built without spheres) must never cull: make it cover everything
    d.cand_radius[b] = m.cand_radius[b] > 0.0 ? (T)(m.cand_radius[b] * 1.000001) : (T)1e30;
  }
}

template <typename T>
void fill_task(const Os2rConfig& cfg, DevTask<T>& d) {
  const Os2rTaskSpec& t = cfg.task;
  std::memset(&d, 0, sizeof(d));
  d.obs_dim = t.obs_dim;
  for (int i = 0; i < OS2R_MAX_OBS; ++i) {
    d.obs_kind[i] = t.obs_kind[i];
    d.obs_src[i] = t.obs_src[i];
    d.obs_low[i] = (T)t.obs_low[i];
    d.obs_high[i] = (T)t.obs_high[i];
    d.done_lo[i] = (T)t.done_lo[i];
    d.done_hi[i] = (T)t.done_hi[i];
  }
  d.reward_id = t.reward_id; d.normalized = t.normalized;
  d.idx_pitch_pos = t.idx_pitch_pos; d.idx_yaw_vel = t.idx_yaw_vel;
  d.idx_hip_pos = t.idx_hip_pos; d.idx_knee_pos = t.idx_knee_pos;
  d.max_episode_steps = t.max_episode_steps;
  d.reset_mode = t.reset_mode; d.n_reset_poses = t.n_reset_poses;
  for (int i = 0; i < OS2R_MAX_RESET_POSES; ++i) {
    d.reset_pose_id[i] = t.reset_pose_id[i]; d.reset_laying[i] = t.reset_laying[i];
    d.reset_pitch[i] = t.reset_pitch[i]; d.reset_hip[i] = t.reset_hip[i]; d.reset_knee[i] = t.reset_knee[i];
  }
  d.reset_simple = t.reset_simple;
  for (int i = 0; i < 6; ++i) d.leg_def[i] = t.leg_def[i];
  d.dof_yaw = t.dof_yaw; d.dof_pitch = t.dof_pitch; d.dof_bc = t.dof_bc; d.dof_hip = t.dof_hip; d.dof_knee = t.dof_knee;
  d.randomize_params = t.randomize_params;
  d.gravity_rollouts = t.gravity_rollouts;
  d.dr_gravity_mean = t.dr_gravity_mean; d.dr_gravity_std = t.dr_gravity_std;
  d.dr_mass_lo = t.dr_mass_lo; d.dr_mass_hi = t.dr_mass_hi;
  d.dr_friction_lo = t.dr_friction_lo; d.dr_friction_hi = t.dr_friction_hi;
  d.dr_damping_lo = t.dr_damping_lo; d.dr_damping_hi = t.dr_damping_hi;
  d.dr_mu_base = t.dr_mu_base; d.dr_mu_lo = t.dr_mu_lo; d.dr_mu_hi = t.dr_mu_hi;
  for (int i = 0; i < OS2R_MAX_DOF; ++i) d.nominal_damping[i] = cfg.model.damping[i];
}

int validate(const Os2rConfig* c, std::string& why) {
  if (!c) { why = "null config"; return 1; }
  if (c->abi_version != OS2R_ABI_VERSION) { why = "abi_version mismatch"; return 1; }
  if (c->dtype != OS2R_F32 && c->dtype != OS2R_F64) { why = "dtype must be OS2R_F32 or OS2R_F64"; return 1; }
  if (c->num_envs <= 0) { why = "num_envs must be positive"; return 1; }
  const Os2rModel& m = c->model;
  if (m.nq < 2 || m.nq > OS2R_MAX_DOF) { why = "model.nq must be 2..5"; return 1; }
  if (m.ncand < 0 || m.ncand > OS2R_MAX_CAND) { why = "model.ncand out of range"; return 1; }
  int last = 0;
  for (int k = 0; k < m.ncand; ++k) {
    if (m.cand_body[k] < last || m.cand_body[k] >= m.nq) { why = "cand_body must be non-decreasing and < nq"; return 1; }
    last = m.cand_body[k];
  }
  for (int i = 0; i < m.nq; ++i) {
    if (m.axis[i] < 0 || m.axis[i] > 2) { why = "joint axis must be 0,1,2"; return 1; }
    if (!(m.mass[i] > 0.0)) { why = "body mass must be positive"; return 1; }
  }
  for (int k = 0; k < 2; ++k)
    if (m.act_dof[k] < 0 || m.act_dof[k] >= m.nq) { why = "act_dof out of range"; return 1; }
  const Os2rTaskSpec& t = c->task;
  if (t.obs_dim < 1 || t.obs_dim > OS2R_MAX_OBS) { why = "task.obs_dim out of range"; return 1; }
  for (int d = 0; d < t.obs_dim; ++d) {
    const int kind = t.obs_kind[d];
    if (kind < OS2R_OBS_POS_NORM || kind > OS2R_OBS_TORQUE_RAW) { why = "unknown obs kind"; return 1; }
    const bool tq = kind == OS2R_OBS_TORQUE_NORM || kind == OS2R_OBS_TORQUE_RAW;
    if (t.obs_src[d] < 0 || t.obs_src[d] >= (tq ? 2 : m.nq)) { why = "obs_src out of range"; return 1; }
  }
  if (t.reward_id < 0 || t.reward_id > OS2R_REWARD_STRAIGHT_V1) { why = "unknown reward id"; return 1; }
  if (t.reward_id != OS2R_REWARD_STRAIGHT_V1 && t.idx_pitch_pos < 0) { why = "reward needs the pitch position observed"; return 1; }
  if (t.reward_id == OS2R_REWARD_HOPPING_V1 && t.idx_yaw_vel < 0) { why = "HoppingV1 needs the yaw velocity observed"; return 1; }
  if (t.reward_id == OS2R_REWARD_STRAIGHT_V1 && (t.idx_hip_pos < 0 || t.idx_knee_pos < 0)) { why = "StraightV1 needs hip and knee positions"; return 1; }
  if (t.n_reset_poses < 1 || t.n_reset_poses > OS2R_MAX_RESET_POSES) { why = "n_reset_poses out of range"; return 1; }
  if (c->substeps < 1 || c->substeps > 1000) { why = "substeps out of range"; return 1; }
  if (!(c->dt > 0.0)) { why = "dt must be positive"; return 1; }
  if (c->pgs_iters < 0 || c->pgs_iters > 10000) { why = "pgs_iters out of range"; return 1; }
  if (!(c->contact_margin >= 0.0)) { why = "contact_margin must be >= 0"; return 1; }
  if (t.gravity_rollouts < 0) { why = "gravity_rollouts must be >= 0"; return 1; }
  if (c->pgs_normal_iters < 0 || c->pgs_normal_iters > 10000) { why = "pgs_normal_iters out of range"; return 1; }
  if (!(c->pgs_tol >= 0.0)) { why = "pgs_tol must be >= 0"; return 1; }
  if (c->pgs_exact < 0 || c->pgs_exact > 10000) { why = "pgs_exact out of range"; return 1; }
  if (c->pgs_exact > 0 && c->dtype != OS2R_F64) { why = "pgs_exact (the exact finish of the contact solve) needs dtype f64"; return 1; }
  return 0;
}

template <typename T>
StepArgs<T> make_args(Os2rSim* s) {
  StepArgs<T> a;
  std::memset(&a, 0, sizeof(a));
  a.model = (const DevModel<T>*)s->model_d;
  a.task = (const DevTask<T>*)s->task_d;
  a.N = s->cfg.num_envs;
  a.env_offset = s->cfg.env_offset;
  a.seed = s->cfg.seed;
  a.step_count = s->step_count;
  a.substeps = s->cfg.substeps;
  a.rollout_steps = 0;
  a.pgs_iters = s->cfg.pgs_iters;
  a.pgs_normal_iters = s->cfg.pgs_normal_iters;
  a.pgs_exact = s->cfg.pgs_normal_iters > 0 ? s->cfg.pgs_exact : 0;   // the coupled pyramid has no fixed box to pivot on
  a.auto_reset = s->cfg.auto_reset;
  a.dt = (T)s->cfg.dt; a.erp = (T)s->cfg.erp; a.max_erv = (T)s->cfg.max_erv; a.margin = (T)s->cfg.contact_margin;
  a.gravity_z = (T)s->cfg.model.gravity_z;
  a.pgs_tol = (T)s->cfg.pgs_tol;
  a.counters = s->counters;
  a.q = (T*)s->q; a.qd = (T*)s->qd; a.hist = (T*)s->hist;
  a.mass_scale = (T*)s->mass_scale; a.damping = (T*)s->damping; a.friction = (T*)s->friction;
  a.mu = (T*)s->mu; a.gravity = (T*)s->gravity;
  a.steps = s->steps; a.episode = s->episode; a.pose = s->pose; a.violations = s->violations;
  a.solver_l = (T*)s->solver_l; a.solver_flags = s->solver_flags;
  a.debug = s->debug;
  a.reason = s->reason;
  a.done_mask = s->done_mask;
  a.mirror = s->mirror_dev;
  task_layout(s->cfg.task, a.layout_kinds, a.layout_srcs, a.layout_dim);
  return a;
}

template <typename T>
int do_reset(Os2rSim* s, const uint8_t* mask, void* obs, hipStream_t st) {
  StepArgs<T> a = make_args<T>(s);
  a.reset_mask = mask;
  a.obs = (T*)obs;
  if (Launcher<T>::reset(s->nq, s->dr, a, st) != 0) { s->err = "no reset kernel for this chain length"; return OS2R_ERR_INVALID; }
  HIP_TRY(s, hipGetLastError());
  return OS2R_OK;
}

template <typename T>
int do_step(Os2rSim* s, const void* actions, void* obs, void* reward, uint8_t* done, void* term, hipStream_t st) {
  StepArgs<T> a = make_args<T>(s);
  a.actions = (const T*)actions; a.obs = (T*)obs; a.reward = (T*)reward; a.done = done; a.term_obs = (T*)term;
  const bool contact = s->cfg.contact != 0 && s->cmask != 0u;
  const bool std_sweeps = is_std_solver<T>(a.pgs_iters, a.pgs_normal_iters, a.pgs_exact, s->cfg.model.nq);
  hipFunction_t jit_fn = !s->jit ? nullptr
      : (std_sweeps && s->jit->fn[contact][s->dr][1]) ? s->jit->fn[contact][s->dr][1] : s->jit->fn[contact][s->dr][0];
  if (s->jit && contact && std_sweeps && s->jit->fn_layout[s->dr] && s->jit->layout_dim == a.layout_dim &&
      s->jit->layout_kinds == a.layout_kinds && s->jit->layout_srcs == a.layout_srcs)
    jit_fn = s->jit->fn_layout[s->dr];
  if (a.counters && jit_fn) { s->err = "no counting variant in run-time code objects"; return OS2R_ERR_INVALID; }
  if (jit_fn) {
    // the robot's own code object: same StepArgs, passed as the kernel-argument buffer
    StepArgs<T> args = a;
    size_t size = sizeof(args);
    void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    const unsigned grid = (unsigned)((a.N + kWave - 1) / kWave);
    HIP_TRY(s, hipModuleLaunchKernel(jit_fn, grid, 1, 1, kWave, 1, 1, 0, st, nullptr, extra));
  } else if (Launcher<T>::step(s->nq, s->model_id, s->cfg.contact != 0, s->dr, a, st) != 0) {
    s->err = a.counters ? "no counting variant of the step kernel for this configuration" : "no step kernel for this chain length / contact mask";
    return OS2R_ERR_INVALID;
  }
  HIP_TRY(s, hipGetLastError());
  s->step_count += 1;
  return OS2R_OK;
}

// K env-steps: one launch of a fused variant where one exists, K launches of the step kernel otherwise (same results)
template <typename T>
int do_rollout(Os2rSim* s, int K, const void* actions, void* obs, void* reward, uint8_t* done, void* term, uint16_t* reason,
               hipStream_t st) {
  const size_t N = (size_t)s->cfg.num_envs, D = (size_t)s->D;
  if (!s->jit && !s->counters) {
    StepArgs<T> a = make_args<T>(s);
    a.actions = (const T*)actions; a.obs = (T*)obs; a.reward = (T*)reward; a.done = done; a.term_obs = (T*)term;
    a.reason = reason;
    a.done_mask = nullptr;   // (a rollout writes neither of the per-step buffers set on the handle)
    a.rollout_steps = K;
    const int rc = Launcher<T>::step(s->nq, s->model_id, s->cfg.contact != 0, s->dr, a, st);
    if (rc == 0) {
      HIP_TRY(s, hipGetLastError());
      s->step_count += (unsigned long long)K;
      return OS2R_OK;
    }
  }
  uint16_t* const reason_keep = s->reason;
  uint8_t* const mask_keep = s->done_mask;
  s->done_mask = nullptr;
  int rc = OS2R_OK;
  for (int k = 0; k < K && rc == OS2R_OK; ++k) {
    s->reason = reason ? reason + (size_t)k * N : nullptr;
    rc = do_step<T>(s, actions ? (const T*)actions + (size_t)k * N * 2 : nullptr, obs ? (T*)obs + (size_t)k * N * D : nullptr,
                    reward ? (T*)reward + (size_t)k * N : nullptr, done ? done + (size_t)k * N : nullptr,
                    term ? (T*)term + (size_t)k * N * D : nullptr, st);
  }
  s->reason = reason_keep;
  s->done_mask = mask_keep;
  return rc;
}

template <typename T>
int init_params(Os2rSim* s, hipStream_t st) {
  const long long N = s->cfg.num_envs;
  for (int i = 0; i < s->nq; ++i) {
    Launcher<T>::fill((T*)s->mass_scale + i * N, N, T(1), st);
    Launcher<T>::fill((T*)s->damping + i * N, N, (T)s->cfg.model.damping[i], st);
    Launcher<T>::fill((T*)s->friction + i * N, N, (T)s->cfg.model.friction[i], st);
    Launcher<T>::fill((T*)s->mu + i * N, N, (T)s->cfg.model.mu[i], st);
  }
  if (s->cfg.task.reset_mode == OS2R_RESET_RANDOM && s->cfg.task.dr_gravity_std > 0.0)
    Launcher<T>::gravity((T*)s->gravity, N, s->cfg.env_offset, s->cfg.seed, s->cfg.task.dr_gravity_mean,
                         s->cfg.task.dr_gravity_std, st);
  else
    Launcher<T>::fill((T*)s->gravity, N, (T)s->cfg.model.gravity_z, st);
  HIP_TRY(s, hipGetLastError());
  return OS2R_OK;
}

int dev_alloc(Os2rSim* s, void** p, size_t bytes) {
  hipError_t e = hipMalloc(p, bytes);
  if (e != hipSuccess) { s->err = std::string("hipMalloc: ") + hipGetErrorString(e); return OS2R_ERR_ALLOC; }
  s->allocs.push_back(*p);
  e = hipMemset(*p, 0, bytes);
  if (e != hipSuccess) { s->err = std::string("hipMemset: ") + hipGetErrorString(e); return OS2R_ERR_HIP; }
  return OS2R_OK;
}

void free_all(Os2rSim* s) {
  for (void* p : s->allocs) (void)hipFree(p);
  s->allocs.clear();
  if (s->ev0) (void)hipEventDestroy(s->ev0);
  if (s->ev1) (void)hipEventDestroy(s->ev1);
  if (s->mirror_host) (void)hipHostFree(s->mirror_host);
  s->mirror_host = s->mirror_dev = nullptr;
}

int param_view(Os2rSim* s, int field, void** base, int* count) {
  switch (field) {
    case OS2R_PARAM_MASS_SCALE: *base = s->mass_scale; *count = s->nq; return 0;
    case OS2R_PARAM_DAMPING: *base = s->damping; *count = s->nq; return 0;
    case OS2R_PARAM_FRICTION: *base = s->friction; *count = s->nq; return 0;
    case OS2R_PARAM_MU: *base = s->mu; *count = s->nq; return 0;
    case OS2R_PARAM_GRAVITY: *base = s->gravity; *count = 1; return 0;
    default: return 1;
  }
}

}  // namespace

extern "C" {

int os2r_abi_version(void) { return OS2R_ABI_VERSION; }

int os2r_create(const Os2rConfig* cfg, Os2rSim** out) {
  if (!out) { g_create_error = "null out pointer"; return OS2R_ERR_INVALID; }
  *out = nullptr;
  std::string why;
  if (validate(cfg, why)) { g_create_error = why; return OS2R_ERR_INVALID; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    g_create_error = "no HIP device visible: the stepper has no CPU fallback";
    return OS2R_ERR_NO_DEVICE;
  }
  if (cfg->device < 0 || cfg->device >= ndev) { g_create_error = "device ordinal out of range"; return OS2R_ERR_INVALID; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess) { g_create_error = "hipGetDeviceProperties failed"; return OS2R_ERR_HIP; }
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    g_create_error = std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only";
    return OS2R_ERR_NO_DEVICE;
  }
  Os2rSim* s = new (std::nothrow) Os2rSim();
  if (!s) { g_create_error = "out of host memory"; return OS2R_ERR_ALLOC; }
  s->cfg = *cfg;
  s->nq = cfg->model.nq;
  s->D = cfg->task.obs_dim;
  s->esz = cfg->dtype == OS2R_F64 ? 8 : 4;
  // parameter arrays are read per lane only when something can make them differ per env:
  // the randomising reset mode, or a later os2r_set_params (which flips this on)
  s->dr = cfg->task.reset_mode == OS2R_RESET_RANDOM;
  s->model_id = static_model_id(cfg->model);
  s->cmask = 0;
  if (cfg->contact)
    for (int k = 0; k < cfg->model.ncand; ++k) s->cmask |= 1u << cfg->model.cand_body[k];
  if (s->model_id < 0) s->jit = find_jit(cfg->model, cfg->dtype, cfg->device, cfg->contact != 0 && s->cmask != 0u, cfg->task);
  int rc = OS2R_OK;
  auto fail = [&](int code) { g_create_error = s->err; free_all(s); delete s; return code; };
  DeviceGuard guard(cfg->device);   // allocate and initialise on the handle's device, then give the caller's back
  if (hipSetDevice(cfg->device) != hipSuccess) { s->err = "hipSetDevice failed"; return fail(OS2R_ERR_HIP); }
  const size_t N = (size_t)cfg->num_envs, n = (size_t)s->nq, e = s->esz;
  if ((rc = dev_alloc(s, &s->q, n * N * e))) return fail(rc);
  if ((rc = dev_alloc(s, &s->qd, n * N * e))) return fail(rc);
  if ((rc = dev_alloc(s, &s->hist, 4 * N * e))) return fail(rc);
  if ((rc = dev_alloc(s, &s->mass_scale, n * N * e))) return fail(rc);
  if ((rc = dev_alloc(s, &s->damping, n * N * e))) return fail(rc);
  if ((rc = dev_alloc(s, &s->friction, n * N * e))) return fail(rc);
  if ((rc = dev_alloc(s, &s->mu, n * N * e))) return fail(rc);
  if ((rc = dev_alloc(s, &s->gravity, N * e))) return fail(rc);
  if ((rc = dev_alloc(s, (void**)&s->steps, N * 4))) return fail(rc);
  if ((rc = dev_alloc(s, (void**)&s->episode, N * 4))) return fail(rc);
  if ((rc = dev_alloc(s, (void**)&s->pose, N))) return fail(rc);
  if ((rc = dev_alloc(s, &s->solver_l, 4 * n * N * e))) return fail(rc);
  if ((rc = dev_alloc(s, (void**)&s->solver_flags, N * 4))) return fail(rc);
  if ((rc = dev_alloc(s, (void**)&s->violations, 4))) return fail(rc);
  if ((rc = dev_alloc(s, &s->b_obs, (size_t)s->D * N * e))) return fail(rc);
  if ((rc = dev_alloc(s, &s->b_rew, N * e))) return fail(rc);
  if ((rc = dev_alloc(s, &s->b_term, (size_t)s->D * N * e))) return fail(rc);
  if ((rc = dev_alloc(s, (void**)&s->b_done, N))) return fail(rc);
  if (cfg->dtype == OS2R_F64) {
    DevModel<double> hm; DevTask<double> ht;
    fill_model(cfg->model, cfg->contact != 0, hm); fill_task(*cfg, ht);
    if ((rc = dev_alloc(s, &s->model_d, sizeof(hm)))) return fail(rc);
    if ((rc = dev_alloc(s, &s->task_d, sizeof(ht)))) return fail(rc);
    if (hipMemcpy(s->model_d, &hm, sizeof(hm), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(s->task_d, &ht, sizeof(ht), hipMemcpyHostToDevice) != hipSuccess) { s->err = "model upload failed"; return fail(OS2R_ERR_HIP); }
  } else {
    DevModel<float> hm; DevTask<float> ht;
    fill_model(cfg->model, cfg->contact != 0, hm); fill_task(*cfg, ht);
    if ((rc = dev_alloc(s, &s->model_d, sizeof(hm)))) return fail(rc);
    if ((rc = dev_alloc(s, &s->task_d, sizeof(ht)))) return fail(rc);
    if (hipMemcpy(s->model_d, &hm, sizeof(hm), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(s->task_d, &ht, sizeof(ht), hipMemcpyHostToDevice) != hipSuccess) { s->err = "model upload failed"; return fail(OS2R_ERR_HIP); }
  }
  if (hipEventCreate(&s->ev0) != hipSuccess || hipEventCreate(&s->ev1) != hipSuccess) { s->err = "hipEventCreate failed"; return fail(OS2R_ERR_HIP); }
  // the violation mirror: pinned host memory that the device writes with plain system-scope stores (no atomics across PCIe)
  {
    void* hp = nullptr; void* dp = nullptr;
    if (hipHostMalloc(&hp, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) { s->err = "hipHostMalloc (violation mirror) failed"; return fail(OS2R_ERR_ALLOC); }
    s->mirror_host = (uint32_t*)hp;
    std::memset(hp, 0, 64);
    if (hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess) { s->err = "hipHostGetDevicePointer (violation mirror) failed"; return fail(OS2R_ERR_HIP); }
    s->mirror_dev = (uint32_t*)dp;
  }
  rc = cfg->dtype == OS2R_F64 ? init_params<double>(s, nullptr) : init_params<float>(s, nullptr);
  if (rc) return fail(rc);
  rc = cfg->dtype == OS2R_F64 ? do_reset<double>(s, nullptr, nullptr, nullptr) : do_reset<float>(s, nullptr, nullptr, nullptr);
  if (rc) return fail(rc);
  if (hipStreamSynchronize(nullptr) != hipSuccess) { s->err = "initial reset failed"; return fail(OS2R_ERR_HIP); }
  *out = s;
  return OS2R_OK;
}

int os2r_destroy(Os2rSim* sim) {
  if (!sim) return OS2R_ERR_INVALID;
  DeviceGuard guard(sim->cfg.device);
  (void)hipDeviceSynchronize();
  free_all(sim);
  delete sim;
  return OS2R_OK;
}

int os2r_reset(Os2rSim* sim, const uint8_t* mask_dev, void* obs_dev, void* stream) {
  if (!sim) return OS2R_ERR_INVALID;
  DeviceGuard guard(sim->cfg.device);
  return sim->cfg.dtype == OS2R_F64 ? do_reset<double>(sim, mask_dev, obs_dev, (hipStream_t)stream)
                                    : do_reset<float>(sim, mask_dev, obs_dev, (hipStream_t)stream);
}

int os2r_step(Os2rSim* sim, const void* actions_dev, void* obs_dev, void* reward_dev, uint8_t* done_dev,
              void* term_obs_dev, void* stream) {
  if (!sim) return OS2R_ERR_INVALID;
  DeviceGuard guard(sim->cfg.device);
  return sim->cfg.dtype == OS2R_F64
             ? do_step<double>(sim, actions_dev, obs_dev, reward_dev, done_dev, term_obs_dev, (hipStream_t)stream)
             : do_step<float>(sim, actions_dev, obs_dev, reward_dev, done_dev, term_obs_dev, (hipStream_t)stream);
}

int os2r_rollout(Os2rSim* sim, int nsteps, const void* actions_dev, void* obs_dev, void* reward_dev, uint8_t* done_dev,
                 void* term_obs_dev, uint16_t* reason_dev, void* stream) {
  if (!sim || nsteps < 1) return OS2R_ERR_INVALID;
  DeviceGuard guard(sim->cfg.device);
  return sim->cfg.dtype == OS2R_F64
             ? do_rollout<double>(sim, nsteps, actions_dev, obs_dev, reward_dev, done_dev, term_obs_dev, reason_dev, (hipStream_t)stream)
             : do_rollout<float>(sim, nsteps, actions_dev, obs_dev, reward_dev, done_dev, term_obs_dev, reason_dev, (hipStream_t)stream);
}

int os2r_get_state(Os2rSim* sim, void* q_dev, void* qd_dev, void* stream) {
  if (!sim) return OS2R_ERR_INVALID;
  DeviceGuard guard(sim->cfg.device);
  const size_t b = (size_t)sim->nq * sim->cfg.num_envs * sim->esz;
  if (q_dev) HIP_TRY(sim, hipMemcpyAsync(q_dev, sim->q, b, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  if (qd_dev) HIP_TRY(sim, hipMemcpyAsync(qd_dev, sim->qd, b, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return OS2R_OK;
}

int os2r_set_state(Os2rSim* sim, const void* q_dev, const void* qd_dev, void* stream) {
  if (!sim) return OS2R_ERR_INVALID;
  DeviceGuard guard(sim->cfg.device);
  const size_t b = (size_t)sim->nq * sim->cfg.num_envs * sim->esz;
  if (q_dev) HIP_TRY(sim, hipMemcpyAsync(sim->q, q_dev, b, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  if (qd_dev) HIP_TRY(sim, hipMemcpyAsync(sim->qd, qd_dev, b, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  // a state set from outside starts like a reset: the contact solver remembers nothing (os2r_set_solver_state restores it)
  HIP_TRY(sim, hipMemsetAsync(sim->solver_flags, 0, (size_t)sim->cfg.num_envs * 4, (hipStream_t)stream));
  HIP_TRY(sim, hipMemsetAsync(sim->solver_l, 0, 4 * b, (hipStream_t)stream));
  return OS2R_OK;
}

int os2r_get_solver_state(Os2rSim* sim, void* lambda_dev, uint32_t* flags_dev, void* stream) {
  if (!sim) return OS2R_ERR_INVALID;
  DeviceGuard guard(sim->cfg.device);
  const size_t N = (size_t)sim->cfg.num_envs;
  if (lambda_dev) HIP_TRY(sim, hipMemcpyAsync(lambda_dev, sim->solver_l, 4 * (size_t)sim->nq * N * sim->esz, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  if (flags_dev) HIP_TRY(sim, hipMemcpyAsync(flags_dev, sim->solver_flags, N * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return OS2R_OK;
}

int os2r_set_solver_state(Os2rSim* sim, const void* lambda_dev, const uint32_t* flags_dev, void* stream) {
  if (!sim || !lambda_dev || !flags_dev) return OS2R_ERR_INVALID;
  DeviceGuard guard(sim->cfg.device);
  const size_t N = (size_t)sim->cfg.num_envs;
  HIP_TRY(sim, hipMemcpyAsync(sim->solver_l, lambda_dev, 4 * (size_t)sim->nq * N * sim->esz, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  HIP_TRY(sim, hipMemcpyAsync(sim->solver_flags, flags_dev, N * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return OS2R_OK;
}

int os2r_get_action_history(Os2rSim* sim, int which, void* out_dev, void* stream) {
  if (!sim || which < 0 || which > 1 || !out_dev) return OS2R_ERR_INVALID;
  DeviceGuard guard(sim->cfg.device);
  const size_t b = 2 * (size_t)sim->cfg.num_envs * sim->esz;
  HIP_TRY(sim, hipMemcpyAsync(out_dev, (char*)sim->hist + which * b, b, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return OS2R_OK;
}

int os2r_set_action_history(Os2rSim* sim, int which, const void* in_dev, void* stream) {
  if (!sim || which < 0 || which > 1 || !in_dev) return OS2R_ERR_INVALID;
  DeviceGuard guard(sim->cfg.device);
  const size_t b = 2 * (size_t)sim->cfg.num_envs * sim->esz;
  HIP_TRY(sim, hipMemcpyAsync((char*)sim->hist + which * b, in_dev, b, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return OS2R_OK;
}

int os2r_set_params(Os2rSim* sim, int field, const void* src_dev, void* stream) {
  void* base; int count;
  if (!sim || !src_dev || param_view(sim, field, &base, &count)) return OS2R_ERR_INVALID;
  DeviceGuard guard(sim->cfg.device);
  HIP_TRY(sim, hipMemcpyAsync(base, src_dev, (size_t)count * sim->cfg.num_envs * sim->esz, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  sim->dr = true;
  return OS2R_OK;
}

int os2r_get_params(Os2rSim* sim, int field, void* dst_dev, void* stream) {
  void* base; int count;
  if (!sim || !dst_dev || param_view(sim, field, &base, &count)) return OS2R_ERR_INVALID;
  DeviceGuard guard(sim->cfg.device);
  HIP_TRY(sim, hipMemcpyAsync(dst_dev, base, (size_t)count * sim->cfg.num_envs * sim->esz, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return OS2R_OK;
}

int os2r_get_episode_info(Os2rSim* sim, int32_t* steps_dev, uint32_t* episode_dev, uint8_t* pose_dev, void* stream) {
  if (!sim) return OS2R_ERR_INVALID;
  DeviceGuard guard(sim->cfg.device);
  const size_t N = (size_t)sim->cfg.num_envs;
  if (steps_dev) HIP_TRY(sim, hipMemcpyAsync(steps_dev, sim->steps, N * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  if (episode_dev) HIP_TRY(sim, hipMemcpyAsync(episode_dev, sim->episode, N * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  if (pose_dev) HIP_TRY(sim, hipMemcpyAsync(pose_dev, sim->pose, N, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return OS2R_OK;
}

int os2r_set_episode_info(Os2rSim* sim, const int32_t* steps_dev, const uint32_t* episode_dev, const uint8_t* pose_dev, void* stream) {
  if (!sim) return OS2R_ERR_INVALID;
  DeviceGuard guard(sim->cfg.device);
  const size_t N = (size_t)sim->cfg.num_envs;
  if (steps_dev) HIP_TRY(sim, hipMemcpyAsync(sim->steps, steps_dev, N * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  if (episode_dev) HIP_TRY(sim, hipMemcpyAsync(sim->episode, episode_dev, N * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  if (pose_dev) HIP_TRY(sim, hipMemcpyAsync(sim->pose, pose_dev, N, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return OS2R_OK;
}

int os2r_model_is_compiled_in(const Os2rModel* model) {
  return model && os2r::static_model_id(*model) >= 0 ? 1 : 0;
}

int os2r_register_model_kernels(const Os2rModel* model, int32_t dtype, int32_t device, const char* path) {
  if (!model || !path || (dtype != OS2R_F32 && dtype != OS2R_F64)) { g_create_error = "os2r_register_model_kernels: bad argument"; return OS2R_ERR_INVALID; }
  DeviceGuard guard(device);
  if (hipSetDevice(device) != hipSuccess) { g_create_error = "os2r_register_model_kernels: hipSetDevice failed"; return OS2R_ERR_HIP; }
  JitEntry e;
  e.model = *model; e.dtype = dtype; e.device = device;
  hipError_t rc = hipModuleLoad(&e.module, path);
  if (rc != hipSuccess) { g_create_error = std::string("hipModuleLoad(") + path + "): " + hipGetErrorString(rc); return OS2R_ERR_HIP; }
  int found = 0;
  for (int c = 0; c < 2; ++c)
    for (int d = 0; d < 2; ++d) {
      const std::string name = std::string("os2r_jit_step_c") + char('0' + c) + "_d" + char('0' + d);
      for (int v = 0; v < 2; ++v)
        if (hipModuleGetFunction(&e.fn[c][d][v], e.module, (name + (v ? "_s" : "")).c_str()) == hipSuccess) ++found;
        else e.fn[c][d][v] = nullptr;
    }
  hipDeviceptr_t lay = nullptr;
  size_t lay_bytes = 0;
  if (hipModuleGetGlobal(&lay, &lay_bytes, e.module, "os2r_jit_layout") == hipSuccess && lay_bytes >= 3 * sizeof(unsigned long long)) {
    unsigned long long v[3] = {};
    if (hipMemcpy(v, lay, sizeof(v), hipMemcpyDeviceToHost) == hipSuccess) {
      bool both = true;
      for (int d = 0; d < 2; ++d)
        if (hipModuleGetFunction(&e.fn_layout[d], e.module, (std::string("os2r_jit_step_c1_d") + char('0' + d) + "_l").c_str()) != hipSuccess) { e.fn_layout[d] = nullptr; both = false; }
      if (both) { e.layout_kinds = v[0]; e.layout_srcs = v[1]; e.layout_dim = (int)v[2]; }
    }
  }
  (void)hipGetLastError();   // a missing variant is not an error
  if (!found) { (void)hipModuleUnload(e.module); g_create_error = std::string(path) + " exports no os2r_jit_step_* kernel"; return OS2R_ERR_INVALID; }
  std::lock_guard<std::mutex> lock(g_jit_mutex);
  g_jit.push_back(e);
  return OS2R_OK;
}

int os2r_get_action_violations(Os2rSim* sim, uint32_t* dst, int32_t clear, void* stream) {
  if (!sim || !dst) return OS2R_ERR_INVALID;
  DeviceGuard guard(sim->cfg.device);
  HIP_TRY(sim, hipMemcpyAsync(dst, sim->violations, 4, hipMemcpyDefault, (hipStream_t)stream));
  if (clear) HIP_TRY(sim, hipMemsetAsync(sim->violations, 0, 4, (hipStream_t)stream));
  return OS2R_OK;
}

int os2r_get_violation_mirror(Os2rSim* sim, const volatile uint32_t** host_words) {
  if (!sim || !host_words) return OS2R_ERR_INVALID;
  *host_words = sim->mirror_host;
  return OS2R_OK;
}

int os2r_get_step_count(Os2rSim* sim, uint64_t* out) {
  if (!sim || !out) return OS2R_ERR_INVALID;
  *out = sim->step_count;
  return OS2R_OK;
}

int os2r_set_step_count(Os2rSim* sim, uint64_t value) {
  if (!sim) return OS2R_ERR_INVALID;
  sim->step_count = value;
  return OS2R_OK;
}

int os2r_bench_steps(Os2rSim* sim, int nsteps, void* stream, float* elapsed_ms) {
  if (!sim || nsteps < 1) return OS2R_ERR_INVALID;
  DeviceGuard guard(sim->cfg.device);
  hipStream_t st = (hipStream_t)stream;
  if (elapsed_ms) HIP_TRY(sim, hipEventRecord(sim->ev0, st));
  for (int k = 0; k < nsteps; ++k) {
    int rc = os2r_step(sim, nullptr, sim->b_obs, sim->b_rew, sim->b_done, sim->b_term, stream);
    if (rc) return rc;
  }
  if (!elapsed_ms) return OS2R_OK;   // enqueue only: several handles on several streams are timed by their caller
  HIP_TRY(sim, hipEventRecord(sim->ev1, st));
  // the caller's clock runs until this returns: poll the event instead of sleeping on it (a blocked host thread is woken tens of
  // microseconds after the last launch has finished -- 1-2 % of a 20-step window)
  for (;;) {
    const hipError_t q = hipEventQuery(sim->ev1);
    if (q == hipSuccess) break;
    if (q != hipErrorNotReady) { sim->err = std::string("hipEventQuery: ") + hipGetErrorString(q); return OS2R_ERR_HIP; }
  }
  HIP_TRY(sim, hipEventElapsedTime(elapsed_ms, sim->ev0, sim->ev1));
  return OS2R_OK;
}

int os2r_bench_steps_multi(Os2rSim* const* sims, void* const* streams, int count, int nsteps) {
  if (!sims || !streams || count < 1 || nsteps < 1) return OS2R_ERR_INVALID;
  for (int i = 0; i < count; ++i)
    if (!sims[i]) return OS2R_ERR_INVALID;
  for (int k = 0; k < nsteps; ++k)
    for (int i = 0; i < count; ++i) {
      Os2rSim* s = sims[i];
      int rc = os2r_step(s, nullptr, s->b_obs, s->b_rew, s->b_done, s->b_term, streams[i]);
      if (rc) return rc;
    }
  return OS2R_OK;
}

int os2r_set_work_counters(Os2rSim* sim, uint64_t* counters_dev) {
  if (!sim) return OS2R_ERR_INVALID;
  sim->counters = (unsigned long long*)counters_dev;
  return OS2R_OK;
}

int os2r_set_done_reasons(Os2rSim* sim, uint16_t* reason_dev) {
  if (!sim) return OS2R_ERR_INVALID;
  sim->reason = reason_dev;
  return OS2R_OK;
}

int os2r_set_done_mask(Os2rSim* sim, uint8_t* mask_dev) {
  if (!sim) return OS2R_ERR_INVALID;
  sim->done_mask = mask_dev;
  return OS2R_OK;
}

#ifdef OS2R_STAMPS
// diagnostic builds only (libos2r_stamps.so): per-wave phase stamps, kStamps x uint64 per workgroup
OS2R_API int os2r_debug_set_stamp_buffer(Os2rSim* sim, unsigned long long* buf_dev) {
  if (!sim) return OS2R_ERR_INVALID;
  sim->debug = buf_dev;
  return OS2R_OK;
}
#endif

const char* os2r_last_error(Os2rSim* sim) { return sim ? sim->err.c_str() : g_create_error.c_str(); }

}  // extern "C"
