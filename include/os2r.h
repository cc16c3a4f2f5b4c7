/*
 * os2r.h — C-ABI of the MI355X-native batched monopod stepper.
 *
 * This is the drop-in boundary for the gym_os2r "env.step" hot path.  The
 * reference reaches its physics backend through these Python->SWIG call sites
 * (all paths relative to the reference checkout):
 *
 *   gym_os2r/runtimes/gazebo_runtime.py:70-77   10x { task.set_action(a); gazebo.run() }
 *   gym_os2r/runtimes/gazebo_runtime.py:80-95   get_observation / get_reward / is_done / get_info
 *   gym_os2r/runtimes/gazebo_runtime.py:111-114 scenario.GazeboSimulator(1/physics_rate, rtf, steps_per_run=1)
 *   gym_os2r/tasks/monopod.py:225,234           model.set_joint_generalized_force_targets / ..._targets()
 *   gym_os2r/tasks/monopod.py:248-249           model.joint_positions / joint_velocities
 *   gym_os2r/randomizers/monopod.py:125-128     model.to_gazebo().reset_joint_positions / _velocities
 *   gym_os2r/randomizers/monopod.py:60          world.to_gazebo().set_gravity
 *   gym_os2r/randomizers/monopod.py:182-215     per-reset SDF randomisation (mass/friction/damping/mu)
 *
 * One Os2rSim owns the state of N independent environments resident in HBM
 * (struct-of-arrays, one GPU lane per environment).  Every entry point returns
 * an int status (0 = OK), never throws, and is stream-ordered on the hipStream_t
 * passed as `void* stream` (NULL = the default stream).  All `*_dev` pointers
 * are device pointers owned by the caller (e.g. torch tensors' data_ptr()); the
 * library owns only its internal state.  A handle is not thread-safe.
 *
 * Plain C: no torch / pybind / HIP types appear in any signature.
 */
#ifndef OS2R_H_
#define OS2R_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: the entry points declared here are its whole dynamic symbol table. */
#if defined(__GNUC__) || defined(__clang__)
#define OS2R_API __attribute__((visibility("default")))
#else
#define OS2R_API
#endif

#define OS2R_ABI_VERSION 5

#define OS2R_MAX_DOF 5      /* yaw, pitch, boom_connector, hip, knee                       */
#define OS2R_MAX_CAND 192   /* ground-contact candidate points of one model                */
#define OS2R_MAX_OBS 12     /* 5 pos + 5 vel + 2 measured torques                          */
#define OS2R_MAX_RESET_POSES 8

/* status codes */
enum {
  OS2R_OK = 0,
  OS2R_ERR_INVALID = 1,   /* bad argument / inconsistent config                            */
  OS2R_ERR_HIP = 2,       /* a HIP runtime call failed, see os2r_last_error                 */
  OS2R_ERR_NO_DEVICE = 3, /* no gfx950 device visible                                       */
  OS2R_ERR_ALLOC = 4
};

/* arithmetic type of the device path */
enum { OS2R_F32 = 0, OS2R_F64 = 1 };

/* ------------------------------------------------------------------------- *
 * Compiled robot model: a serial chain of nq revolute joints hanging off the
 * world, produced by the model compiler from a URDF (+ STL collision meshes)
 * with fixed joints lumped into their parent body
 * (gym_os2r/models/models/<variant>/<variant>.urdf).
 * Joint i connects body i-1 (body -1 = world) to body i.
 * ------------------------------------------------------------------------- */
typedef struct Os2rModel {
  int32_t nq;                          /* 2..5 degrees of freedom                            */
  int32_t axis[OS2R_MAX_DOF];          /* joint axis in the joint frame: 0=x 1=y 2=z         */
  double rfix[OS2R_MAX_DOF][9];        /* row-major rotation: joint frame in parent body     */
  double rpos[OS2R_MAX_DOF][3];        /* joint origin in parent body frame [m]              */
  double mass[OS2R_MAX_DOF];           /* (lumped) body mass [kg]                            */
  double com[OS2R_MAX_DOF][3];         /* centre of mass in body frame [m]                   */
  double icom[OS2R_MAX_DOF][6];        /* inertia about COM, body axes: xx xy xz yy yz zz    */
  double damping[OS2R_MAX_DOF];        /* viscous joint damping [N m s/rad]                  */
  double friction[OS2R_MAX_DOF];       /* Coulomb joint friction [N m]                       */
  double mu[OS2R_MAX_DOF];             /* contact friction coefficient of body i vs ground   */
  int32_t act_dof[2];                  /* dof index of hip_joint, knee_joint                 */
  double max_torque[2];                /* [N m], settings.yaml spaces/action                 */
  double gravity_z;                    /* world gravity along z [m/s^2] (negative)           */
  int32_t ncand;                       /* number of contact candidate points                 */
  int32_t cand_body[OS2R_MAX_CAND];    /* body index of each candidate, non-decreasing       */
  double cand_p[OS2R_MAX_CAND][3];     /* candidate position in its body frame [m]           */
  double cand_center[OS2R_MAX_DOF][3]; /* bounding sphere of body i's candidates (body frame) */
  double cand_radius[OS2R_MAX_DOF];    /*   used to skip the scan of a body far from the ground */
} Os2rModel;

/* observation slot kinds (gym_os2r/tasks/monopod.py:238-272, monopod_no_norm.py:222-246) */
enum {
  OS2R_OBS_POS_NORM = 0,       /* 2*(x-low)/(high-low)-1                                   */
  OS2R_OBS_POS_PERIODIC_NORM,  /* wrap to [-pi,pi) then the affine map                     */
  OS2R_OBS_VEL_TANH,           /* tanh(0.05*v)                                              */
  OS2R_OBS_TORQUE_NORM,        /* previous action through the affine map (low=-1, high=1)  */
  OS2R_OBS_POS_RAW,            /* no_norm task: x                                           */
  OS2R_OBS_POS_PERIODIC_RAW,   /* no_norm task: wrap only                                   */
  OS2R_OBS_VEL_RAW,            /* no_norm task: v                                           */
  OS2R_OBS_TORQUE_RAW          /* no_norm task: previous action                             */
};

/* reward ids (gym_os2r/rewards/__init__.py) */
enum {
  OS2R_REWARD_BALANCING_V1 = 0, /* :66-81  (StandingV1 :135-149 is the same formula)        */
  OS2R_REWARD_BALANCING_V2 = 1, /* :83-101                                                   */
  OS2R_REWARD_BALANCING_V3 = 2, /* :103-131                                                  */
  OS2R_REWARD_STANDING_V1 = 3,
  OS2R_REWARD_HOPPING_V1 = 4,   /* :151-180                                                  */
  OS2R_REWARD_STRAIGHT_V1 = 5   /* :183-207                                                  */
};

/* reset modes */
enum {
  OS2R_RESET_FIXED = 0,   /* MonopodEnvNoRandomizer (randomizers/monopod_no_rand.py:59-98)  */
  OS2R_RESET_RANDOM = 1   /* MonopodEnvRandomizer   (randomizers/monopod.py:89-128,182-215) */
};

typedef struct Os2rTaskSpec {
  int32_t obs_dim;
  int32_t obs_kind[OS2R_MAX_OBS];
  int32_t obs_src[OS2R_MAX_OBS];   /* dof index (pos/vel kinds) or action index (torque)    */
  double obs_low[OS2R_MAX_OBS];    /* normalisation limits (periodic: -(pi+eps))            */
  double obs_high[OS2R_MAX_OBS];
  double done_lo[OS2R_MAX_OBS];    /* done iff !(done_lo <= y <= done_hi), y = value before */
  double done_hi[OS2R_MAX_OBS];    /*   the affine/tanh map (after the periodic wrap)       */
  int32_t reward_id;
  int32_t normalized;              /* 1: MonopodTask, 0: monopod_no_norm.MonopodTask        */
  int32_t idx_pitch_pos;           /* obs indices the rewards read; -1 if masked/absent     */
  int32_t idx_yaw_vel;
  int32_t idx_hip_pos;
  int32_t idx_knee_pos;
  int32_t max_episode_steps;       /* gym TimeLimit (gym_os2r/__init__.py:19,56); 0 = none  */
  /* reset */
  int32_t reset_mode;
  int32_t n_reset_poses;
  int32_t reset_pose_id[OS2R_MAX_RESET_POSES];   /* index into the settings 'resets' table  */
  int32_t reset_laying[OS2R_MAX_RESET_POSES];
  double reset_pitch[OS2R_MAX_RESET_POSES];
  double reset_hip[OS2R_MAX_RESET_POSES];        /* precomputed IK (utils/reset.py) or 1.57 */
  double reset_knee[OS2R_MAX_RESET_POSES];
  int32_t reset_simple;            /* 1: task_mode 'simple' (hip,knee ~ U(-1,1), quirk)     */
  double leg_def[6];               /* ul, ll, cph, lb, hip_offset, clipping_adjust [mm]     */
  int32_t dof_yaw, dof_pitch, dof_bc, dof_hip, dof_knee; /* chain dof index or -1           */
  /* domain randomisation (randomizers/monopod.py:56-61,182-215); used when reset_mode==1  */
  int32_t randomize_params;        /* 1: resample mass/friction/damping/mu at every reset   */
  double dr_mass_lo, dr_mass_hi;          /* coefficient, U(0.8,1.2)                        */
  double dr_friction_lo, dr_friction_hi;  /* absolute,    U(0.01,0.05)                      */
  double dr_damping_lo, dr_damping_hi;    /* coefficient, U(0.8,1.2), zeros ignored         */
  double dr_mu_base, dr_mu_lo, dr_mu_hi;  /* 0.33 * U(0.8,1.2)                              */
  double dr_gravity_mean, dr_gravity_std; /* N(-9.8,0.2), drawn once at create              */
  int32_t gravity_rollouts;        /* > 0: gravity is drawn anew for an environment after every    */
                                   /*   this many of its rollouts (the reference re-creates the    */
                                   /*   simulator then: randomizers/monopod.py:36-41,56-61,371);   */
                                   /*   0: once at create                                          */
  int32_t reserved_;
} Os2rTaskSpec;

typedef struct Os2rConfig {
  int32_t abi_version;     /* must be OS2R_ABI_VERSION                                      */
  int32_t dtype;           /* OS2R_F32 / OS2R_F64                                            */
  int64_t num_envs;        /* environments owned by this handle (this rank's shard)          */
  int64_t env_offset;      /* global index of local env 0 (multi-GPU sharding; RNG key)      */
  uint64_t seed;
  int32_t device;          /* HIP device ordinal                                             */
  int32_t substeps;        /* physics_rate/agent_rate = 10 (runtimes/gazebo_runtime.py:46)   */
  double dt;               /* 1/physics_rate = 1e-4 s                                        */
  int32_t contact;         /* 0: ground contact off (bring-up config C2), 1: on              */
  int32_t pgs_iters;       /* upper bound on the projected Gauss-Seidel sweeps per substep   */
                           /*   over all rows (phase 2)                                      */
  int32_t pgs_normal_iters;/* preceding sweeps over normal + joint-friction rows that fix    */
                           /*   the tangential bounds (0: coupled pyramid)    */
  int32_t auto_reset;      /* SubprocVecEnv semantics (common/vec_env/subproc_vec_env.py:15) */
  double erp;              /* contact error-reduction parameter                              */
  double max_erv;          /* cap on the error-reduction velocity [m/s]                      */
  double contact_margin;   /* candidates closer than this to the ground join the contact [m] */
  double pgs_tol;          /* an environment stops sweeping once a checked sweep moved no more     */
                           /*   energy than this [J]; 0: exact fixed points only                   */
  int32_t pgs_exact;       /* exact finish of the boxed LCP: an environment that   */
                           /*   has not converged after the first three sweeps of phase 2 (which    */
                           /*   starts from the impulses of the environment's previous iteration,   */
                           /*   os2r_get_solver_state) solves its                                   */
                           /*   free rows exactly (a 5x5 system in the whitened velocities), with  */
                           /*   active-set pivots (a step cut at a bound; an inconsistent free set */
                           /*   left by a step to the first bound), at most this many solves per   */
                           /*   substep; a checked sweep follows each unblocked solve.             */
                           /*   0: sweeps only, checked every                                      */
                           /*   4th (the round-1/2 solver).  Ignored with pgs_normal_iters == 0;   */
                           /*   needs dtype OS2R_F64.                                              */
  int32_t reserved0_;
  Os2rModel model;
  Os2rTaskSpec task;
} Os2rConfig;

/* per-environment parameter arrays, SoA [count][num_envs] in the handle's dtype */
enum {
  OS2R_PARAM_MASS_SCALE = 0, /* [nq]  body mass coefficient                                 */
  OS2R_PARAM_DAMPING = 1,    /* [nq]  absolute damping                                       */
  OS2R_PARAM_FRICTION = 2,   /* [nq]  absolute Coulomb friction                              */
  OS2R_PARAM_MU = 3,         /* [nq]  body-vs-ground friction coefficient                    */
  OS2R_PARAM_GRAVITY = 4     /* [1]   gravity_z                                              */
};

typedef struct Os2rSim Os2rSim;

OS2R_API int os2r_abi_version(void);

/* Allocates device state for cfg->num_envs environments, initialises per-env
 * parameters to the model's nominal values (gravity: N(mean,std) per env when
 * task.reset_mode==OS2R_RESET_RANDOM) and performs a full reset. */
OS2R_API int os2r_create(const Os2rConfig* cfg, Os2rSim** out);
OS2R_API int os2r_destroy(Os2rSim* sim);

/* Reset the environments whose mask byte is non-zero (mask_dev == NULL: all).
 * Replaces GazeboEnvRandomizer.reset -> randomize_task -> task.reset_task.
 * obs_dev (nullable) receives the [num_envs, obs_dim] observation of every env. */
OS2R_API int os2r_reset(Os2rSim* sim, const uint8_t* mask_dev, void* obs_dev, void* stream);

/* One env-step for every environment: `substeps` physics iterations with the
 * action held, then observation, reward, done; done environments are reset in
 * the same launch when auto_reset is set.  Replaces GazeboRuntime.step.
 *   actions_dev  [num_envs,2] in the handle's dtype, values in [-1,1]
 *                (NULL: draw U(-1,1) actions on device from the counter RNG)
 *   obs_dev      [num_envs,obs_dim]   observation (post-reset if auto-reset)
 *   reward_dev   [num_envs]
 *   done_dev     [num_envs] uint8     bit0 done, bit1 TimeLimit truncation,
 *                                     bit2 non-finite state guard
 *   term_obs_dev [num_envs,obs_dim]   nullable; observation before auto-reset
 *                                     (info['terminal_observation'])            */
OS2R_API int os2r_step(Os2rSim* sim, const void* actions_dev, void* obs_dev, void* reward_dev,
              uint8_t* done_dev, void* term_obs_dev, void* stream);

/* `nsteps` env-steps of every environment, as `nsteps` calls of os2r_step would make them -- bit for bit -- but without a
 * device-wide barrier between the env-steps: where a fused variant of the step kernel exists (the compiled-in robots
 * with ground contact, the default solver settings and a reference task layout) the whole rollout is ONE launch in which
 * every wave advances its own 64 environments step after step, state in registers; otherwise the library makes the
 * `nsteps` launches itself.  For open-loop action sequences and random rollouts (the reference's workers advance
 * independently of each other: gym_os2r/common/vec_env/subproc_vec_env.py:15-21); a policy in the loop needs os2r_step.
 *   actions_dev  [nsteps][num_envs][2] or NULL (on-device U(-1,1) actions, step counter as in os2r_step)
 *   obs_dev      [nsteps][num_envs][obs_dim], reward_dev [nsteps][num_envs], done_dev [nsteps][num_envs] uint8,
 *   term_obs_dev [nsteps][num_envs][obs_dim] (nullable), reason_dev [nsteps][num_envs] uint16 (nullable; the done
 *                reasons of os2r_set_done_reasons per step -- the buffer set there is not written by a rollout)  */
OS2R_API int os2r_rollout(Os2rSim* sim, int nsteps, const void* actions_dev, void* obs_dev, void* reward_dev,
                          uint8_t* done_dev, void* term_obs_dev, uint16_t* reason_dev, void* stream);

/* Model-specialised kernels.  A robot that is not one of the four compiled-in reference variants
 * runs on generic kernels that read its constants through scalar loads (about half the speed).
 * A host binding may instead compile the step kernels for that robot -- gym_os2r_amd/jit.py does:
 * it writes the robot's constants as a constexpr table and runs `hipcc --genco` on
 * gym-os2r_amd/csrc/os2r_jit_unit.hip -- and register the code object here.  os2r_create then
 * uses it for every handle on `device` whose dtype matches and whose Os2rModel equals `model`
 * bit for bit (gravity_z excluded: it is a per-handle value).  Variants the code object does not
 * export (os2r_jit_step_c{0,1}_d{0,1}: contact off/on, per-env parameters off/on) fall back to the
 * generic kernels.  A code object may also export os2r_jit_step_c1_d{0,1}_l together with the data
 * symbol os2r_jit_layout = {kinds, sources, slots} (4 bits per observation slot): the contact
 * kernels with that observation layout folded in; handles whose task has exactly that layout use
 * them, and of several code objects of one robot the one built for the handle's layout is taken.
 * Errors: os2r_last_error(NULL).                                                            */
OS2R_API int os2r_model_is_compiled_in(const Os2rModel* model);
OS2R_API int os2r_register_model_kernels(const Os2rModel* model, int32_t dtype, int32_t device,
                                const char* code_object_path);

/* Caller-provided actions outside [-1, 1]: the reference asserts on them in Python
 * (tasks/monopod.py:222, runtimes/gazebo_runtime.py:67-68) and its backend clamps the torque
 * (tasks/monopod.py:313-316).  The kernel clamps and counts them; this copies the running count
 * of offending environments to dst (device or pinned host memory, ordered on the stream, so a
 * host binding can look at it one call later without stalling) and clears it if `clear`.   */
OS2R_API int os2r_get_action_violations(Os2rSim* sim, uint32_t* dst, int32_t clear, void* stream);
/* The same count without a copy on the step path (ABI 5): *host_words points at two 32-bit words of pinned host memory owned by
 * the handle.  The first wave of every os2r_step / os2r_rollout launch stores there [0] the running count as EARLIER launches left
 * it (what os2r_get_action_violations would copy before this launch; never cleared by this path) and [1] the low 32 bits of the
 * launch's step counter (os2r_get_step_count before the call), so a host binding reads the verdict on step k once [1] > k --
 * a plain load, no event, no memcpy between the launches (a 4-byte copy plus an event per step cost a 65 536-env gym-level
 * loop 8 % of its rate: profiles/r05_host_surface.txt).  Valid until os2r_destroy.                                          */
OS2R_API int os2r_get_violation_mirror(Os2rSim* sim, const volatile uint32_t** host_words);

/* State access in chain dof order, SoA [nq][num_envs], handle's dtype.  os2r_set_state also clears the contact
 * solver's state (below): a state set from outside starts like a reset.                                     */
OS2R_API int os2r_get_state(Os2rSim* sim, void* q_dev, void* qd_dev, void* stream);
OS2R_API int os2r_set_state(Os2rSim* sim, const void* q_dev, const void* qd_dev, void* stream);

/* The contact solver's state (fp64 handles with the exact finish; carried but unused otherwise).  The reference's
 * backend keeps one persistent constraint solver per world (behind gym_os2r/runtimes/gazebo_runtime.py:76,111-114);
 * here every environment remembers the impulses that ended its last physics iteration, and the next one -- of the same
 * env-step or of the next -- starts its contact solve from them.  A reset clears it.
 *   lambda_dev [4*nq][num_envs], handle's dtype: rows b, nq + b, 2nq + b: normal and the two tangential (world x, y)
 *              impulses of body b's ground contact; row 3nq + j: Coulomb friction impulse of joint j
 *   flags_dev  [num_envs] uint32: bit b: body b had a contact (its three impulses are remembered); bit 31: an iteration
 *              has run since the reset (the joint impulses are remembered)
 * Part of a checkpoint: restore it after os2r_set_state.  get: either pointer may be NULL.                        */
OS2R_API int os2r_get_solver_state(Os2rSim* sim, void* lambda_dev, uint32_t* flags_dev, void* stream);
OS2R_API int os2r_set_solver_state(Os2rSim* sim, const void* lambda_dev, const uint32_t* flags_dev, void* stream);

/* Action history: [2][num_envs] SoA; which=0 last applied action, 1 the one before
 * (tasks/monopod.py:95-98,232-235).                                              */
OS2R_API int os2r_get_action_history(Os2rSim* sim, int which, void* out_dev, void* stream);
OS2R_API int os2r_set_action_history(Os2rSim* sim, int which, const void* in_dev, void* stream);

/* Per-env parameter arrays (domain randomisation), SoA [count][num_envs]. */
OS2R_API int os2r_set_params(Os2rSim* sim, int field, const void* src_dev, void* stream);
OS2R_API int os2r_get_params(Os2rSim* sim, int field, void* dst_dev, void* stream);

/* Episode bookkeeping: elapsed steps (int32), episode index (uint32), reset pose
 * id (uint8, info['reset_orientation']); any pointer may be NULL.               */
OS2R_API int os2r_get_episode_info(Os2rSim* sim, int32_t* steps_dev, uint32_t* episode_dev,
                          uint8_t* pose_dev, void* stream);
/* The inverse (any pointer may be NULL): together with os2r_set_state, os2r_set_action_history, os2r_set_params
 * and os2r_set_step_count it restores a handle exactly -- a checkpoint resumed on another handle continues bit
 * for bit (the reference has no save/restore, only env.seed).                                              */
OS2R_API int os2r_set_episode_info(Os2rSim* sim, const int32_t* steps_dev, const uint32_t* episode_dev,
                          const uint8_t* pose_dev, void* stream);

/* Global step counter that keys the on-device action RNG. */
OS2R_API int os2r_get_step_count(Os2rSim* sim, uint64_t* out);
OS2R_API int os2r_set_step_count(Os2rSim* sim, uint64_t value);

/* Timing helper for benchmarks: runs `nsteps` os2r_step launches with on-device
 * random actions on the given stream, bracketed by HIP events recorded on that
 * stream; returns the elapsed GPU time in milliseconds. Every output of os2r_step
 * (observation, reward, done, terminal observation) goes to internal scratch
 * buffers: the timed launch is the one a gym-level env.step makes.
 * elapsed_ms == NULL: the launches are only enqueued (no events, no synchronisation) -- for a caller that drives
 * several handles on several streams (shards of one batch that advance independently) and times them itself.  */
OS2R_API int os2r_bench_steps(Os2rSim* sim, int nsteps, void* stream, float* elapsed_ms);
/* The same for `count` handles on `count` streams -- the shards of one batch --, enqueue only: step k of every shard is
 * enqueued before step k + 1 of any (round robin), so that all the streams start together.                        */
OS2R_API int os2r_bench_steps_multi(Os2rSim* const* sims, void* const* streams, int count, int nsteps);

/* Work counters (measurement support, bench.py's roofline): while a buffer of OS2R_NUM_WORK_COUNTERS uint64 (device
 * memory, zeroed by the caller) is set, os2r_step launches the counting variant of the step kernel -- the same
 * arithmetic, bit for bit -- whose waves add the work they did to it: [0] wave x physics iterations, [1] bodies whose
 * candidate scan ran, [2] bodies whose contact rows were set up, [3] phase-2 sweeps x bodies they covered,
 * [4] phase-2 sweeps executed, [5] (environment, body) contacts, [6] phase-2 sweeps x environments still live in
 * them, [7] wave x iterations that evaluated sin/cos in full, [8] exact free-set solves executed by waves, [9] exact
 * solves x environments that took part.  Counting variants exist for the compiled-in robots
 * with ground contact, the default sweep counts and a reference task layout (OS2R_ERR_INVALID otherwise).
 * NULL switches counting off.                                                                                   */
#define OS2R_NUM_WORK_COUNTERS 10
OS2R_API int os2r_set_work_counters(Os2rSim* sim, uint64_t* counters_dev);

/* Done reasons (replaces the debug line that names the observation which caused a reset,
 * gym_os2r/tasks/monopod.py:288-296): while a buffer of num_envs uint16 (device memory) is set, every os2r_step
 * writes per environment which observation slots were outside the reset space at the end of the step -- bit d:
 * slot d of the task's observation layout (a non-finite value counts) -- i.e. what set bit0 of `done`; 0 for an
 * environment that is not done or only truncated.  NULL switches it off.                                       */
OS2R_API int os2r_set_done_reasons(Os2rSim* sim, uint16_t* reason_dev);

/* Done mask (ABI 5): while a buffer of num_envs uint8 (device memory) is set, every os2r_step also writes 1 where `done` is
 * non-zero and 0 elsewhere -- the boolean `done` that GazeboRuntime.step returns (gym_os2r/runtimes/gazebo_runtime.py:91-97)
 * and SubprocVecEnv stacks (common/vec_env/subproc_vec_env.py:119-123), so a host binding needs no kernel of its own to turn
 * the flag bits into it.  A binding that hands out a fresh array per step sets the pointer before each call (a pointer store).
 * Not written by os2r_rollout.  NULL switches it off.                                                                     */
OS2R_API int os2r_set_done_mask(Os2rSim* sim, uint8_t* mask_dev);

OS2R_API const char* os2r_last_error(Os2rSim* sim); /* sim == NULL: error of the last failed create */

#ifdef __cplusplus
}
#endif
#endif /* OS2R_H_ */
