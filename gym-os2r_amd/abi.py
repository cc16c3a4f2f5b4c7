"""ctypes mirror of include/os2r.h (struct layouts, enums) and dict -> struct helpers.

Pure host-side plumbing: no compute happens here.  The same struct types are
what the tests hand to the CPU oracle, so one Python description of a model /
task drives both sides of a parity check.
"""
from __future__ import annotations

import ctypes as C
from typing import Mapping, Optional

ABI_VERSION = 5
MAX_DOF = 5
MAX_CAND = 192
MAX_OBS = 12
MAX_RESET_POSES = 8

OK, ERR_INVALID, ERR_HIP, ERR_NO_DEVICE, ERR_ALLOC = 0, 1, 2, 3, 4
F32, F64 = 0, 1

(OBS_POS_NORM, OBS_POS_PERIODIC_NORM, OBS_VEL_TANH, OBS_TORQUE_NORM,
 OBS_POS_RAW, OBS_POS_PERIODIC_RAW, OBS_VEL_RAW, OBS_TORQUE_RAW) = range(8)

(REWARD_BALANCING_V1, REWARD_BALANCING_V2, REWARD_BALANCING_V3, REWARD_STANDING_V1,
 REWARD_HOPPING_V1, REWARD_STRAIGHT_V1) = range(6)

RESET_FIXED, RESET_RANDOM = 0, 1

PARAM_MASS_SCALE, PARAM_DAMPING, PARAM_FRICTION, PARAM_MU, PARAM_GRAVITY = range(5)

DONE_BIT, TRUNCATED_BIT, NONFINITE_BIT = 1, 2, 4

# solver defaults (config_struct): sweeps only / with the exact finish
DEFAULT_PGS_ITERS, DEFAULT_PGS_EXACT = 20, 12
# normal sweeps that fix the friction box (StdSolver::kNormalIters of csrc/os2r_device.hpp: the kernels compiled for the default
# settings).  Two since the end of round 5 (three before): after two sweeps the box-fixing normal impulses are within 6e-4 of the
# converged normal-only solve -- a hundredth of what the fixed box itself is off (6 %: the normal impulses the iteration ends with) --,
# phase 2 needs the same solves, the closed loop is unchanged (docs/studies/round5_solver.md 9)
DEFAULT_PGS_NORMAL_ITERS = 2


def default_pgs_iters_exact(nq: int) -> int:
    """Sweep cap of the exact finish (csrc/os2r_device.hpp sweep_cap_base + kExactRounds): 14 for the 5-dof robot, 12 for the
    smaller ones -- the three sweeps before the first check and the re-test sweeps together.  The kernels built for these
    settings have compile-time loop bounds."""
    return (6 if int(nq) >= 5 else 4) + 8


DEFAULT_PGS_ITERS_EXACT = default_pgs_iters_exact(5)


class Os2rModel(C.Structure):
    _fields_ = [
        ("nq", C.c_int32),
        ("axis", C.c_int32 * MAX_DOF),
        ("rfix", (C.c_double * 9) * MAX_DOF),
        ("rpos", (C.c_double * 3) * MAX_DOF),
        ("mass", C.c_double * MAX_DOF),
        ("com", (C.c_double * 3) * MAX_DOF),
        ("icom", (C.c_double * 6) * MAX_DOF),
        ("damping", C.c_double * MAX_DOF),
        ("friction", C.c_double * MAX_DOF),
        ("mu", C.c_double * MAX_DOF),
        ("act_dof", C.c_int32 * 2),
        ("max_torque", C.c_double * 2),
        ("gravity_z", C.c_double),
        ("ncand", C.c_int32),
        ("cand_body", C.c_int32 * MAX_CAND),
        ("cand_p", (C.c_double * 3) * MAX_CAND),
        ("cand_center", (C.c_double * 3) * MAX_DOF),
        ("cand_radius", C.c_double * MAX_DOF),
    ]


class Os2rTaskSpec(C.Structure):
    _fields_ = [
        ("obs_dim", C.c_int32),
        ("obs_kind", C.c_int32 * MAX_OBS),
        ("obs_src", C.c_int32 * MAX_OBS),
        ("obs_low", C.c_double * MAX_OBS),
        ("obs_high", C.c_double * MAX_OBS),
        ("done_lo", C.c_double * MAX_OBS),
        ("done_hi", C.c_double * MAX_OBS),
        ("reward_id", C.c_int32),
        ("normalized", C.c_int32),
        ("idx_pitch_pos", C.c_int32),
        ("idx_yaw_vel", C.c_int32),
        ("idx_hip_pos", C.c_int32),
        ("idx_knee_pos", C.c_int32),
        ("max_episode_steps", C.c_int32),
        ("reset_mode", C.c_int32),
        ("n_reset_poses", C.c_int32),
        ("reset_pose_id", C.c_int32 * MAX_RESET_POSES),
        ("reset_laying", C.c_int32 * MAX_RESET_POSES),
        ("reset_pitch", C.c_double * MAX_RESET_POSES),
        ("reset_hip", C.c_double * MAX_RESET_POSES),
        ("reset_knee", C.c_double * MAX_RESET_POSES),
        ("reset_simple", C.c_int32),
        ("leg_def", C.c_double * 6),
        ("dof_yaw", C.c_int32),
        ("dof_pitch", C.c_int32),
        ("dof_bc", C.c_int32),
        ("dof_hip", C.c_int32),
        ("dof_knee", C.c_int32),
        ("randomize_params", C.c_int32),
        ("dr_mass_lo", C.c_double), ("dr_mass_hi", C.c_double),
        ("dr_friction_lo", C.c_double), ("dr_friction_hi", C.c_double),
        ("dr_damping_lo", C.c_double), ("dr_damping_hi", C.c_double),
        ("dr_mu_base", C.c_double), ("dr_mu_lo", C.c_double), ("dr_mu_hi", C.c_double),
        ("dr_gravity_mean", C.c_double), ("dr_gravity_std", C.c_double),
        ("gravity_rollouts", C.c_int32), ("reserved_", C.c_int32),
    ]


class Os2rConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("dtype", C.c_int32),
        ("num_envs", C.c_int64),
        ("env_offset", C.c_int64),
        ("seed", C.c_uint64),
        ("device", C.c_int32),
        ("substeps", C.c_int32),
        ("dt", C.c_double),
        ("contact", C.c_int32),
        ("pgs_iters", C.c_int32),
        ("pgs_normal_iters", C.c_int32),
        ("auto_reset", C.c_int32),
        ("erp", C.c_double),
        ("max_erv", C.c_double),
        ("contact_margin", C.c_double),
        ("pgs_tol", C.c_double),
        ("pgs_exact", C.c_int32),
        ("reserved0_", C.c_int32),
        ("model", Os2rModel),
        ("task", Os2rTaskSpec),
    ]


def model_struct(m: Mapping) -> Os2rModel:
    """Compiled-model dict (model_compiler.compile_urdf / assets/models.json) -> Os2rModel."""
    s = Os2rModel()
    nq = int(m["nq"])
    if not 1 <= nq <= MAX_DOF:
        raise ValueError(f"nq={nq} outside 1..{MAX_DOF}")
    if int(m["ncand"]) > MAX_CAND:
        raise ValueError("too many contact candidates")
    s.nq = nq
    for i in range(nq):
        s.axis[i] = int(m["axis"][i])
        for k in range(9):
            s.rfix[i][k] = float(m["rfix"][i][k])
        for k in range(3):
            s.rpos[i][k] = float(m["rpos"][i][k])
            s.com[i][k] = float(m["com"][i][k])
        for k in range(6):
            s.icom[i][k] = float(m["icom"][i][k])
        s.mass[i] = float(m["mass"][i])
        s.damping[i] = float(m["damping"][i])
        s.friction[i] = float(m["friction"][i])
        s.mu[i] = float(m["mu"][i])
    for k in range(2):
        s.act_dof[k] = int(m["act_dof"][k])
        s.max_torque[k] = float(m["max_torque"][k])
    s.gravity_z = float(m["gravity_z"])
    s.ncand = int(m["ncand"])
    last = -1
    for k in range(s.ncand):
        b = int(m["cand_body"][k])
        if b < last:
            raise ValueError("cand_body must be non-decreasing")
        last = b
        s.cand_body[k] = b
        for j in range(3):
            s.cand_p[k][j] = float(m["cand_p"][k][j])
    # bounding sphere of each body's candidates (centre = mean point): lets the kernels skip the
    # scan of a body whose sphere is clear of the ground for every lane of a wave
    import numpy as _np
    pts = _np.array([[s.cand_p[k][j] for j in range(3)] for k in range(s.ncand)]).reshape(-1, 3)
    bodies = _np.array([s.cand_body[k] for k in range(s.ncand)], dtype=int)
    for b in range(nq):
        sel = pts[bodies == b]
        if len(sel):
            c = sel.mean(axis=0)
            r = float(_np.max(_np.linalg.norm(sel - c, axis=1))) * (1 + 1e-12) + 1e-12
            for j in range(3):
                s.cand_center[b][j] = float(c[j])
            s.cand_radius[b] = r
    return s


def task_struct(t: Mapping) -> Os2rTaskSpec:
    """Flat task-spec dict (tasks.MonopodTask.kernel_spec()) -> Os2rTaskSpec."""
    s = Os2rTaskSpec()
    d = int(t["obs_dim"])
    if not 1 <= d <= MAX_OBS:
        raise ValueError("obs_dim out of range")
    s.obs_dim = d
    for i in range(d):
        s.obs_kind[i] = int(t["obs_kind"][i])
        s.obs_src[i] = int(t["obs_src"][i])
        s.obs_low[i] = float(t["obs_low"][i])
        s.obs_high[i] = float(t["obs_high"][i])
        s.done_lo[i] = float(t["done_lo"][i])
        s.done_hi[i] = float(t["done_hi"][i])
    for name in ("reward_id", "normalized", "idx_pitch_pos", "idx_yaw_vel", "idx_hip_pos",
                 "idx_knee_pos", "max_episode_steps", "reset_mode", "reset_simple",
                 "dof_yaw", "dof_pitch", "dof_bc", "dof_hip", "dof_knee", "randomize_params"):
        setattr(s, name, int(t[name]))
    n = len(t["reset_pose_id"])
    if not 1 <= n <= MAX_RESET_POSES:
        raise ValueError("between 1 and %d reset poses required" % MAX_RESET_POSES)
    s.n_reset_poses = n
    for i in range(n):
        s.reset_pose_id[i] = int(t["reset_pose_id"][i])
        s.reset_laying[i] = int(t["reset_laying"][i])
        s.reset_pitch[i] = float(t["reset_pitch"][i])
        s.reset_hip[i] = float(t["reset_hip"][i])
        s.reset_knee[i] = float(t["reset_knee"][i])
    for i in range(6):
        s.leg_def[i] = float(t["leg_def"][i])
    for name in ("dr_mass_lo", "dr_mass_hi", "dr_friction_lo", "dr_friction_hi", "dr_damping_lo",
                 "dr_damping_hi", "dr_mu_base", "dr_mu_lo", "dr_mu_hi", "dr_gravity_mean",
                 "dr_gravity_std"):
        setattr(s, name, float(t[name]))
    s.gravity_rollouts = int(t.get("gravity_rollouts", 0))
    return s


def config_struct(model: Mapping, task: Mapping, *, num_envs: int, dtype: int = F64,
                  env_offset: int = 0, seed: int = 0, device: int = 0, substeps: int = 10,
                  dt: float = 1e-4, contact: bool = True, pgs_iters: Optional[int] = None, pgs_normal_iters: Optional[int] = None,
                  auto_reset: bool = True, erp: float = 0.01, max_erv: float = 1e-3,
                  contact_margin: float = 1e-3, pgs_tol: Optional[float] = None,
                  pgs_exact: Optional[int] = None) -> Os2rConfig:
    """Solver defaults: fp64 -- 2 normal sweeps (`pgs_normal_iters`; None: the default), then at most `pgs_iters` = 14 (12 below five dof) sweeps over all
    rows with the exact finish (`pgs_exact` = 12 free-set solves at most per physics iteration); fp32 -- sweeps only
    (20, checked every 4th: the exact finish needs fp64's headroom for its regularised 5 x 5 solve).  Passing
    `pgs_exact=0, pgs_iters=20` selects the round-1/2 solver in fp64 too."""
    c = Os2rConfig()
    c.abi_version = ABI_VERSION
    c.dtype = int(dtype)
    c.num_envs = int(num_envs)
    c.env_offset = int(env_offset)
    c.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    c.device = int(device)
    c.substeps = int(substeps)
    c.dt = float(dt)
    c.contact = 1 if contact else 0
    if pgs_normal_iters is None:
        pgs_normal_iters = DEFAULT_PGS_NORMAL_ITERS
    if pgs_exact is None:
        pgs_exact = DEFAULT_PGS_EXACT if dtype == F64 and pgs_normal_iters > 0 else 0
    if pgs_iters is None:
        # (the exact finish runs only on the fixed box: with pgs_normal_iters == 0, the coupled pyramid, the library ignores pgs_exact)
        pgs_iters = default_pgs_iters_exact(int(model_struct(model).nq)) if pgs_exact > 0 and pgs_normal_iters > 0 else DEFAULT_PGS_ITERS
    c.pgs_iters = int(pgs_iters)
    c.pgs_exact = int(pgs_exact)
    c.pgs_normal_iters = int(pgs_normal_iters)
    c.auto_reset = 1 if auto_reset else 0
    c.erp = float(erp)
    c.max_erv = float(max_erv)
    c.contact_margin = float(contact_margin)
    # stopping tolerance of the sweeps [J]: far below the solver's truncation error, above the rounding floor of the
    # measure in the handle's arithmetic (fp64: 1e-24; fp32: 1e-13)
    c.pgs_tol = float(pgs_tol) if pgs_tol is not None else (1e-24 if dtype == F64 else 1e-13)
    c.model = model_struct(model)
    c.task = task_struct(task)
    return c
