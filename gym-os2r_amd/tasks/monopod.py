"""Monopod task definition: spaces, observation layout, reward and done rules.

Host-side mirror of ``gym_os2r.tasks.monopod.MonopodTask`` (tasks/monopod.py:15-374).
The reference task talks to one simulated model through ScenarIO; here the task
is the *specification* of the epilogue the step kernel runs for every
environment: ``kernel_spec()`` flattens what ``create_spaces()`` derives from the
settings tree (observation mask, periodic joints, limits, reward) into the
``Os2rTaskSpec`` of include/os2r.h.  The numpy methods (``calculate_reward``,
``get_state_info``) keep the reference's call signatures for Python callers.
"""
from __future__ import annotations

import warnings
from collections import deque
from typing import Deque, Dict, Tuple

import numpy as np

from .. import abi
from ..config import SettingsConfig
from ..spaces import Box

_EPS = np.finfo(float).eps


# --- exact thresholds ---------------------------------------------------------------
def _ordered(x: float) -> int:
    """Map a float64 to an int64 whose order matches the float order."""
    i = np.array([x], dtype=np.float64).view(np.int64)[0]
    return int(i) if i >= 0 else int(-(i & 0x7FFFFFFFFFFFFFFF))


def _unordered(i: int) -> float:
    if i >= 0:
        return float(np.array([i], dtype=np.int64).view(np.float64)[0])
    return float(np.array([(-i) | (1 << 63)], dtype=np.uint64).view(np.float64)[0])


def _last_true(pred, lo: float, hi: float) -> float:
    """Largest float64 x in [lo, hi] with pred(x), for pred true at lo and monotone."""
    if pred(hi):
        return hi
    a, b = _ordered(lo), _ordered(hi)
    while b - a > 1:
        m = (a + b) // 2
        if pred(_unordered(m)):
            a = m
        else:
            b = m
    return _unordered(a)


def _first_true(pred, lo: float, hi: float) -> float:
    """Smallest float64 x in [lo, hi] with pred(x), for pred true at hi and monotone."""
    return -_last_true(lambda x: pred(-x), -hi, -lo)


class MonopodTask:
    """Task mode + reward + reset positions -> spaces and the kernel epilogue spec."""

    normalized = True

    SUPPORTED_TASK_MODES = ("free_hip", "fixed_hip", "fixed", "fixed_hip_torque", "simple", "fixed_hip_simple")
    REQUIRED_KWARGS = ("task_mode", "reward_class", "reset_positions")
    ACTION_HISTORY_LEN = 10

    def __init__(self, agent_rate: float, **kwargs):
        """Same contract as the reference constructor (tasks/monopod.py:37-103): the three required kwargs, every
        kwarg becomes an attribute, an optional ``config`` object replaces the packaged settings, and a bad task
        mode / reset position / missing kwarg raises RuntimeError."""
        self._validate_kwargs(kwargs)
        self.agent_rate = agent_rate
        self.supported_task_modes = list(self.SUPPORTED_TASK_MODES)
        self.cfg = kwargs.get("config") or SettingsConfig()
        # what the kernel's epilogue is derived from: the settings subtree of this task mode
        self.spaces_definition = self._spaces_of(self.cfg, kwargs["task_mode"], kwargs["reset_positions"])
        sd = self.spaces_definition
        self.action_names = list(sd["action"])
        self.joint_names = list(sd["observation"])
        self.observing_measured_torque = sd["observing_measured_torque"]
        self.observation_name_mask = sd["observation_mask"]
        self.observation_index: Dict[str, int] = {}
        # filled by create_spaces() / the runtime
        self.action_space = self.observation_space = self.reset_space = None
        self.model = self.model_name = self.current_reset_orientation = None
        self.np_random = np.random.default_rng()
        zero = np.zeros(len(self.action_names))
        self.action_history: Deque = deque((zero.copy() for _ in range(self.ACTION_HISTORY_LEN)), maxlen=self.ACTION_HISTORY_LEN)
        for name, value in kwargs.items():            # task_mode, reward_class, reset_positions, and any override
            setattr(self, name, value)

    @classmethod
    def _validate_kwargs(cls, kwargs):
        required = list(cls.REQUIRED_KWARGS)
        missing = [k for k in required if k not in kwargs]
        if missing:
            raise RuntimeError("Missing required kwarg: " + missing[0] + ". We require the following kwargs, " + str(required)
                               + "\n in the MonopodTask class. (These can be specified in env init)")
        if len(kwargs) != len(required):
            warnings.warn("# WARNING: Supplied Kwargs, " + str(kwargs) + " Contains more entries than expected. Required Kwargs are "
                          + str(required) + ". Could be caused by config object.", SyntaxWarning, stacklevel=3)

    @classmethod
    def _spaces_of(cls, cfg, task_mode, reset_positions):
        known_resets = list(cfg.get_config("/resets").keys())
        if not set(reset_positions) <= set(known_resets):
            raise RuntimeError("One or more of the reset positions provided were not in the supported reset positions. "
                               + str(known_resets))
        if task_mode not in cls.SUPPORTED_TASK_MODES:
            raise RuntimeError("task mode " + task_mode + " not supported in monopod environment.")
        try:
            return cfg.get_config("task_modes/" + task_mode + "/spaces")
        except KeyError:
            raise RuntimeError("task mode " + task_mode + " does not contain spaces definition in monopod environment config file.")

    # -----------------------------------------------------------------------------
    def create_spaces(self) -> Tuple[Box, Box]:
        """tasks/monopod.py:105-200 (normalised) / monopod_no_norm.py:105-184."""
        self.max_torques = np.array(list(self.spaces_definition["action"].values()), dtype=np.float64)
        low_act, high_act = np.array([-1, -1]), np.array([1, 1])
        action_space = Box(low=low_act, high=high_act, dtype=np.float64)

        obs_lim = np.array([info["limits"] for info in self.spaces_definition["observation"].values()],
                           dtype=np.float64)
        low = np.concatenate((obs_lim[:, 1], obs_lim[:, 3]))
        high = np.concatenate((obs_lim[:, 0], obs_lim[:, 2]))
        if self.observing_measured_torque:
            low = np.array([*low, *low_act], dtype=np.float64)
            high = np.array([*high, *high_act], dtype=np.float64)

        names = [n + "_pos" for n in self.joint_names] + [n + "_vel" for n in self.joint_names]
        if self.observing_measured_torque:
            names += [n + "_torque" for n in self.action_names]
        self.observation_names_all = names
        self.observation_mask, self.velocities_index, obs_index = [], [], {}
        for obs_i, name in enumerate(names):
            if name not in self.observation_name_mask:
                obs_index[name] = len(self.observation_mask)
                if "_vel" in name:
                    self.velocities_index.append(len(self.observation_mask))
                self.observation_mask.append(obs_i)
        low, high = low[self.observation_mask], high[self.observation_mask]
        self.observation_index = obs_index

        self.periodic_joints = []
        for joint, info in self.spaces_definition["observation"].items():
            if info["periodic_pos"] and joint + "_pos" in obs_index:
                self.periodic_joints.append(obs_index[joint + "_pos"])
        low[self.periodic_joints] = -(np.pi + _EPS)
        high[self.periodic_joints] = np.pi + _EPS

        self.obs_limits = {"high": high.copy(), "low": low.copy()}
        self.mask_inf_obs = np.zeros(len(high), dtype=bool)
        self.mask_inf_obs[self.velocities_index] = True
        if self.normalized:
            low = np.full_like(low, -1.0)
            high = np.full_like(high, 1.0)
        obs_space = Box(low=low, high=high, dtype=np.float64)

        self.reward = self.reward_class(self.observation_index, normalized=self.normalized)
        assert self.reward.is_task_supported(self.task_mode), \
            f"'{self.task_mode}' task mode not supported by reward class '{self.reward}'"
        self.reset_space = Box(low=low + _EPS, high=high - _EPS, dtype=np.float64)
        self.action_space, self.observation_space = action_space, obs_space
        return action_space, obs_space

    # -----------------------------------------------------------------------------
    def normalize_observation(self, raw_masked: np.ndarray) -> np.ndarray:
        """raw (masked) [pos.., vel.., torque..] -> observation (tasks/monopod.py:257-272)."""
        obs = np.array(raw_masked, dtype=np.float64)
        pj = self.periodic_joints
        obs[..., pj] = np.mod(obs[..., pj] + np.pi, 2 * np.pi) - np.pi
        if self.normalized:
            high, low, m = self.obs_limits["high"], self.obs_limits["low"], self.mask_inf_obs
            obs[..., ~m] = 2 * (obs[..., ~m] - low[~m]) / (high[~m] - low[~m]) - 1
            obs[..., m] = np.tanh(0.05 * obs[..., m])
        return obs

    def calculate_reward(self, obs, action):
        return self.reward.calculate_reward(obs, action)

    def get_state_info(self, obs, actions):
        """(reward, done) for a given observation and action history.

        The reference body refers to an undefined name (tasks/monopod.py:348,364);
        the intent -- reward(obs, actions), done = obs outside reset_space -- is kept.
        """
        reward = self.calculate_reward(obs, actions)
        done = not self.reset_space.contains(np.asarray(obs, dtype=np.float64))
        return reward, done

    def get_info(self) -> Dict:
        return {"reset_orientation": self.current_reset_orientation}

    # -----------------------------------------------------------------------------
    def _done_thresholds(self, kind: int, low: float, high: float) -> Tuple[float, float]:
        """Bounds on the pre-map value y such that the reference's test
        ``reset_space.contains(obs)`` (tasks/monopod.py:198,287) holds iff lo <= y <= hi.
        Found by bisection over float64 on the reference's own formula, so the
        equivalence is exact for the monotone maps involved."""
        rlo, rhi = -1.0 + _EPS, 1.0 - _EPS
        if kind in (abi.OBS_POS_NORM, abi.OBS_POS_PERIODIC_NORM, abi.OBS_TORQUE_NORM):
            def n(y):
                return 2 * (np.float64(y) - low) / (high - low) - 1
            mid = 0.5 * (low + high)
            span = abs(high - low)
            hi = _last_true(lambda y: n(y) <= rhi, mid, high + span)
            lo = _first_true(lambda y: n(y) >= rlo, low - span, mid)
            return lo, hi
        if kind == abi.OBS_VEL_TANH:
            hi = _last_true(lambda v: np.tanh(0.05 * np.float64(v)) <= rhi, 0.0, 1e4)
            lo = _first_true(lambda v: np.tanh(0.05 * np.float64(v)) >= rlo, -1e4, 0.0)
            return lo, hi
        # raw kinds: reset_space = Box(low + eps, high - eps) on the value itself
        return float(np.float64(low) + _EPS), float(np.float64(high) - _EPS)

    def kernel_spec(self, model: dict, *, reset_mode: int = abi.RESET_FIXED,
                    randomize_params: bool = False, max_episode_steps: int = 0, gravity_rollouts: int = 0) -> dict:
        """Flatten the task into the ``Os2rTaskSpec`` fields (see include/os2r.h)."""
        if self.observation_space is None:
            self.create_spaces()
        # A custom RewardBase subclass has no in-kernel formula: the kernel then evaluates a
        # placeholder the task mode supports and the runtime recomputes the reward on the host
        # from the observation and action history (slow path, see HipRuntime.step).
        self.host_reward = getattr(self.reward, "kernel_id", None) is None
        kernel_reward = (self.reward.kernel_id if not self.host_reward else
                         abi.REWARD_BALANCING_V1 if "planarizer_pitch_joint_pos" in self.observation_index
                         else abi.REWARD_STRAIGHT_V1)
        if self.host_reward and kernel_reward == abi.REWARD_STRAIGHT_V1 and not (
                "hip_joint_pos" in self.observation_index and "knee_joint_pos" in self.observation_index):
            raise RuntimeError("custom reward classes need the pitch or the hip+knee positions observed")
        dof = {name: i for i, name in enumerate(model["dof_names"])}
        for jn in self.joint_names:
            if jn not in dof:
                raise RuntimeError(f"joint {jn!r} of task mode {self.task_mode!r} is not a movable "
                                   f"joint of model {model['name']!r}")
        nj = len(self.joint_names)
        kinds, srcs, lows, highs, dlo, dhi = [], [], [], [], [], []
        for slot, raw_i in enumerate(self.observation_mask):
            name = self.observation_names_all[raw_i]
            low, high = float(self.obs_limits["low"][slot]), float(self.obs_limits["high"][slot])
            if raw_i < nj:
                periodic = slot in self.periodic_joints
                if self.normalized:
                    kind = abi.OBS_POS_PERIODIC_NORM if periodic else abi.OBS_POS_NORM
                else:
                    kind = abi.OBS_POS_PERIODIC_RAW if periodic else abi.OBS_POS_RAW
                src = dof[self.joint_names[raw_i]]
            elif raw_i < 2 * nj:
                kind = abi.OBS_VEL_TANH if self.normalized else abi.OBS_VEL_RAW
                src = dof[self.joint_names[raw_i - nj]]
            else:
                kind = abi.OBS_TORQUE_NORM if self.normalized else abi.OBS_TORQUE_RAW
                src = raw_i - 2 * nj
            lo_t, hi_t = self._done_thresholds(kind, low, high)
            kinds.append(kind); srcs.append(src); lows.append(low); highs.append(high)
            dlo.append(lo_t); dhi.append(hi_t)
            del name

        resets = self.cfg.get_config("/resets")
        pose_names = list(resets.keys())
        definition = self.cfg.get_config("task_modes/" + self.task_mode + "/definition")
        leg_def = [definition[k] for k in ("upper_leg_length", "lower_leg_length",
                                           "central_pivot_height", "length_boom", "hip_offset",
                                           "clipping_adjust")]
        from ..utils.reset import leg_joint_angles
        pose_id, laying, pitch, hip, knee = [], [], [], [], []
        for pname in self.reset_positions:
            conf = resets[pname]
            pose_id.append(pose_names.index(pname))
            laying.append(1 if conf["laying_down"] else 0)
            pitch.append(float(conf["planarizer_pitch_joint"]))
            if conf["laying_down"]:
                ang = (1.57, 0.0)                      # randomizers/monopod_no_rand.py:76
            else:
                rd = dict(definition)
                rd["planarizer_pitch_joint"] = conf["planarizer_pitch_joint"]
                ang = leg_joint_angles(rd)
            hip.append(float(ang[0])); knee.append(float(ang[1]))

        idx = self.observation_index
        return {
            "obs_dim": len(kinds), "obs_kind": kinds, "obs_src": srcs, "obs_low": lows,
            "obs_high": highs, "done_lo": dlo, "done_hi": dhi,
            "reward_id": int(kernel_reward), "normalized": 1 if self.normalized else 0,
            "idx_pitch_pos": idx.get("planarizer_pitch_joint_pos", -1),
            "idx_yaw_vel": idx.get("planarizer_yaw_joint_vel", -1),
            "idx_hip_pos": idx.get("hip_joint_pos", -1),
            "idx_knee_pos": idx.get("knee_joint_pos", -1),
            "max_episode_steps": int(max_episode_steps),
            "reset_mode": int(reset_mode),
            "reset_pose_id": pose_id, "reset_laying": laying, "reset_pitch": pitch,
            "reset_hip": hip, "reset_knee": knee,
            "reset_simple": 1 if self.task_mode == "simple" else 0,
            "leg_def": leg_def,
            "dof_yaw": dof.get("planarizer_yaw_joint", -1),
            "dof_pitch": dof.get("planarizer_pitch_joint", -1),
            "dof_bc": dof.get("boom_connector_joint", -1),
            "dof_hip": dof.get("hip_joint", -1),
            "dof_knee": dof.get("knee_joint", -1),
            "randomize_params": 1 if randomize_params else 0,
            # gym_os2r/randomizers/monopod.py:182-215 and :58
            "dr_mass_lo": 0.8, "dr_mass_hi": 1.2,
            "dr_friction_lo": 0.01, "dr_friction_hi": 0.05,
            "dr_damping_lo": 0.8, "dr_damping_hi": 1.2,
            "dr_mu_base": 0.33, "dr_mu_lo": 0.8, "dr_mu_hi": 1.2,
            "dr_gravity_mean": -9.8, "dr_gravity_std": 0.2 if reset_mode == abi.RESET_RANDOM else 0.0,
            "gravity_rollouts": int(gravity_rollouts) if reset_mode == abi.RESET_RANDOM else 0,
        }
