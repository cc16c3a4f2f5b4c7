"""Reward classes with the reference's names and call signature, plus the id the kernels use.

Host-side mirror of ``gym_os2r.rewards`` (gym_os2r/rewards/__init__.py:9-207) and
``gym_os2r.rewards.rewards_utils.tolerance`` (rewards_utils.py:76-122).  On the hot
path the reward is evaluated inside the step kernel (``kernel_id`` selects the
formula); the numpy methods here serve ``get_state_info`` / ``calculate_reward``
calls made from Python with an observation in hand, exactly as the reference's
task does (gym_os2r/tasks/monopod.py:331-366).
"""
from __future__ import annotations

import numpy as np

from . import abi

_DEFAULT_VALUE_AT_MARGIN = 0.1


def _sigmoids(x, value_at_1, sigmoid):
    if sigmoid in ("cosine", "linear", "quadratic"):
        if not 0 <= value_at_1 < 1:
            raise ValueError(f"`value_at_1` must be nonnegative and smaller than 1, got {value_at_1}.")
    elif not 0 < value_at_1 < 1:
        raise ValueError(f"`value_at_1` must be strictly between 0 and 1, got {value_at_1}.")
    if sigmoid == "gaussian":
        return np.exp(-0.5 * (x * np.sqrt(-2 * np.log(value_at_1))) ** 2)
    if sigmoid == "hyperbolic":
        return 1 / np.cosh(x * np.arccosh(1 / value_at_1))
    if sigmoid == "long_tail":
        return 1 / ((x * np.sqrt(1 / value_at_1 - 1)) ** 2 + 1)
    if sigmoid == "reciprocal":
        return 1 / (abs(x) * (1 / value_at_1 - 1) + 1)
    if sigmoid == "cosine":
        sx = x * (np.arccos(2 * value_at_1 - 1) / np.pi)
        with np.errstate(invalid="ignore"):
            return np.where(abs(sx) < 1, (1 + np.cos(np.pi * sx)) / 2, 0.0)
    if sigmoid == "linear":
        sx = x * (1 - value_at_1)
        return np.where(abs(sx) < 1, 1 - sx, 0.0)
    if sigmoid == "quadratic":
        sx = x * np.sqrt(1 - value_at_1)
        return np.where(abs(sx) < 1, 1 - sx ** 2, 0.0)
    if sigmoid == "tanh_squared":
        return 1 - np.tanh(x * np.arctanh(np.sqrt(1 - value_at_1))) ** 2
    raise ValueError(f"Unknown sigmoid type {sigmoid!r}.")


def tolerance(x, bounds=(0.0, 0.0), margin=0.0, sigmoid="gaussian",
              value_at_margin=_DEFAULT_VALUE_AT_MARGIN):
    """1 inside ``bounds``, decaying to ``value_at_margin`` at distance ``margin`` outside."""
    lower, upper = bounds
    if lower > upper:
        raise ValueError("Lower bound must be <= upper bound.")
    if margin < 0:
        raise ValueError("`margin` must be non-negative.")
    in_bounds = np.logical_and(lower <= x, x <= upper)
    if margin == 0:
        value = np.where(in_bounds, 1.0, 0.0)
    else:
        d = np.where(x < lower, lower - x, x - upper) / margin
        value = np.where(in_bounds, 1.0, _sigmoids(d, value_at_margin, sigmoid))
    return float(value) if np.isscalar(x) else value


class RewardBase:
    """Base class: ``observation_index`` maps '<joint>_pos'/'<joint>_vel' to obs slots."""

    kernel_id = None  # subclasses evaluated in-kernel set one of abi.REWARD_*

    def __init__(self, observation_index: dict, normalized: bool):
        self.observation_index = observation_index
        self.normalized = normalized
        self.supported_task_modes = []
        self._all_task_modes = ["free_hip", "fixed_hip", "fixed", "simple",
                                "fixed_hip_torque", "fixed_hip_simple"]

    def calculate_reward(self, obs, actions):
        raise NotImplementedError

    def is_task_supported(self, task_mode: str) -> bool:
        return task_mode in self.supported_task_modes

    def get_supported_task_modes(self):
        return self.supported_task_modes

    def _height(self):
        return 0.11 / 1.57 * self.normalized + 0.11 * (1 - self.normalized)


_LEG_MODES = ["free_hip", "fixed_hip", "fixed_hip_torque", "fixed_hip_simple", "fixed"]


class BalancingV1(RewardBase):
    kernel_id = abi.REWARD_BALANCING_V1

    def __init__(self, observation_index, normalized):
        super().__init__(observation_index, normalized)
        self.supported_task_modes = list(_LEG_MODES)

    def calculate_reward(self, obs, actions):
        h = self._height()
        bp = obs[self.observation_index["planarizer_pitch_joint_pos"]]
        return tolerance(bp, (h, 4 * h))


class BalancingV2(RewardBase):
    kernel_id = abi.REWARD_BALANCING_V2

    def __init__(self, observation_index, normalized):
        super().__init__(observation_index, normalized)
        self.supported_task_modes = list(_LEG_MODES)

    def calculate_reward(self, obs, actions):
        action = np.asarray(actions[0], dtype=np.float64)
        h = self._height()
        bp = obs[self.observation_index["planarizer_pitch_joint_pos"]]
        balancing = tolerance(bp, (h, 4 * h))
        small_control = tolerance(action, margin=1, value_at_margin=0.4, sigmoid="quadratic")
        return balancing * np.prod(small_control)


class BalancingV3(RewardBase):
    kernel_id = abi.REWARD_BALANCING_V3

    def __init__(self, observation_index, normalized):
        super().__init__(observation_index, normalized)
        self.supported_task_modes = list(_LEG_MODES)

    def calculate_reward(self, obs, actions):
        action = np.asarray(actions[0], dtype=np.float64)
        action_old = np.asarray(actions[1], dtype=np.float64)
        h = self._height()
        bp = obs[self.observation_index["planarizer_pitch_joint_pos"]]
        balancing = tolerance(bp, (h, 4 * h), margin=0.01, sigmoid="long_tail")
        small_delta = tolerance(action - action_old, margin=1, value_at_margin=0.1,
                                sigmoid="quadratic")
        return balancing * np.prod(small_delta)


class StandingV1(RewardBase):
    kernel_id = abi.REWARD_STANDING_V1

    def __init__(self, observation_index, normalized):
        super().__init__(observation_index, normalized)
        self.supported_task_modes = list(_LEG_MODES)

    def calculate_reward(self, obs, actions):
        h = self._height()
        bp = obs[self.observation_index["planarizer_pitch_joint_pos"]]
        return tolerance(bp, (h, 4 * h))


class HoppingV1(RewardBase):
    kernel_id = abi.REWARD_HOPPING_V1

    def __init__(self, observation_index, normalized):
        super().__init__(observation_index, normalized)
        self.supported_task_modes = list(_LEG_MODES)

    def calculate_reward(self, obs, actions):
        action = np.asarray(actions[0], dtype=np.float64)
        action_old = np.asarray(actions[1], dtype=np.float64)
        h = self._height()
        bp = obs[self.observation_index["planarizer_pitch_joint_pos"]]
        balancing = tolerance(bp, (h, 4 * h))
        small_delta = tolerance(action - action_old, margin=0.1, value_at_margin=0,
                                sigmoid="quadratic")
        h_vel = obs[self.observation_index["planarizer_yaw_joint_vel"]]
        move = tolerance(h_vel, bounds=(0.25, 0.3), margin=0.15, value_at_margin=0.1,
                         sigmoid="tanh_squared")
        return balancing * np.prod(small_delta) * move


class StraightV1(RewardBase):
    kernel_id = abi.REWARD_STRAIGHT_V1

    def __init__(self, observation_index, normalized):
        super().__init__(observation_index, normalized)
        self.supported_task_modes = ["simple"]

    def calculate_reward(self, obs, actions):
        action = np.asarray(actions[0], dtype=np.float64)
        small_control = tolerance(action / 20, margin=1, value_at_margin=0,
                                  sigmoid="quadratic").mean()
        small_control = (4 + small_control) / 5
        hip = obs[self.observation_index["hip_joint_pos"]]
        knee = obs[self.observation_index["knee_joint_pos"]]
        hip_reward = tolerance(hip, bounds=(0, 0), margin=1, sigmoid="linear")
        knee_reward = tolerance(knee, bounds=(0, 0), margin=1, sigmoid="linear")
        return hip_reward * knee_reward * small_control


__all__ = ["RewardBase", "BalancingV1", "BalancingV2", "BalancingV3", "StandingV1", "HoppingV1",
           "StraightV1", "tolerance"]
