"""Environment ids of the reference (gym_os2r/__init__.py:16-128) and the factories around them
(gym_os2r/common/__init__.py:12-53)."""
from __future__ import annotations

import numpy as np

from . import rewards
from .tasks import monopod, monopod_no_norm

_MAX_FLOAT = float(np.finfo(np.float32).max)


def _spec(task_cls, mode, reward, resets, max_steps):
    return {"max_episode_steps": max_steps,
            "kwargs": {"task_cls": task_cls, "agent_rate": 1000, "physics_rate": 10000,
                       "real_time_factor": _MAX_FLOAT, "task_mode": mode, "reward_class": reward,
                       "reset_positions": resets}}


_ALL_POSES = ["stand", "half_stand", "ground", "lay", "float"]
REGISTRY = {
    "Monopod-stand-v1": _spec(monopod.MonopodTask, "fixed_hip", rewards.StandingV1, ["ground"], 100_000),
    "Monopod-balance-v1": _spec(monopod.MonopodTask, "fixed_hip_simple", rewards.BalancingV1, ["stand"], 100_000),
    "Monopod-balance-v2": _spec(monopod.MonopodTask, "fixed_hip_simple", rewards.BalancingV2, ["stand"], 100_000),
    "Monopod-balance-v3": _spec(monopod.MonopodTask, "fixed_hip_simple", rewards.BalancingV2, _ALL_POSES, 10_000),
    "Monopod-nonorm-balance-v1": _spec(monopod_no_norm.MonopodTask, "fixed_hip_simple", rewards.BalancingV1, ["stand"], 100_000),
    "Monopod-nonorm-balance-v2": _spec(monopod_no_norm.MonopodTask, "fixed_hip_simple", rewards.BalancingV2, ["stand"], 100_000),
    "Monopod-nonorm-balance-v3": _spec(monopod_no_norm.MonopodTask, "fixed_hip_simple", rewards.BalancingV2, _ALL_POSES, 10_000),
    "Monopod-hop-v1": _spec(monopod.MonopodTask, "free_hip", rewards.HoppingV1, ["stand"], 100_000),
    "Monopod-simple-v1": _spec(monopod.MonopodTask, "simple", rewards.StraightV1, ["stand"], 100_000),
}


def make(env_id: str, num_envs: int = 1, **kwargs):
    """``gym.make(env_id, **kwargs)`` for the batched runtime; kwargs override the registered ones."""
    from .runtimes import HipRuntime
    if env_id not in REGISTRY:
        raise KeyError(f"No registered env with id: {env_id}")
    spec = REGISTRY[env_id]
    kw = dict(spec["kwargs"])
    kw.setdefault("max_episode_steps", spec["max_episode_steps"])   # gym's TimeLimit wrapper
    kw.update(kwargs)
    return HipRuntime(num_envs=num_envs, **kw)
