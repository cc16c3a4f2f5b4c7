"""SubprocVecEnv-shaped surface over one batched runtime.

Mirrors the methods of ``gym_os2r.common.vec_env.SubprocVecEnv`` (common/vec_env/
subproc_vec_env.py:52-262; base class vec_env.py:32-245) that a trainer calls; the per-process
Pipe protocol is replaced by the batch dimension of the kernel, so ``step_async`` only stores the
actions and ``step_wait`` performs the launch.
"""
from __future__ import annotations

import numpy as np


class HipVecEnv:
    def __init__(self, env):
        self.env = env
        self.num_envs = env.num_envs
        self.observation_space = env.observation_space
        self.action_space = env.action_space
        self.waiting = False
        self.closed = False
        self._actions = None
        self._per_env = {}

    def step_async(self, actions):
        self._actions = actions
        self.waiting = True

    def step_wait(self):
        obs, rew, done, info = self.env.step(self._actions)
        self.waiting = False
        return obs, rew, done, info

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def reset(self):
        return self.env.reset()

    def seed(self, seed=None):
        """SubprocVecEnv seeds worker r with seed + r (subproc_vec_env.py:160-164); here every random stream is keyed
        by (seed, global env index), so one seed gives every environment its own stream: -> [seed] * num_envs."""
        out = self.env.seed(seed)
        s0 = out[0] if isinstance(out, (list, tuple)) and out else out
        return [s0 for _ in range(self.num_envs)]

    def close(self):
        if not self.closed:
            self.env.close()
            self.closed = True

    def get_state_info(self, state, actions):
        """Per-env (reward, done) for given observations and action histories, on the host."""
        state = np.asarray(state, dtype=np.float64)
        if state.ndim == 1:
            return self.env.get_state_info(state, actions)
        return [self.env.get_state_info(s, a) for s, a in zip(state, actions)]

    # One batched runtime stands for all the environments: an attribute of it (task, spaces, reward class ...)
    # is shared by construction, where SubprocVecEnv keeps one copy per worker process
    # (common/vec_env/subproc_vec_env.py:125-214).  `indices` therefore selects how many answers come back;
    # a per-environment value -- a sequence with one entry per environment of the batch, e.g. what
    # `set_attr` stored for a subset -- is indexed by it.
    def get_attr(self, attr_name, indices=None):
        idx = self._indices(indices)
        per_env = self._per_env.get(attr_name)
        shared = getattr(self.env, attr_name) if hasattr(self.env, attr_name) else getattr(self.env.unwrapped, attr_name)
        if per_env is None:
            return [shared for _ in idx]
        return [per_env.get(i, shared) for i in idx]

    def set_attr(self, attr_name, value, indices=None):
        """All environments (indices None): the attribute of the shared runtime is set.  A subset: the value is
        recorded for those environments only and returned by `get_attr` for them; attributes that the kernel
        reads per environment have their own per-environment interface (`HipSim.set_params`, `set_state`)."""
        if indices is None:
            setattr(self.env.unwrapped, attr_name, value)
            self._per_env.pop(attr_name, None)
            return
        slot = self._per_env.setdefault(attr_name, {})
        for i in self._indices(indices):
            slot[i] = value

    def env_method(self, method_name, *args, indices=None, **kwargs):
        idx = self._indices(indices)
        fn = getattr(self.env, method_name)
        result = fn(*args, **kwargs)                  # one call serves the batch
        return [result for _ in idx]

    def get_images(self):
        raise NotImplementedError("headless stepper: no rendering")

    def render(self, mode="human"):
        return None

    def _indices(self, indices):
        if indices is None:
            return list(range(self.num_envs))
        if isinstance(indices, (int, np.integer)):
            indices = [int(indices)]
        idx = [int(i) for i in indices]
        if any(i < 0 or i >= self.num_envs for i in idx):
            raise IndexError(f"env index out of range for {self.num_envs} environments")
        return idx

    @property
    def unwrapped(self):
        return self.env.unwrapped
