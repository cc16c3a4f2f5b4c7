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
        return self.env.seed(seed)

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

    def get_attr(self, attr_name, indices=None):
        n = len(self._indices(indices))
        return [getattr(self.env, attr_name)] * n

    def set_attr(self, attr_name, value, indices=None):
        setattr(self.env.unwrapped, attr_name, value)

    def env_method(self, method_name, *args, indices=None, **kwargs):
        n = len(self._indices(indices))
        return [getattr(self.env, method_name)(*args, **kwargs)] * n

    def get_images(self):
        raise NotImplementedError("headless stepper: no rendering")

    def render(self, mode="human"):
        return None

    def _indices(self, indices):
        if indices is None:
            return list(range(self.num_envs))
        if isinstance(indices, int):
            return [indices]
        return list(indices)

    @property
    def unwrapped(self):
        return self.env.unwrapped
