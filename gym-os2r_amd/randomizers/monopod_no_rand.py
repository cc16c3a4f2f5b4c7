"""Deterministic reset wrapper (mirror of gym_os2r/randomizers/monopod_no_rand.py:14-101).

The reference removes and re-inserts the Gazebo model on every reset and then writes the reset
pose; here the pose table (one of ``reset_positions`` chosen uniformly, leg angles from the IK of
utils/reset.py, zero velocities) is applied inside the reset / step kernel.
"""
from typing import Callable

from .. import abi


class _EnvWrapper:
    def __init__(self, env: Callable, **kwargs):
        self.env = env(**kwargs) if callable(env) else env

    def __getattr__(self, name):
        return getattr(self.env, name)

    @property
    def unwrapped(self):
        return self.env.unwrapped

    def get_state_info(self, state, actions):
        return self.env.unwrapped.task.get_state_info(state, actions)


class MonopodEnvNoRandomizer(_EnvWrapper):
    def __init__(self, env: Callable, **kwargs):
        super().__init__(env, **kwargs)
        self.env.configure_reset(abi.RESET_FIXED, randomize_params=False)
