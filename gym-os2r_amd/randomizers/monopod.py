"""Domain-randomising reset wrapper (mirror of gym_os2r/randomizers/monopod.py:363-385).

Per reset (randomizers/monopod.py:89-128,182-215): boom pitch x U(0.8,1.2), leg angles from the IK
plus |N(0,0.2)| perturbations, random mirroring, yaw ~ U(-0.2,0.2); every link mass x U(0.8,1.2),
joint friction ~ U(0.01,0.05), joint damping x U(0.8,1.2), link contact mu = 0.33 x U(0.8,1.2);
gravity ~ N(-9.8, 0.2) once per environment (:56-61).  The reference rewrites and re-parses an SDF
file for this; here the samples come from the counter RNG inside the reset / step kernel and land
in the per-env parameter arrays.
"""
from typing import Callable

from .. import abi
from .monopod_no_rand import _EnvWrapper


class MonopodEnvRandomizer(_EnvWrapper):
    def __init__(self, env: Callable, num_physics_rollouts: int = 0, **kwargs):
        super().__init__(env, **kwargs)
        # The reference draws gravity in randomize_physics, which runs when the simulator is (re)created: once per
        # process with the default 0, after every `num_physics_rollouts` rollouts otherwise
        # (randomizers/monopod.py:36-41,56-61,371).  Here: per environment, in the reset of the step kernel.
        if int(num_physics_rollouts) < 0:
            raise ValueError("num_physics_rollouts must be >= 0")
        self.num_physics_rollouts = int(num_physics_rollouts)
        self.env.configure_reset(abi.RESET_RANDOM, randomize_params=True, gravity_rollouts=self.num_physics_rollouts)
