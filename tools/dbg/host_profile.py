"""Diagnostic: where the host time of one env.step() goes (cProfile over 300 steps)."""
import cProfile
import functools
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

from gym_os2r_amd.common import make_env_from_id
from gym_os2r_amd.randomizers.monopod import MonopodEnvRandomizer

n = 65536
env = MonopodEnvRandomizer(env=functools.partial(make_env_from_id, env_id="Monopod-hop-v1", num_envs=n))
env.seed(42)
obs = env.reset()
act = torch.rand(n, 2, dtype=obs.dtype, device=obs.device) * 2 - 1
for _ in range(20):
    env.step(act)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    env.step(act)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
