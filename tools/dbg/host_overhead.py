"""Diagnostic: env-steps/s through the Python surfaces (HipRuntime.step, the randomizer wrapper, HipVecEnv)
against the bare C-ABI loop, same workload."""
import functools
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

import gym_os2r_amd as g
from gym_os2r_amd.common import make_env_from_id
from gym_os2r_amd.randomizers.monopod import MonopodEnvRandomizer


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    steps = 300
    make_env = functools.partial(make_env_from_id, env_id="Monopod-hop-v1", num_envs=n)   # free_hip model
    env = MonopodEnvRandomizer(env=make_env)
    env.seed(42)
    obs = env.reset()
    act = torch.rand(n, 2, dtype=obs.dtype, device=obs.device) * 2 - 1
    for _ in range(20):
        env.step(act)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        obs, rew, done, info = env.step(act)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"wrapper.step with a device action tensor: {n * steps / dt / 1e6:.1f} M env-steps/s ({dt / steps * 1e6:.0f} us per step)")
    t0 = time.perf_counter()
    for _ in range(steps):
        a = torch.rand(n, 2, dtype=obs.dtype, device=obs.device) * 2 - 1       # a 'policy' on the device
        obs, rew, done, info = env.step(a)
        _ = info["terminal_observation"] if "terminal_observation" in info else None
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"with a torch 'policy' each step:          {n * steps / dt / 1e6:.1f} M env-steps/s ({dt / steps * 1e6:.0f} us per step)")
    sim = env.unwrapped.sim if hasattr(env, "unwrapped") else env.env.sim
    ms = sim.bench_steps(steps)
    print(f"bare C-ABI loop (device RNG actions):     {n * steps / (ms * 1e-3) / 1e6:.1f} M env-steps/s ({ms / steps * 1e3:.0f} us per step)")


if __name__ == "__main__":
    main()
