#!/usr/bin/env python3
"""The reference's examples/multiprocessing_epochs.py shape -- `make_mp_envs(env_id, nenvs, seed, randomizer)` and
a `step_async / step_wait` loop -- on the HIP runtime: the 'workers' are lanes of one GPU batch.

  python examples/vec_env_epochs.py [--envs 64] [--steps 500]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from gym_os2r_amd.common import make_mp_envs
from gym_os2r_amd.randomizers.monopod import MonopodEnvRandomizer


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=64)
    ap.add_argument("--steps", type=int, default=500)
    args = ap.parse_args()
    envs = make_mp_envs("Monopod-balance-v2", args.envs, seed=0, randomizer=MonopodEnvRandomizer, max_episode_steps=200)
    obs = envs.reset()
    rng = np.random.default_rng(0)
    t0, episodes = time.time(), 0
    for _ in range(args.steps):
        envs.step_async(rng.uniform(-1, 1, (args.envs, 2)))          # host actions are accepted as well
        obs, rewards, dones, info = envs.step_wait()
        episodes += int(dones.sum())
    dt = time.time() - t0
    infos = info.as_list(envs.unwrapped.pose_names)                   # per-env dicts, as a SubprocVecEnv returns them
    print(f"{args.envs} envs x {args.steps} steps in {dt:.2f} s; {episodes} episodes ended; info[0] = {infos[0]}")
    envs.close()


if __name__ == "__main__":
    main()
