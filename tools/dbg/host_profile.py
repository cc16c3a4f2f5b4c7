"""Diagnostic: where the host time of one HipRuntime.step() goes.  A batch of 64 environments (the GPU is idle: what is timed is
the host), fresh random actions every step from a pre-generated [K, N, 2] tensor (a view per step: no kernel of a policy in the
way), cProfile over the steps, plus stage-by-stage timings of the pieces step() is made of.
  python tools/dbg/host_profile.py [N] [steps]"""
import cProfile
import functools
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

from gym_os2r_amd.common import make_env_from_id
from gym_os2r_amd.randomizers.monopod import MonopodEnvRandomizer

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
env = MonopodEnvRandomizer(env=functools.partial(make_env_from_id, env_id="Monopod-hop-v1", num_envs=n, max_episode_steps=100_000))
env.seed(42)
obs = env.reset()
rt = env.unwrapped
sim = rt.sim
sim.bench_steps(1000)     # the stationary regime (every robot on the ground), as bench.py
acts = torch.rand(64, n, 2, dtype=obs.dtype, device=obs.device) * 2 - 1


def loop(fn, k=steps):
    for i in range(30):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(k):
        fn(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (t1 - t0) / k * 1e6, (time.perf_counter() - t0) / k * 1e6


o, r, d, t = (torch.empty(n, sim.D, dtype=sim.dtype, device=sim.device), torch.empty(n, dtype=sim.dtype, device=sim.device),
              torch.empty(n, dtype=torch.uint8, device=sim.device), torch.empty(n, sim.D, dtype=sim.dtype, device=sim.device))
sets = [[torch.empty_like(x) for x in (o, r, d, t)] for _ in range(3)]      # rotating output sets: what fresh tensors amount to
flag = torch.zeros(2, dtype=torch.int32).pin_memory()
ev = torch.cuda.Event()
print(f"# {n} environments, {steps} steps; columns: host us per call (enqueue returned) / wall us per call (device drained)")
for label, fn in (
        ("sim.step_into(acts[i])", lambda i: sim.step_into(acts[i & 63], o, r, d, t)),
        ("sim.step_into, three output sets in rotation", lambda i: sim.step_into(acts[i & 63], *sets[i % 3])),
        ("sim.step_into again (one set)", lambda i: sim.step_into(acts[i & 63], o, r, d, t)),
        ("sim.step(acts[i]) (4 torch.empty)", lambda i: sim.step(acts[i & 63])),
        ("sim.step + action_violations_into (4-byte D2H copy)", lambda i: (sim.step(acts[i & 63]), sim.action_violations_into(flag[0:1], clear=False))),
        ("sim.step + event.record", lambda i: (sim.step(acts[i & 63]), ev.record())),
        ("sim.step + (flags != 0)", lambda i: sim.step(acts[i & 63])[2] != 0),
        ("sim.step(want_mask=True)", lambda i: sim.step(acts[i & 63], want_mask=True)),
        ("sim.bench_steps(1) (device RNG actions, scratch outputs)", lambda i: sim.bench_enqueue(1)),
        ("HipRuntime.step(acts[i])", lambda i: rt.step(acts[i & 63])),
        ("wrapper.step(acts[i])", lambda i: env.step(acts[i & 63]))):
    h, w = loop(fn)
    print(f"{label:58s} {h:8.1f} {w:8.1f}")

pr = cProfile.Profile()
pr.enable()
for i in range(steps):
    rt.step(acts[i & 63])
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
