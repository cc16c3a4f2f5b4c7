"""Diagnostic: env-steps/s through the Python surfaces users call (HipRuntime.step, the randomizer wrapper, HipVecEnv)
against the bare C-ABI loop on the same workload (C4: free_hip, contact, domain randomisation), and the host's own cost per
call with the GPU idle (a batch of 64 environments: the launch is ~40 us of one wave, so the loop is host-bound and what
is timed is Python + ctypes + torch allocations + the HIP runtime's enqueue).

  python tools/dbg/host_overhead.py [N] [steps]      -> profiles/r05_host_surface.txt is this script's output at N = 65536

Replaces the drop-in surface of gym_os2r/runtimes/gazebo_runtime.py:65-97 (GazeboRuntime.step)."""
import functools
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

from gym_os2r_amd.common import make_env_from_id, make_mp_envs
from gym_os2r_amd.randomizers.monopod import MonopodEnvRandomizer

ENV_ID = "Monopod-hop-v1"   # free_hip model; with the randomizer wrapper this is the C4 physics (contact + DR + random resets)


ACTS = {}


def timed(fn, steps, warm=30):
    # a fresh U(-1, 1) action tensor for EVERY step, as a policy would hand over, generated before the clock starts (views of one
    # block: no policy kernels in the way).  Holding one action for hundreds of steps is another workload (513 us per step at
    # 65 536 environments against 147: the robots are driven into the ground), and so is cycling through a few dozen tensors
    # (the regime drifts by +-5 %: the first version of this table compared surfaces across that drift)
    a = ACTS["make"](steps + warm)
    it = iter(a)
    ACTS["next"] = lambda: next(it)
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    t_host = time.perf_counter() - t0          # the host is done enqueueing here
    torch.cuda.synchronize()
    return time.perf_counter() - t0, t_host


def surfaces(n, steps, preroll):
    """-> list of (label, env-steps/s, us per step wall, us per step of host time until the last enqueue returned)"""
    out = []
    make_env = functools.partial(make_env_from_id, env_id=ENV_ID, num_envs=n, max_episode_steps=100_000)
    env = MonopodEnvRandomizer(env=make_env)
    env.seed(42)
    obs = env.reset()
    rt = env.unwrapped
    sim = rt.sim
    sim.bench_steps(preroll)                   # the stationary regime (every robot on the ground), as bench.py
    ACTS["make"] = lambda m: torch.rand(m, n, 2, dtype=obs.dtype, device=obs.device) * 2 - 1

    def act():
        return ACTS["next"]()

    ms = sim.bench_steps(steps)
    out.append(("bare C-ABI loop, os2r_bench_steps (device RNG actions)", n * steps / (ms * 1e-3), ms / steps * 1e3, 0.0))

    o, r, d, t = (torch.empty(n, sim.D, dtype=sim.dtype, device=sim.device), torch.empty(n, dtype=sim.dtype, device=sim.device),
                  torch.empty(n, dtype=torch.uint8, device=sim.device), torch.empty(n, sim.D, dtype=sim.dtype, device=sim.device))
    w, h = timed(lambda: sim.step_into(act(), o, r, d, t), steps)
    out.append(("HipSim.step_into (ctypes, caller's buffers)", n * steps / w, w / steps * 1e6, h / steps * 1e6))
    w, h = timed(lambda: sim.step(act()), steps)
    out.append(("HipSim.step (ctypes, fresh output tensors)", n * steps / w, w / steps * 1e6, h / steps * 1e6))
    w, h = timed(lambda: rt.step(act()), steps)
    out.append(("HipRuntime.step (device action tensor)", n * steps / w, w / steps * 1e6, h / steps * 1e6))
    w, h = timed(lambda: env.step(act()), steps)
    out.append(("randomizer wrapper .step", n * steps / w, w / steps * 1e6, h / steps * 1e6))

    def with_policy():
        a = torch.rand(n, 2, dtype=obs.dtype, device=obs.device) * 2 - 1     # a 'policy' on the device: three small kernels per step
        _, _, _, info = env.step(a)
        _ = info["terminal_observation"]
    w, h = timed(with_policy, steps)
    out.append(("wrapper .step + a torch 'policy' each step", n * steps / w, w / steps * 1e6, h / steps * 1e6))
    env.close()

    venv = make_mp_envs(ENV_ID, n, 42, MonopodEnvRandomizer, max_episode_steps=100_000)
    venv.reset()
    venv.unwrapped.sim.bench_steps(preroll)
    w, h = timed(lambda: venv.step(act()), steps)
    out.append(("HipVecEnv.step (make_mp_envs)", n * steps / w, w / steps * 1e6, h / steps * 1e6))

    def async_wait():
        venv.step_async(act())
        venv.step_wait()
    w, h = timed(async_wait, steps)
    out.append(("HipVecEnv.step_async + step_wait", n * steps / w, w / steps * 1e6, h / steps * 1e6))
    venv.close()
    return out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    print(f"# {torch.cuda.get_device_name(0)}; torch {torch.__version__}; binding {os.environ.get('OS2R_BINDING', 'ctypes')}")
    for nn, pre in ((n, 1000), (64, 200)):
        print(f"\n== {nn} environments ({'GPU-bound if the surface is good' if nn > 4096 else 'GPU idle: host cost per call'}) ==")
        print(f"{'surface':62s} {'M env-steps/s':>14s} {'us/step':>9s} {'host us/call':>13s}")
        rows = surfaces(nn, steps, pre)
        base = rows[0][1]
        for label, v, us, host in rows:
            print(f"{label:62s} {v / 1e6:14.2f} {us:9.1f} {host:13.1f}   ({v / base:5.3f} of the C loop)")


if __name__ == "__main__":
    main()
