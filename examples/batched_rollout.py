#!/usr/bin/env python3
"""What the reference's examples/fixed_hip.py does -- random-action episodes of Monopod-balance-v1 behind the
randomizer wrapper -- with the batch dimension of the HIP runtime: N environments per call instead of one.

  python examples/batched_rollout.py [--envs 4096] [--steps 2000] [--task-mode fixed_hip]

Everything stays on the GPU: the 'policy' below draws its actions there, `env.step` returns device tensors, and
environments that finish are reset inside the same launch (their last observation is in
`info['terminal_observation']`), exactly like a SubprocVecEnv worker does it in the reference.
"""
import argparse
import functools
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gym_os2r_amd import randomizers
from gym_os2r_amd.common import make_env_from_id


def pipelined(args):
    """A small policy network on shard A while the physics of shard B runs: every shard has its own handle and stream,
    `step_async(a, split=i)` enqueues its env-step and returns, `step_wait(split=i)` hands back its tensors ordered behind
    that launch on the caller's stream.  The results are those of the single batch (the RNG is keyed by the global index)."""
    from gym_os2r_amd.common import make_mp_envs
    vec = make_mp_envs(args.env_id, args.envs, 42, randomizers.monopod.MonopodEnvRandomizer, num_splits=args.splits,
                       task_mode=args.task_mode, max_episode_steps=args.max_episode_steps)
    obs_all = vec.reset()
    dev, dt = obs_all.device, obs_all.dtype
    D = obs_all.shape[1]
    torch.manual_seed(0)
    policy = torch.nn.Sequential(torch.nn.Linear(D, 64), torch.nn.Tanh(), torch.nn.Linear(64, 2), torch.nn.Tanh()).to(dev, dt)
    obs = [obs_all[sl] for sl in vec.split_slices]
    ret = [torch.zeros(sl.stop - sl.start, dtype=dt, device=dev) for sl in vec.split_slices]
    torch.cuda.synchronize()
    t0 = time.time()
    with torch.no_grad():
        for i in range(vec.num_splits):                       # prime the pipeline: every shard gets its first launch
            vec.step_async(policy(obs[i]), split=i)
        for _ in range(args.steps - 1):
            for i in range(vec.num_splits):
                o, r, d, info = vec.step_wait(split=i)        # shard i's results; the other shards' physics keep running
                ret[i].add_(r)
                vec.step_async(policy(o), split=i)            # its next actions, its next launch
        for i in range(vec.num_splits):
            o, r, d, info = vec.step_wait(split=i)
            ret[i].add_(r)
    torch.cuda.synchronize()
    dt_s = time.time() - t0
    total = float(torch.cat(ret).mean())
    print(f"{vec.num_splits} shards, policy and physics pipelined: {args.envs} envs x {args.steps} steps in {dt_s:.2f} s = "
          f"{args.envs * args.steps / dt_s / 1e6:.1f} M env-steps/s; mean return per env {total:.2f}")
    vec.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env-id", default="Monopod-balance-v1")
    ap.add_argument("--task-mode", default="fixed_hip")
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--max-episode-steps", type=int, default=500)
    ap.add_argument("--splits", type=int, default=0,
                    help="pipeline the rollout over that many shards of the batch (HipVecEnv num_splits): while the physics of one "
                         "shard runs on its stream, the policy of the next shard is evaluated -- what step_async / step_wait of the "
                         "reference's SubprocVecEnv are for (common/vec_env/subproc_vec_env.py:114-123)")
    ap.add_argument("--graph", action="store_true",
                    help="capture one loop iteration (policy, env step, bookkeeping) in a HIP graph and replay it: for small "
                         "batches the loop is bound by launches, not by the physics")
    args = ap.parse_args()

    if args.splits > 0:
        return pipelined(args)
    make_env = functools.partial(make_env_from_id, env_id=args.env_id, num_envs=args.envs, task_mode=args.task_mode,
                                 max_episode_steps=args.max_episode_steps)
    env = randomizers.monopod.MonopodEnvRandomizer(env=make_env)
    env.seed(42)
    obs = env.reset()
    returns = torch.zeros(args.envs, dtype=obs.dtype, device=obs.device)
    finished = torch.zeros((), dtype=torch.int64, device=obs.device)
    sum_returns = torch.zeros((), dtype=obs.dtype, device=obs.device)
    sim = env.unwrapped.sim
    # caller-owned buffers: `step_into` is then one kernel launch and nothing else (what a HIP graph can hold)
    actions = torch.zeros(args.envs, 2, dtype=obs.dtype, device=obs.device)
    reward = torch.zeros(args.envs, dtype=obs.dtype, device=obs.device)
    done_u8 = torch.zeros(args.envs, dtype=torch.uint8, device=obs.device)

    def iteration():
        actions.copy_(torch.rand(args.envs, 2, dtype=obs.dtype, device=obs.device) * 2 - 1)   # a random policy
        sim.step_into(actions, obs, reward, done_u8)
        done = done_u8 != 0
        returns.add_(reward)
        # episode bookkeeping without reading anything back: the host never waits for the GPU inside the loop
        finished.add_(done.sum())
        sum_returns.add_(torch.where(done, returns, torch.zeros_like(returns)).sum())
        returns.copy_(torch.where(done, torch.zeros_like(returns), returns))

    graph = None
    if args.graph:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                iteration()                     # warm-up outside the capture
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            iteration()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(args.steps):
        graph.replay() if graph is not None else iteration()
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(f"{'HIP graph: ' if graph is not None else ''}{args.envs} envs x {args.steps} steps in {dt:.2f} s = {args.envs * args.steps / dt / 1e6:.1f} M env-steps/s; "
          f"{int(finished)} episodes finished, mean return {float(sum_returns) / max(int(finished), 1):.2f}")
    r, d = env.get_state_info(obs[0].cpu().numpy(), [actions[0].cpu().numpy(), actions[0].cpu().numpy()])
    print("get_state_info of env 0:", r, d)
    env.close()


if __name__ == "__main__":
    main()
