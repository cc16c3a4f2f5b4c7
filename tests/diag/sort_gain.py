"""Diagnostic (not a test): what would grouping the environments by contact mask buy the step kernel?

Runs the bench workload (C4) to its steady state, computes every environment's set of links within the
contact margin on the host (CPU oracle: forward kinematics + candidate test), then times one env-step of
the environments as they are, ordered by that mask globally, and ordered inside windows of --windows
environments only (the state is permuted physically, so every access stays coalesced: this is the gain of the
grouping alone), with identical actions, and checks that every environment's result is bit-identical.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
from gym_os2r_amd import abi
from gym_os2r_amd.sim import HipSim
from oracle import oracle_py as o

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=65536)
ap.add_argument("--warm", type=int, default=300)
ap.add_argument("--windows", type=int, nargs="+", default=[256, 1024, 4096])
ap.add_argument("--rounds", type=int, default=5)
a = ap.parse_args()
args = argparse.Namespace(workload="C4", envs_per_gpu=a.envs, seed=42, dtype="f64", pgs_iters=None, pgs_exact=None, pgs_normal_iters=None, pgs_tol=None, runtime_model=False)
cfg, model, spec = bench.build_config(args, 0, 1)
n, nq = a.envs, cfg.model.nq


def masks(q, qd):
    out = np.zeros(n, dtype=np.int64)
    for e in range(n):
        _, _, rw, ow = o.dynamics(cfg.model, q[:, e], qd[:, e], np.zeros(nq))
        act, _, _ = o.contact_points(cfg.model, rw, ow, cfg.contact_margin)
        out[e] = sum(1 << b for b in range(nq) if act[b])
    return out


def bodies_per_wave(m):
    w = m.reshape(-1, 64)
    u = np.bitwise_or.reduce(w, axis=1)
    return np.mean([bin(int(x)).count("1") for x in u])


def timed_step(sim, acts):
    obs = torch.empty((n, sim.D), device="cuda", dtype=torch.float64); rew = torch.empty(n, device="cuda", dtype=torch.float64)
    done = torch.empty(n, device="cuda", dtype=torch.uint8)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    ev[0].record(); sim.step_into(acts, obs, rew, done); ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) * 1e3, done


def reordered(ck, idx):
    c = lambda t: t[..., idx].contiguous()
    return {"q": c(ck["q"]), "qd": c(ck["qd"]), "hist0": c(ck["hist0"]), "hist1": c(ck["hist1"]),
            "params": {f: c(v) for f, v in ck["params"].items()}, "steps": c(ck["steps"]),
            "episode": c(ck["episode"]), "pose": c(ck["pose"]), "step_count": ck["step_count"]}


A = HipSim(cfg)
A.reset()
A.bench_steps(a.warm)
B = HipSim(cfg)
B.reset(); B.bench_steps(3)
gen = torch.Generator(device="cuda"); gen.manual_seed(5)
for r in range(a.rounds):
    ck = A.checkpoint()
    q, qd = (t.cpu().numpy() for t in (ck["q"], ck["qd"]))
    m = masks(q, qd)
    orders = [("as-is", np.arange(n)), ("global", np.argsort(m, kind="stable"))]
    for w in a.windows:
        orders.append((f"w{w}", np.concatenate([w0 + np.argsort(m[w0:w0 + w], kind="stable") for w0 in range(0, n, w)])))
    acts = torch.rand((n, 2), generator=gen, device="cuda", dtype=torch.float64) * 2 - 1
    line = [f"round {r}:"]
    ref = None
    for name, order in orders:
        idx = torch.as_tensor(order, device="cuda")
        best = 1e9
        for rep in range(4):                       # the same step four times over: the fastest launch counts
            B.restore(reordered(ck, idx))
            us, done = timed_step(B, acts[idx].contiguous())
            best = min(best, us)
        qb, qdb = B.get_state()
        if ref is None:
            ref = (best, qb, qdb, done == 0)
            line.append(f"{name} {bodies_per_wave(m):.2f} bodies/wave {best:.1f} us |")
        else:
            k = ref[3][idx]
            same = bool(torch.equal(ref[1][:, idx][:, k], qb[:, k]) and torch.equal(ref[2][:, idx][:, k], qdb[:, k]))
            line.append(f"{name} {bodies_per_wave(m[order]):.2f} {best:.1f} us ({(ref[0] / best - 1) * 100:+.1f} %){'' if same else ' DIFFERENT'} |")
    print(" ".join(line), flush=True)
    A.bench_steps(7)
