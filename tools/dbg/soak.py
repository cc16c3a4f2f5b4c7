"""Diagnostic soak: a long random-action rollout of the bench workload; counts guard trips and checks the
state stays finite and bounded."""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

import bench
from gym_os2r_amd import abi
from gym_os2r_amd.sim import HipSim

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
args = types.SimpleNamespace(workload="C4", envs_per_gpu=65536, seed=7, dtype=sys.argv[2] if len(sys.argv) > 2 else "f64",
                             pgs_iters=None, pgs_exact=None, pgs_normal_iters=None, pgs_tol=None, runtime_model=False)
cfg, model, spec = bench.build_config(args, 0, 1)
sim = HipSim(cfg)
sim.reset()
n_done = torch.zeros((), dtype=torch.int64, device="cuda")
n_trunc = torch.zeros_like(n_done); n_bad = torch.zeros_like(n_done)
worst_q = torch.zeros((), dtype=torch.float64, device="cuda"); worst_qd = torch.zeros_like(worst_q)
for k in range(steps):
    obs, rew, d, _ = sim.step(None, want_terminal=False)
    n_done += (d & abi.DONE_BIT != 0).sum(); n_trunc += (d & abi.TRUNCATED_BIT != 0).sum(); n_bad += (d & abi.NONFINITE_BIT != 0).sum()
    if k % 500 == 499:
        q, qd = sim.get_state()
        assert bool(torch.isfinite(q).all()) and bool(torch.isfinite(qd).all()) and bool(torch.isfinite(obs).all())
        worst_q = torch.maximum(worst_q, q.abs().max().double()); worst_qd = torch.maximum(worst_qd, qd.abs().max().double())
        print(f"step {k + 1}: done {int(n_done)} truncated {int(n_trunc)} non-finite {int(n_bad)} max|q| {float(worst_q):.2f} max|qd| {float(worst_qd):.1f} "
              f"mean reward {float(rew.mean()):.4f}", flush=True)
print("soak ok")
