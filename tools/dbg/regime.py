"""Diagnostic: launch time and termination rate per 100 env-steps from the reset, for V1 and C4 -- shows where the
rollout becomes stationary (what bench.py --preroll is for)."""
import sys, argparse, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from gym_os2r_amd.sim import HipSim
for wl in ("V1", "C4"):
    ns = argparse.Namespace(workload=wl, envs_per_gpu=65536, dtype="f64", seed=42, pgs_iters=None, pgs_exact=None, pgs_normal_iters=3, pgs_tol=None, runtime_model=False)
    cfg, _, _ = bench.build_config(ns, 0, 1)
    sim = HipSim(cfg, device="cuda:0")
    out = []
    for k in range(12):
        ms = sim.bench_steps(100) / 100
        obs, rew, done, _ = sim.step(None, want_terminal=False)
        out.append((round(ms * 1e3, 1), round(float((done != 0).float().mean()), 4)))
    print(wl, out)
    sim.close()
