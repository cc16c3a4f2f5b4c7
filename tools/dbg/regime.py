"""Diagnostic: launch time, termination rate and the solver's activity per 100 env-steps from the reset -- shows where the
rollout becomes stationary (what bench.py --preroll is for), and whether it stays so.

  python tools/dbg/regime.py [WORKLOAD ...] [STEPS]        (default: V1 C4, 1200 steps)"""
import sys, argparse, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from gym_os2r_amd.sim import HipSim, Os2rError

args = sys.argv[1:]
steps = int(args.pop()) if args and args[-1].isdigit() else 1200
for wl in (args or ["V1", "C4"]):
    ns = argparse.Namespace(workload=wl, envs_per_gpu=4096 if wl == "C2" else 65536, dtype="f64", seed=42, pgs_iters=None, pgs_exact=None,
                            pgs_normal_iters=None, pgs_tol=None, runtime_model=False)
    cfg, _, _ = bench.build_config(ns, 0, 1)
    sim = HipSim(cfg, device="cuda:0")
    print(wl, "per 100 env-steps: [from step] us per launch | done rate | counted over 10 further steps: sweeps, solves per wave-iteration, lanes per solve, bodies in contact per env")
    for k in range(steps // 100):
        ms = sim.bench_steps(90) / 90
        act = ""
        try:
            sim.count_work(True)
            sim.bench_steps(9)
            c = sim.work_counters()
            sim.count_work(False)
            wi = max(c["wave_iterations"], 1)
            act = (f"sweeps {c['sweeps'] / wi:.2f} solves {c['exact_solves'] / wi:.2f} lanes/solve {c['lane_exact_solves'] / max(c['exact_solves'], 1):.2f} "
                   f"contacts/env {c['lane_contacts'] / (wi * 64.0):.2f} row bodies {c['row_bodies'] / wi:.2f}")
        except Os2rError:
            sim.count_work(False)
            sim.bench_steps(9)
        obs, rew, done, _ = sim.step(None, want_terminal=False)
        print(f"  [{100 * k:5d}] {ms * 1e3:6.1f} us | {float((done != 0).float().mean()):.4f} | {act}", flush=True)
    sim.close()
