"""Diagnostic: how does the time of one env-step launch scale with the batch and with the physics iterations per
env-step?  Separates what a launch costs once (dispatch, first loads, epilogue, write-back) from what a wave
costs per physics iteration."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from gym_os2r_amd.sim import HipSim


def run(n, substeps, warm=600, steps=400):
    args = argparse.Namespace(workload="C4", envs_per_gpu=n, seed=42, dtype="f64", pgs_iters=None, pgs_exact=None, pgs_normal_iters=None, pgs_tol=None, runtime_model=False)
    cfg, _, _ = bench.build_config(args, 0, 1)
    sim = HipSim(cfg)
    sim.reset()
    sim.bench_steps(warm)                      # the steady state of the bench, reached with the nominal 10 iterations
    if substeps != cfg.substeps:
        sim.close()
        cfg.substeps = substeps
        sim2 = HipSim(cfg)
        sim2.reset(); sim2.bench_steps(warm * 10 // substeps if substeps < 10 else warm)
        sim = sim2
    us = sim.bench_steps(steps) / steps * 1e3
    sim.close()
    return us


for n in (16384, 32768, 49152, 65536, 81920, 98304, 131072, 196608, 262144):
    us = run(n, 10)
    print(f"envs {n:7d} ({n / 65536:.2f} waves per SIMD): {us:8.1f} us per launch, {n / us:7.1f} M env-steps/s", flush=True)
for sub in (1, 2, 5, 10, 20):
    us = run(65536, sub)
    print(f"65536 envs, {sub:2d} physics iterations per env-step: {us:8.1f} us per launch", flush=True)
