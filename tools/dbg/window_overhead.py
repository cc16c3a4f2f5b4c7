"""Diagnostic: what the driver's 20-step window pays beside its kernels.  bench.py brackets K launches with synchronize on both
sides; the GPU is idle at t0, so the window holds the first launch's latency and the wake-up behind the last kernel.
  python tools/dbg/window_overhead.py [K] [repeats]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import types
import torch
import bench
from gym_os2r_amd.sim import HipSim

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
R = int(sys.argv[2]) if len(sys.argv) > 2 else 12
args = types.SimpleNamespace(workload="C4", envs_per_gpu=65536, seed=42, dtype="f64", pgs_iters=None, pgs_exact=None, pgs_normal_iters=None, pgs_tol=None, runtime_model=False)
cfg, _, _ = bench.build_config(args, 0, 1)
sim = HipSim(cfg)
sim.bench_steps(1200)
rows = []
for r in range(R):
    sim.bench_enqueue(40)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ms = sim.bench_steps(K)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    rows.append(((t1 - t0) * 1e6, (t2 - t1) * 1e6, ms * 1e3))
for a, b, k in rows:
    print(f"bench_steps({K}) call {a:8.1f} us  + synchronize {b:6.1f} us  = {a + b:8.1f}; kernels (events) {k:8.1f} us; outside the kernels {a + b - k:6.1f} us = {(a + b - k) / K:5.2f} us per step")
