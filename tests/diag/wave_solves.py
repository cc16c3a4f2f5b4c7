#!/usr/bin/env python3
"""Who makes the slow waves of a launch?  

  python tests/diag/wave_solves.py [--workload C4|C3|V1] [--envs 16384] [--steps 10] [--warm 1|0] [--first K] [--show 6]

Rolls the bench workload into its stationary regime with the CPU oracle and reads the oracle's solver diagnostics -- the
phase-2 sweeps and exact solves of every (environment, physics iteration) -- as the kernel would experience them: 64
consecutive environments are a wave, a wave runs as many solve rounds in an iteration as the slowest of its lanes, and a
launch lasts as long as its slowest wave.  Prints the distribution of a wave's solves and re-test sweeps per env-step,
the environment-iteration histogram, and for the slowest waves which lane drove each iteration (one lane whose first
solve is cut in every iteration = a contact that slides through the env-step and that every cold start takes for sticking:
what the warm start of DESIGN.md 3.2 removes; --warm 0 switches it off in the oracle, --first K sets the sweeps before the
first check after a warm start, for comparison)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench                          # noqa: E402
from oracle import oracle_py as O     # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C4")
    ap.add_argument("--envs", type=int, default=16384)
    ap.add_argument("--preroll", type=int, default=600)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warm", type=int, default=1)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--show", type=int, default=6)
    a = ap.parse_args()

    class A:
        workload = a.workload; envs_per_gpu = a.envs; dtype = "f64"; seed = 42
        pgs_iters = None; pgs_normal_iters = None; pgs_tol = None; pgs_exact = None; runtime_model = False
    cfg, model, spec = bench.build_config(A, 0, 1)
    L = O.use_laboratory()   # the laboratory build (oracle/Makefile)
    L.orc_set_experimental_warm(int(a.warm), int(a.first))
    o = O.OracleSim(cfg, threads=os.cpu_count() or 1)
    for _ in range(a.preroll):
        o.step(None)
    o.solver_counts()
    W = cfg.num_envs // 64
    first = None
    tots, rounds, hist, last = [], [], np.zeros(16, dtype=np.int64), None
    for _ in range(a.steps):
        o.step(None)
        sw, so = o.solver_counts()
        so = so.reshape(so.shape[0], W, 64).astype(int)
        sw = sw.reshape(sw.shape[0], W, 64).astype(int)
        first = int(sw.min()) if first is None else min(first, int(sw.min()))
        tots.append(so.max(axis=2).sum(axis=0))
        rounds.append(sw.max(axis=2).sum(axis=0))
        hist += np.bincount(so.ravel(), minlength=16)[:16]
        last = (sw, so)
    T, R = np.concatenate(tots), np.concatenate(rounds) - first * cfg.substeps
    # modelled phase-2 time of a wave and env-step: 1050 ticks a sweep, 4200 a solve (profiles/r03_stamps_C4.txt)
    M = 1050.0 * np.concatenate(rounds) + 4200.0 * T
    print(f"  modelled phase-2 ticks per wave and env-step: mean {M.mean() / 1e3:.1f} k  p99 {np.percentile(M, 99) / 1e3:.1f} k  max {M.max() / 1e3:.1f} k"
          f"  (first sweeps {first})")
    print(f"{a.workload}, {cfg.num_envs} envs = {W} waves, {a.steps} env-steps after {a.preroll}, warm start {'on' if a.warm else 'off'}")
    print(f"  exact solves per wave and env-step: mean {T.mean():.2f}  p50 {np.percentile(T, 50):.0f}  p90 {np.percentile(T, 90):.0f}  "
          f"p99 {np.percentile(T, 99):.0f}  max {T.max()};  re-test sweeps: mean {R.mean():.2f}  p99 {np.percentile(R, 99):.0f}  max {R.max()}")
    print(f"  (environment, iteration) pairs by number of solves: {hist[:np.max(np.nonzero(hist)) + 1]}")
    wmax = np.stack(tots).max(axis=1)
    print(f"  slowest wave of each env-step: {wmax}  (mean wave {T.mean():.1f})")
    sw, so = last
    tot = so.max(axis=2).sum(axis=0)
    for w in np.argsort(-tot)[:a.show]:
        am = so[:, w, :].argmax(axis=1)
        lane = np.bincount(am).argmax()
        print(f"  wave {w}: {tot[w]} solves; per iteration {so[:, w, :].max(axis=1)}; driven by lanes {am}; lane {lane}: solves {so[:, w, lane]} sweeps {sw[:, w, lane]}")
    L.orc_set_experimental_warm(1, 0)


if __name__ == "__main__":
    main()
