"""Diagnostic: where does a physics iteration spend its cycles?  Needs `make -C gym-os2r_amd/csrc stamps`.
Loads libos2r_stamps.so (in-kernel s_memtime stamps per phase) and prints each phase's share of
the shader-clock ticks of one env-step, averaged over waves.  Shares only: the stamp build is
slower than the real kernel (its fences forbid overlap)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import gym_os2r_amd  # noqa: F401
from gym_os2r_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "gym-os2r_amd", "libos2r_stamps.so")
import bench

PHASES = ["sincos", "ABA passes", "inverse mass matrix", "whitening (Cholesky, y)", "contact: forward kinematics",
          "contact: candidate scan", "contact: row setup (+ FK of next body)", "PGS phase 1", "PGS phase 2", "map back + integrate", "prologue (once per env-step)", "epilogue: state stores (once per env-step)",
          "  dyn: body velocities", "  dyn: inward body 4", "  dyn: inward body 3", "  dyn: inward body 2", "  dyn: inward body 1",
          "  dyn: inward body 0", "PGS phase 2: exact solves", "  Minv: inward", "epilogue: guard + history loads (once)", "epilogue: observation (once)", "epilogue: reward (once)", "epilogue: done, reset, obs store (once)",
          "  exact solve: pass 1 (S, h)", "  exact solve: factorisation + proximal solves", "  exact solve: pass 2 (impulses, cut test)", "  exact solve: step length", "  exact solve: apply"]
COLS = list(range(24)) + list(range(26, 31))   # the columns that are phases (24 / 25: the wave's life on the two clocks)
NS = 32   # kStamps in os2r_device.hpp


def main():
    class A:  # bench.build_config arguments
        workload = sys.argv[1] if len(sys.argv) > 1 else "C4"
        envs_per_gpu = int(os.environ.get("OS2R_ENVS", "65536")); dtype = "f64"; seed = 42; pgs_normal_iters = None
        pgs_iters = int(os.environ["OS2R_PGS_ITERS"]) if "OS2R_PGS_ITERS" in os.environ else None
        pgs_exact = int(os.environ["OS2R_PGS_EXACT"]) if "OS2R_PGS_EXACT" in os.environ else None
        pgs_tol = float(os.environ["OS2R_PGS_TOL"]) if "OS2R_PGS_TOL" in os.environ else None; runtime_model = False
    cfg, model, spec = bench.build_config(A, 0, 1)
    from gym_os2r_amd.sim import HipSim
    sim = HipSim(cfg)
    lib = _lib.load()
    nwg = (cfg.num_envs + 63) // 64
    buf = torch.zeros(nwg * NS, dtype=torch.int64, device="cuda")
    lib.os2r_debug_set_stamp_buffer.argtypes = [C.c_void_p, C.c_void_p]
    sim.bench_steps(int(sys.argv[2]) if len(sys.argv) > 2 else 50)
    assert lib.os2r_debug_set_stamp_buffer(sim._h, C.c_void_p(buf.data_ptr())) == 0
    acc = np.zeros(NS)
    totals = []
    for _ in range(20):
        sim.step(None, want_terminal=False)
        torch.cuda.synchronize()
        per_wave = buf.cpu().numpy().reshape(nwg, NS)
        acc += per_wave.mean(axis=0)
        tot_w = per_wave[:, COLS].sum(axis=1)
        totals.append(tot_w.copy())
        spread = (tot_w.min(), tot_w.mean(), tot_w.max(), tot_w.std(), np.percentile(tot_w, 50), np.percentile(tot_w, 99))
    acc /= 20
    ghz = float(np.median(per_wave[:, 24] / np.maximum(per_wave[:, 25], 1)) * 0.1)
    acc = acc[COLS]
    ms = sim.bench_steps(200) / 200
    print(f"stamp build: {ms * 1e3:.1f} us per env-step launch -> {ms * 1e6 / acc.sum():.3f} ns per tick of the stamped part")
    print(f"in-kernel clock: {ghz:.3f} GHz (median over waves of s_memtime / s_memrealtime x 100 MHz, last of 20 stamped launches after the pre-roll)")
    if os.environ.get("OS2R_CLOCK_JSON"):
        import json
        json.dump({"ghz": ghz, "note": f"{A.workload}, {cfg.num_envs} envs, median over waves, stamp build, after {sys.argv[2] if len(sys.argv) > 2 else 50} pre-roll steps"},
                  open(os.environ["OS2R_CLOCK_JSON"], "w"))
    tot = acc.sum()
    print(f"workload {A.workload}: {tot:.0f} ticks per env-step per wave ({tot / cfg.substeps:.0f} per physics iteration)")
    print(f"per-wave ticks of the last step: min {spread[0]:.0f} mean {spread[1]:.0f} max {spread[2]:.0f} (max/mean {spread[2] / spread[1]:.3f}; std {spread[3]:.0f}, median {spread[4]:.0f}, p99 {spread[5]:.0f})")
    for name, v in zip(PHASES, acc):
        print(f"  {name:42s} {v / cfg.substeps:9.0f} ticks/iter  {100 * v / tot:5.1f} %")
    tail_report(per_wave, totals, cfg.substeps)


def tail_report(per_wave, totals, substeps):
    """What the slowest waves of the last stamped launch did differently, and whether a slow wave stays slow."""
    tot = per_wave[:, COLS].sum(axis=1)
    order = np.argsort(-tot)
    mean = per_wave[:, COLS].mean(axis=0)
    k = max(len(tot) // 100, 1)
    top = per_wave[np.ix_(order[:k], COLS)].mean(axis=0)
    print(f"slowest {k} waves of the last launch against the mean wave ({top.sum():.0f} vs {mean.sum():.0f} ticks): the difference by phase")
    diff = top - mean
    for i in np.argsort(-diff)[:8]:
        print(f"  {PHASES[i]:42s} {diff[i] / substeps:+9.0f} ticks/iter  {100 * diff[i] / diff.sum():5.1f} % of the gap   (mean {mean[i] / substeps:.0f})")
    T = np.stack(totals)                      # [launch, wave]
    c = np.corrcoef(T[-2], T[-1])[0, 1]
    c10 = np.corrcoef(T[-11], T[-1])[0, 1] if len(T) > 10 else float('nan')
    print(f"correlation of a wave's ticks between consecutive launches {c:.2f}, ten launches apart {c10:.2f}")
    avg = T.mean(axis=0)
    print(f"mean over the 20 launches per wave: max/mean {avg.max() / avg.mean():.3f} (a launch: {np.mean(T.max(axis=1) / T.mean(axis=1)):.3f})")


if __name__ == "__main__":
    main()
