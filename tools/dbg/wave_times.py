"""Diagnostic: when do the waves of a launch start and end, and where do they run?  Needs `make -C gym-os2r_amd/csrc stamps_light`
(the shipped kernel's code plus each wave's start / end on the 100 MHz clock, its life in shader cycles and its HW_ID / XCC_ID).

  python tools/dbg/wave_times.py [WORKLOAD] [PREROLL]

Prints, over 20 launches after the pre-roll: the launch's span (first start to last end), the spread of the starts, the
distribution of the waves' lives, what the last waves to end have in common (place, start time), the clock."""
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

NS = 32


def main():
    class A:
        workload = sys.argv[1] if len(sys.argv) > 1 else "C4"
        envs_per_gpu = int(os.environ.get("OS2R_ENVS", "65536")); dtype = "f64"; seed = 42; pgs_normal_iters = None
        pgs_iters = None; pgs_exact = int(os.environ["OS2R_PGS_EXACT"]) if "OS2R_PGS_EXACT" in os.environ else None
        pgs_tol = float(os.environ["OS2R_PGS_TOL"]) if "OS2R_PGS_TOL" in os.environ else None; runtime_model = False
    cfg, model, spec = bench.build_config(A, 0, 1)
    from gym_os2r_amd.sim import HipSim
    sim = HipSim(cfg)
    lib = _lib.load()
    nwg = (cfg.num_envs + 63) // 64
    buf = torch.zeros(nwg * NS, dtype=torch.int64, device="cuda")
    lib.os2r_debug_set_stamp_buffer.argtypes = [C.c_void_p, C.c_void_p]
    pre = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
    ms0 = sim.bench_steps(pre) / pre
    assert lib.os2r_debug_set_stamp_buffer(sim._h, C.c_void_p(buf.data_ptr())) == 0
    ms = sim.bench_steps(200) / 200
    print(f"{A.workload}, {cfg.num_envs} envs: {ms * 1e3:.1f} us per launch (200 launches, stamps written), {ms0 * 1e3:.1f} over the pre-roll")
    spans, skews, lives, lasts = [], [], [], []
    for it in range(20):
        sim.bench_steps(3)
        torch.cuda.synchronize()
        w = buf.cpu().numpy().reshape(nwg, NS)
        st, en = w[:, 22].astype(np.int64), w[:, 23].astype(np.int64)
        t0 = st.min()
        spans.append((en.max() - t0) * 0.01); skews.append((st.max() - t0) * 0.01)
        life = (en - st) * 0.01
        lives.append(life)
        lasts.append(np.argsort(-en)[:8])
    lives = np.stack(lives)
    print(f"launch span (first start -> last end): mean {np.mean(spans):.1f} us  (min {np.min(spans):.1f}, max {np.max(spans):.1f});  last start - first start: mean {np.mean(skews):.1f} us, max {np.max(skews):.1f}")
    print(f"wave life [us]: mean {lives.mean():.1f}  p50 {np.percentile(lives, 50):.1f}  p99 {np.percentile(lives, 99):.1f}  max per launch mean {lives.max(axis=1).mean():.1f};  max/mean per launch {np.mean(lives.max(axis=1) / lives.mean(axis=1)):.3f}")
    ghz = float(np.median(w[:, 24] / np.maximum(w[:, 25], 1)) * 0.1)
    print(f"in-kernel clock {ghz:.3f} GHz")
    # the last launch in detail
    hw = w[:, 21].astype(np.uint64)
    xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xf
    hwid = (hw & np.uint64(0xffffffff)).astype(np.int64)
    simd = (hwid >> 4) & 3; cu = (hwid >> 8) & 15; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 7
    place = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    nplaces = len(set(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist(), simd.tolist())))
    print(f"last launch: {nplaces} distinct (xcc, se, sh, cu, simd) places for {nwg} waves; waves per XCC {np.bincount(xcc, minlength=8).tolist()}")
    per_cu = np.bincount(place)
    print(f"  waves per CU: {np.bincount(per_cu[per_cu > 0]).tolist()} (index = waves on a CU)")
    order = np.argsort(-en)
    rel_s, rel_e = (st - st.min()) * 0.01, (en - st.min()) * 0.01
    print("  the 12 waves that ended last: wave, start us, end us, life us, xcc/se/sh/cu/simd, waves on its SIMD")
    key = ((place * 4) + simd)
    cnt = {k: int((key == k).sum()) for k in set(key.tolist())}
    for i in order[:12]:
        print(f"    {i:5d}  {rel_s[i]:7.1f} {rel_e[i]:7.1f} {rel_e[i] - rel_s[i]:7.1f}  {xcc[i]}/{se[i]}/{sh[i]}/{cu[i]}/{simd[i]}  {cnt[int(key[i])]}")
    late = rel_s > 5.0
    print(f"  waves that started more than 5 us after the first: {int(late.sum())} (their mean start {rel_s[late].mean() if late.any() else 0:.1f} us, mean end {rel_e[late].mean() if late.any() else 0:.1f} us); SIMDs with 2 waves: {sum(1 for v in cnt.values() if v >= 2)}")
    sim.close()


if __name__ == "__main__":
    main()
