#!/usr/bin/env python3
"""Per-unit fp64 flop prices of the step kernel, fitted to rocprofv3 PMC counts (bench.py's roofline_valu).

bench.py counts the WORK of its timed window with the counting variant of the step kernel (include/os2r.h:
os2r_set_work_counters) and prices it with profiles/flop_model.json.  This tool produces that file:

  on the GPU box, same rollout twice (it is deterministic: same seed, same launches):
    rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 --output-format csv \\
        -d gpurun_out/flopmodel/pmc_C4 -- python3 tools/flop_model.py run --workload C4 --steps 1200
    python3 tools/flop_model.py counts --workload C4 --steps 1200 --out gpurun_out/flopmodel/counts_C4.json
  anywhere:
    python3 tools/flop_model.py fit gpurun_out/flopmodel profiles/flop_model

`run` launches `steps` env-steps from the reset (the light free-fall phase, the collapse and the steady state
are all in the window, so the regressors vary); `counts` repeats them with the counting kernel, reading the
counters back after every launch; `fit` aligns the two by launch index and solves, per workload,
      64 * (2*FMA + MUL + ADD)_launch  =  k_launch_wave * waves + k_scanned_body * scanned + k_row_body * rows
                                          + k_body_sweep * body_sweeps + k_sweep * sweeps + k_exact_solve * exact_solves
by non-negative least squares (PMC counts wave-instructions; a wave-instruction is priced as 64 lane operations whatever its
exec mask: issued flops).  The physics-iteration count per launch is fixed (10), so its price is inside
k_launch_wave.  The report gives the residual of the fit per launch.
"""
import argparse
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_sim(workload, envs):
    import bench
    ns = argparse.Namespace(workload=workload, envs_per_gpu=envs, dtype="f64", seed=42, pgs_iters=None, pgs_exact=None, pgs_normal_iters=3,
                            pgs_tol=None, runtime_model=False)
    from gym_os2r_amd.sim import HipSim
    cfg, _, _ = bench.build_config(ns, 0, 1)
    return HipSim(cfg, device="cuda:0")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["run", "counts", "fit"])
    ap.add_argument("paths", nargs="*")
    ap.add_argument("--workload", default="C4")
    ap.add_argument("--envs", type=int, default=65536)
    ap.add_argument("--steps", type=int, default=1200)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    if a.mode == "run":
        sim = make_sim(a.workload, a.envs)
        sim.bench_steps(a.steps)
        sim.close()
    elif a.mode == "counts":
        sim = make_sim(a.workload, a.envs)
        sim.count_work(True)
        rows = []
        for _ in range(a.steps):
            ms = sim.bench_steps(1)
            c = sim.work_counters()
            c["ms"] = ms
            rows.append(c)
        sim.close()
        with open(a.out, "w") as f:
            json.dump({"workload": a.workload, "envs": a.envs, "launches": rows}, f)
    else:
        fit(a.paths[0], a.paths[1])


def pmc_per_launch(d):
    """{counter: [value per step-kernel dispatch, in dispatch order]} from a rocprofv3 --pmc output directory"""
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "step_kernel" not in r["Kernel_Name"]:
                continue
            acc.setdefault(r["Counter_Name"], {}).setdefault(int(r["Dispatch_Id"]), 0.0)
            acc[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    return {k: [v[i] for i in sorted(v)] for k, v in acc.items()}


def fit(src, dst):
    import numpy as np
    out, md = {}, ["# fp64 flop prices of the step kernel's work units (tools/flop_model.py)\n"]
    for cf in sorted(glob.glob(os.path.join(src, "counts_*.json"))):
        counts = json.load(open(cf))
        wl = counts["workload"]
        pmc = pmc_per_launch(os.path.join(src, "pmc_" + wl))
        L = counts["launches"]
        n = min(len(L), len(pmc["SQ_INSTS_VALU_FMA_F64"]))
        y = 64.0 * (2 * np.array(pmc["SQ_INSTS_VALU_FMA_F64"][:n]) + np.array(pmc["SQ_INSTS_VALU_MUL_F64"][:n])
                    + np.array(pmc["SQ_INSTS_VALU_ADD_F64"][:n]))
        waves = (counts["envs"] + 63) // 64
        X = np.array([[waves, c["scanned_bodies"], c["row_bodies"], c["body_sweeps"], c["sweeps"], c.get("exact_solves", 0)] for c in L[:n]], dtype=float)
        from scipy.optimize import nnls
        scale = np.maximum(X.max(axis=0), 1.0)
        k, _ = nnls(X / scale, y)          # prices cannot be negative (the regressors are correlated in time)
        k = k / scale
        res = (X @ k - y) / y
        names = ["launch_wave", "scanned_body", "row_body", "body_sweep", "sweep", "exact_solve"]
        out[wl + "_f64"] = {"flops_per_unit": dict(zip(names, k.tolist()), wave_iteration=0.0),
                            "fit": {"launches": int(n), "rel_residual_rms": float(np.sqrt((res ** 2).mean())),
                                    "rel_residual_max": float(np.abs(res).max()),
                                    "flops_per_env_step_first_20": float(y[:20].mean() / counts["envs"]),
                                    "flops_per_env_step_last_200": float(y[-200:].mean() / counts["envs"])},
                            "source": f"profiles/flop_model.json[{wl}_f64]: least squares of 64*(2*FMA+MUL+ADD) (rocprofv3 --pmc, per launch) "
                                      f"on the work counters of the same {n} launches from the reset"}
        md.append(f"## {wl} (f64, {counts['envs']} envs, {n} launches from the reset)\n")
        md.append("| unit | lane-flops per unit |\n|---|---|\n" + "".join(f"| {a} | {b:.1f} |\n" for a, b in zip(names, k)))
        md.append(f"\nrelative residual per launch: rms {np.sqrt((res ** 2).mean()):.2e}, max {np.abs(res).max():.2e}; PMC flops per env-step: "
                  f"{y[:20].mean() / counts['envs']:.0f} (first 20 launches after the reset), {y[-200:].mean() / counts['envs']:.0f} (last 200)\n")
    # bench.py reads the entry of its workload; keep one flat default for C4
    with open(dst + ".json", "w") as f:
        json.dump(out, f, indent=1)
    with open(dst + ".md", "w") as f:
        f.write("\n".join(md))
    print("\n".join(md))


if __name__ == "__main__":
    main()
