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
(PMC counts wave-instructions; a wave-instruction is priced as 64 lane operations whatever its exec mask: issued flops).
The physics-iteration count per launch is fixed (10), so its price is inside k_launch_wave.

Round 4: the split is made identifiable.  In the steady state sweeps, rows and solves move together
and a free non-negative least squares put the sweeps' flops into `launch_wave` (priced `sweep` = `body_sweep` = 0).  The
two sweep units are therefore FIXED from the operation count of the source (os2r_device.hpp: contact_row / joint_rows --
a row with nz non-zeros is nz FMA for the residual, one FMA for the impulse, one ADD for its change, nz FMA for the update;
a joint row starts its residual with a MUL), times 64 lanes:
      k_sweep      = 64 * sum over the joints j of (4 (j + 1) + 2)               (the joint-friction rows of one sweep)
      k_body_sweep = 64 * mean over the bodies that can touch of 3 (4 (b + 1) + 3)   (the three rows of one contact)
Round 5: `scanned_body` and `row_body` are collinear in the steady state (both 3.0 per wave-iteration) and the
fit priced the same code on the same robot 4.6 k / 24.0 k lane-flops in C4 against 11.9 k / 11.2 k in C3.  They are now priced
from the source's operation count as well (FMA = 2, MUL = ADD = 1; min / max, compares, selects and the reciprocal estimate are not
in the PMC counters; a Newton reciprocal `rcp_t` is 4 FMA):
      k_scanned_body(b) = 64 * (9 * candidates_b + 5 * runs_b + 4)     (scan of the compiled-in models: per candidate two FMA for the
                          height, an ADD for the run's weight, two FMA for the moments; per run of equal coordinates an ADD and two FMA)
      k_row_body(b)     = 64 * (59 + 15 (b + 1) + 3 b (b + 1) + 6 b)   (contact point, Jacobian rows of b + 1 joints, G = J Lc, three
                          squared norms and their reciprocals)
each the mean over the bodies that lie on the ground in the steady state, and only `launch_wave` and `exact_solve` are fitted
(non-negative least squares on what the fixed units leave).  Workloads that share a robot share these four prices by construction
(asserted).  `fit` refuses a result that prices the solver unit at zero while its counter is not, and reports the residual.
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
    ns = argparse.Namespace(workload=workload, envs_per_gpu=envs, dtype="f64", seed=42, pgs_iters=None, pgs_exact=None, pgs_normal_iters=None,
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


def sweep_prices(workload, nbodies=None):
    """(lane-flops of the joint rows of one phase-2 sweep, of the three rows of one contact in one sweep) x 64 lanes, from the
    operation count of os2r_device.hpp (contact_row, joint_rows) for the robot of the workload; the contact price is the mean
    over the `nbodies` most distal bodies that can touch (the links that lie on the ground in the steady state; default: all)"""
    import bench
    import gym_os2r_amd as g
    mode = bench.WORKLOADS[workload][0]
    model = g.get_model(g.config.SettingsConfig().get_config(f"task_modes/{mode}/model"))
    nq = len(model["mass"])
    bodies = sorted({int(b) for b in model["cand_body"]})
    if nbodies:
        bodies = bodies[-int(nbodies):]
    joint = sum(4 * (j + 1) + 2 for j in range(nq))
    contact = sum(3 * (4 * (b + 1) + 3) for b in bodies) / max(len(bodies), 1)
    return 64.0 * joint, 64.0 * contact


def geometry_prices(workload, nbodies=None):
    """(lane-flops of one body's candidate scan, of one body's contact-row set-up) x 64 lanes from the operation count of
    os2r_device.hpp (step 5 of substep), means over the `nbodies` most distal bodies that can touch"""
    import bench
    import numpy as np
    import gym_os2r_amd as g
    mode = bench.WORKLOADS[workload][0]
    model = g.get_model(g.config.SettingsConfig().get_config(f"task_modes/{mode}/model"))
    cb = np.array([int(b) for b in model["cand_body"]])
    cp = np.array(model["cand_p"], dtype=float)
    bodies = sorted(set(cb.tolist()))
    if nbodies:
        bodies = bodies[-int(nbodies):]
    scan, rows = [], []
    for b in bodies:
        p = cp[cb == b]
        runs = min(1 + int((p[1:, a] != p[:-1, a]).sum()) for a in range(3))   # CandMeta: the coordinate with the fewest runs
        scan.append(9 * len(p) + 5 * runs + 4)
        rows.append(59 + 15 * (b + 1) + 3 * b * (b + 1) + 6 * b)
    return 64.0 * sum(scan) / len(scan), 64.0 * sum(rows) / len(rows)


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
        # the sweep units from the source's operation count (see the header); the model of the workload gives the bodies
        in_form = int(round(float(np.median(X[-200:, 2])) / (waves * 10.0)))     # bodies with rows per wave and iteration, steady state
        k_sweep, k_body = sweep_prices(wl, max(in_form, 1))
        k_scan, k_rows = geometry_prices(wl, max(in_form, 1))
        fixed = k_body * X[:, 3] + k_sweep * X[:, 4] + k_scan * X[:, 1] + k_rows * X[:, 2]
        free = [0, 5]
        Xf = X[:, free]
        scale = np.maximum(Xf.max(axis=0), 1.0)
        kf, _ = nnls(Xf / scale, np.maximum(y - fixed, 0.0))          # prices cannot be negative (the regressors are correlated in time)
        kf = kf / scale
        k = np.array([kf[0], k_scan, k_rows, k_body, k_sweep, kf[1]])
        res = (X @ k - y) / y
        names = ["launch_wave", "scanned_body", "row_body", "body_sweep", "sweep", "exact_solve"]
        for j, nm in enumerate(names):
            assert not (nm in ("body_sweep", "sweep", "exact_solve") and k[j] <= 0.0 and X[:, j].sum() > 0), f"{wl}: unit {nm} priced 0 while its counter is not"
        out[wl + "_f64"] = {"flops_per_unit": dict(zip(names, k.tolist()), wave_iteration=0.0),
                            "fit": {"launches": int(n), "rel_residual_rms": float(np.sqrt((res ** 2).mean())),
                                    "rel_residual_max": float(np.abs(res).max()),
                                    "flops_per_env_step_first_20": float(y[:20].mean() / counts["envs"]),
                                    "flops_per_env_step_last_200": float(y[-200:].mean() / counts["envs"])},
                            "source": f"profiles/flop_model.json[{wl}_f64]: sweep, scan and row units from the source's operation count, launch_wave and exact_solve by least squares of 64*(2*FMA+MUL+ADD) "
                                      f"(rocprofv3 --pmc, per launch) on the work counters of the same {n} launches from the reset"}
        md.append(f"## {wl} (f64, {counts['envs']} envs, {n} launches from the reset)\n")
        md.append("| unit | lane-flops per unit |\n|---|---|\n" + "".join(f"| {a} | {b:.1f} |\n" for a, b in zip(names, k)))
        md.append(f"\nrelative residual per launch: rms {np.sqrt((res ** 2).mean()):.2e}, max {np.abs(res).max():.2e}; PMC flops per env-step: "
                  f"{y[:20].mean() / counts['envs']:.0f} (first 20 launches after the reset), {y[-200:].mean() / counts['envs']:.0f} (last 200)\n")
    # workloads on the same robot price the units that are counted from the source alike
    if "C3_f64" in out and "C4_f64" in out:
        for nm in ("scanned_body", "row_body", "body_sweep", "sweep"):
            a_, b_ = out["C3_f64"]["flops_per_unit"][nm], out["C4_f64"]["flops_per_unit"][nm]
            assert abs(a_ - b_) <= 0.1 * max(a_, b_), (nm, a_, b_)
    # bench.py reads the entry of its workload; keep one flat default for C4
    with open(dst + ".json", "w") as f:
        json.dump(out, f, indent=1)
    with open(dst + ".md", "w") as f:
        f.write("\n".join(md))
    print("\n".join(md))


if __name__ == "__main__":
    main()
