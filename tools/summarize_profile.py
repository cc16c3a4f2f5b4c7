#!/usr/bin/env python3
"""Condense a tools/profile.sh output directory (rocprofv3 CSVs) into one JSON + markdown summary.

Usage: python tools/summarize_profile.py gpurun_out/prof_<tag> profiles/<name> [kernel substring] [traffic.json to write] [clock.json from tools/dbg/stamps.py]
Counter corrections follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE/WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE under-reports wide coalesced streaming reads by 2x (reported as-is and
doubled, both are given; this kernel's reads are 8-byte-per-lane SoA loads, uncalibrated width).
"""
import collections
import csv
import glob
import json
import os
import sys


def counters(d, kernel_substr):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "pmc_*", "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if kernel_substr in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def main():
    src, dst = sys.argv[1], sys.argv[2]
    kern = sys.argv[3] if len(sys.argv) > 3 else "step_kernel"
    stats = []
    for f in glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")):
        stats = list(csv.DictReader(open(f)))
    trace_rows = []
    for f in glob.glob(os.path.join(src, "trace", "*", "*_kernel_trace.csv")):
        trace_rows = [r for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"]]
    c, n = counters(src, kern)
    step = next((r for r in stats if kern in r["Name"]), None)
    out = {"kernel": step["Name"] if step else None,
           "calls": int(step["Calls"]) if step else None,
           "avg_ns": float(step["AverageNs"]) if step else None,
           "min_ns": float(step["MinNs"]) if step else None,
           "max_ns": float(step["MaxNs"]) if step else None,
           "pct_of_gpu_time": float(step["Percentage"]) if step else None,
           "counters_per_launch": c}
    if trace_rows:
        # the bench's timed window is the last `steps` launches of the command (pre-roll and warm-up come first)
        import numpy as np
        dur = np.array([int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sorted(trace_rows, key=lambda r: int(r["Dispatch_Id"]))])
        nwin = int(os.environ.get("OS2R_TIMED_STEPS", "1000"))
        out["timed_window"] = {"launches": int(min(nwin, len(dur))), "avg_ns": float(dur[-nwin:].mean()), "min_ns": int(dur[-nwin:].min()),
                               "max_ns": int(dur[-nwin:].max())}
        out["avg_ns_per_100_launches"] = [float(dur[a:a + 100].mean()) for a in range(0, len(dur), 100)]
        r = trace_rows[0]
        out["launch"] = {k: r[k] for k in ("Grid_Size_X", "Workgroup_Size_X", "LDS_Block_Size", "Scratch_Size")}
    # register allocation: from the code object of the library that ran, not from rocprofv3's VGPR_Count /
    # Accum_VGPR_Count fields (which are not the allocation)
    if out["kernel"]:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import kernel_meta
        lib = os.environ.get("OS2R_LIBRARY") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gym-os2r_amd", "libos2r.so")
        co = kernel_meta.lookup(kernel_meta.kernel_meta(lib), out["kernel"]) if os.path.exists(lib) else None
        if co:
            out["code_object"] = dict(co, library=os.path.basename(lib),
                                      arch_vgprs=co["vgpr_count"] - co["agpr_count"],
                                      waves_per_simd_by_registers=max(1, min(8, 512 // (((co["vgpr_count"] + 7) // 8) * 8))))
    clock = sys.argv[5] if len(sys.argv) > 5 else None
    if clock and os.path.exists(clock):
        out["in_kernel_clock"] = json.load(open(clock))
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        out["hbm_bytes_per_launch_raw"] = (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
        out["hbm_bytes_per_launch"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
    if "SQ_INSTS_VALU" in c and "SQ_WAVES" in c:
        out["valu_insts_per_wave"] = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    with open(dst + ".json", "w") as f:
        json.dump(out, f, indent=1)
    with open(dst + ".md", "w") as f:
        f.write(f"# rocprofv3 summary: {os.path.basename(src)}\n\n")
        f.write("## kernel-trace --stats\n\n| kernel | calls | avg ns | min ns | max ns | % |\n|---|---|---|---|---|---|\n")
        for r in stats:
            f.write(f"| `{r['Name']}` | {r['Calls']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} | {float(r['Percentage']):.3f} |\n")
        if "timed_window" in out:
            w_ = out["timed_window"]
            f.write(f"\nTimed window of the command (its last {w_['launches']} launches of the step kernel; pre-roll and warm-up precede it): "
                    f"avg {w_['avg_ns']:.0f} ns, min {w_['min_ns']}, max {w_['max_ns']}.  Average per 100 launches from the reset: "
                    + ", ".join(f"{v / 1e3:.1f}" for v in out["avg_ns_per_100_launches"]) + " us.\n")
        if "launch" in out:
            f.write("\n## launch\n\n" + ", ".join(f"{k}={v}" for k, v in out["launch"].items()) + "\n")
        if "code_object" in out:
            c_ = out["code_object"]
            f.write(f"\n## code object ({c_['library']}, AMDGPU metadata note)\n\nunified registers per lane {c_['vgpr_count']} = {c_['arch_vgprs']} VGPR + "
                    f"{c_['agpr_count']} AGPR (-> {c_['waves_per_simd_by_registers']} wave(s) per SIMD), SGPRs {c_['sgpr_count']}, SGPR spills {c_['sgpr_spill_count']}, "
                    f"VGPR spills {c_['vgpr_spill_count']}, scratch {c_['private_segment_fixed_size']} B/lane, LDS {c_['group_segment_fixed_size']} B/workgroup\n")
        if "in_kernel_clock" in out:
            k_ = out["in_kernel_clock"]
            f.write(f"\n## in-kernel clock (diagnostic stamp build, s_memtime / s_memrealtime)\n\n{k_.get('ghz'):.3f} GHz ({k_.get('note', '')})\n")
        f.write("\n## PMC (mean per launch of the step kernel; separate passes)\n\n| counter | value |\n|---|---|\n")
        for k in sorted(c):
            f.write(f"| {k} | {c[k]:.6g} |\n")
        if "hbm_bytes_per_launch" in out:
            f.write(f"\nHBM bytes per launch: raw (FETCH+WRITE)*1024 = {out['hbm_bytes_per_launch_raw']:.4g}; "
                    f"with the gfx950 FETCH_SIZE x2 correction = {out['hbm_bytes_per_launch']:.4g}\n")
    if "hbm_bytes_per_launch" in out and len(sys.argv) > 4:
        # profiles/traffic.json, read by bench.py for roofline.traffic
        with open(sys.argv[4], "w") as f:
            json.dump({"hbm_bytes_per_launch": out["hbm_bytes_per_launch"],
                       "hbm_bytes_per_launch_raw": out["hbm_bytes_per_launch_raw"],
                       "kernel": out["kernel"], "source": os.path.basename(dst) + ".json",
                       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over the "
                                 "default bench.py command; (2*FETCH_SIZE + WRITE_SIZE)*1024 per "
                                 "MI355X_MICROARCH.md (FETCH_SIZE counts half of streamed reads on gfx950)"}, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
