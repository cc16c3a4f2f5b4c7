#!/usr/bin/env python3
"""Instruction-issue breakdown of the step kernel from a tools/summarize_profile.py summary.

Usage: python tools/issue_breakdown.py profiles/<summary>.json profiles/r01_issue_breakdown
Writes <dst>.json (read by bench.py for the executed-flops roofline) and <dst>.md.  Counters are the means
per launch of tools/profile.sh + tools/profile_issue.sh, divided by the waves of the launch.
"""
import json
import os
import sys

ORDER = ["SQC_ICACHE_REQ", "SQC_ICACHE_HITS", "SQC_ICACHE_MISSES", "SQ_IFETCH", "SQ_INSTS_BRANCH", "SQ_ACTIVE_INST_SCA",
         "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_MISC", "SQ_ACTIVE_INST_VMEM", "SQ_INST_CYCLES_SALU", "SQ_WAIT_INST_LDS",
         "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_TRANS_F64",
         "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_CVT", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_ADDR_CONFLICT",
         "SQ_INSTS_LDS_LOAD", "SQ_INSTS_LDS_STORE", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES",
         "SQ_ACTIVE_INST_ANY", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_WAIT_ANY"]


def main():
    src, dst = sys.argv[1], sys.argv[2]
    d = json.load(open(src))
    c = d["counters_per_launch"]
    waves = c["SQ_WAVES"]
    pw = {k: c[k] / waves for k in ORDER if k in c}
    lanes = 64
    flops = 2 * pw["SQ_INSTS_VALU_FMA_F64"] + pw["SQ_INSTS_VALU_MUL_F64"] + pw["SQ_INSTS_VALU_ADD_F64"]   # per lane = per env-step
    spec = None
    fp = os.path.join(os.path.dirname(os.path.abspath(src)), "flops.json")
    if os.path.exists(fp):
        spec = json.load(open(fp)).get("C4_f64", {}).get("flops_per_env_step")
    tot, valu, sca, lds, misc = (pw["SQ_WAVE_CYCLES"], pw["SQ_ACTIVE_INST_VALU"], pw["SQ_INST_CYCLES_SALU"],
                                 pw["SQ_ACTIVE_INST_LDS"], pw["SQ_ACTIVE_INST_MISC"])
    idle = tot - valu - sca - lds - misc
    note = (f"FMA counted as 2; v_max/v_min/compare are not in these counters. Wave-cycle accounting (quad-cycles per wave): "
            f"total {tot:.0f}, VALU {valu:.0f}, scalar {sca:.0f}, LDS {lds:.0f}, branch/misc {misc:.0f}, idle {idle:.0f} "
            f"of which SQ_WAIT_INST_LDS {pw['SQ_WAIT_INST_LDS']:.0f}; SQ_WAIT_ANY {pw['SQ_WAIT_ANY']:.0f}")
    source = ("tools/profile.sh + tools/profile_issue.sh (rocprofv3 --pmc, separate passes over the default bench.py command), "
              f"step kernel, mean per launch; kernel version of profiles/{os.path.splitext(os.path.basename(src))[0]}")
    json.dump({"source": source, "per_wave": pw, "fp64_flops_per_env_step_counted": flops, "note": note},
              open(dst + ".json", "w"), indent=1)
    with open(dst + ".md", "w") as f:
        f.write("# Instruction-issue breakdown of the step kernel (C4, f64, default bench run)\n\n" + source + "\n\n")
        f.write("| counter | per wave and env-step |\n|---|---|\n")
        for k in ORDER:
            if k in pw:
                f.write(f"| {k} | {pw[k]:.1f} |\n")
        f.write(f"\nCounted fp64 flops per env-step and lane (2·FMA + MUL + ADD): {flops:.0f}"
                + (f" (analytic count of the specification: {spec})" if spec else "") + ".\n\n" + note + "\n")
    print(open(dst + ".md").read())


if __name__ == "__main__":
    main()
