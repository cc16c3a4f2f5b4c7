#!/bin/bash
# Round-5: the rocprofv3 profile (kernel trace + PMC passes) of the C3 workload, as tools/sessions/r5_final.sh does for the default (C4) command
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r5_prof_c3
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
rm -rf gpurun_out/prof_r05c3
bash tools/profile.sh r05c3 --no-count --no-gym-level --workload C3 > "$OUT/profile.log" 2>&1; echo "profile rc=$?"
bash tools/profile_issue.sh r05c3 --no-count --no-gym-level --workload C3 > "$OUT/profile_issue.log" 2>&1; echo "profile_issue rc=$?"
OS2R_TIMED_STEPS=1000 python tools/summarize_profile.py gpurun_out/prof_r05c3 "$OUT/r05_step_kernel_f64_C3" step_kernel "$OUT/traffic_C3.json" > /dev/null; echo "summary rc=$?"
python tools/issue_breakdown.py "$OUT/r05_step_kernel_f64_C3.json" "$OUT/r05_issue_breakdown_C3" > /dev/null 2>&1; echo "issue rc=$?"
sed -n 1,16p "$OUT/r05_step_kernel_f64_C3.md" | cut -c1-200
rm -rf gpurun_out/prof_r05c3
