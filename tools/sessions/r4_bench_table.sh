#!/bin/bash
# GPU suite + the bench table of the kernels in the tree: tools/sessions/r4_bench_table.sh OUTDIR [notest]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/${1:-r4_table}
mkdir -p "$OUT"
cd "$ROOT"
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); rv = d.get("roofline_valu", {}); a = rv.get("activity", {})
    print(sys.argv[1].split("/")[-1], round(d["value"] / 1e6, 1), "M/s", round(d["ms_per_step"] * 1e3, 2), "us/step kernel", round(d["roofline"]["kernel_ms_per_launch"] * 1e3, 2),
          "| sweeps", round(a.get("phase2_sweeps_per_wave_iteration", 0), 2), "solves", round(a.get("exact_solves_per_wave_iteration", 0), 2),
          "envs/solve", round(a.get("envs_per_exact_solve", 0), 2))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
if [ "${2:-}" != "notest" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1; echo "pytest rc $?"; tail -4 "$OUT/pytest.log"
fi
B="timeout -k 10 200 python bench.py --no-cpu-baseline"
$B > "$OUT/bench_C4.json" 2> "$OUT/bench_C4.err"; show "$OUT/bench_C4.json" | tee -a "$OUT/table.txt"
$B --steps 20 --warmup 5 --no-count > "$OUT/bench_C4_20_5.json" 2>/dev/null; show "$OUT/bench_C4_20_5.json" | tee -a "$OUT/table.txt"
for w in C3 V1; do $B --workload $w > "$OUT/bench_$w.json" 2>/dev/null; show "$OUT/bench_$w.json" | tee -a "$OUT/table.txt"; done
$B --pgs-tol 1e-3 --no-count > "$OUT/bench_floor.json" 2>/dev/null; show "$OUT/bench_floor.json" | tee -a "$OUT/table.txt"
$B --steps 20 --warmup 5 --no-count > "$OUT/bench_C4_20_5b.json" 2>/dev/null; show "$OUT/bench_C4_20_5b.json" | tee -a "$OUT/table.txt"
$B --rollout 10 --steps 1000 > "$OUT/rollout_10.json" 2>/dev/null; show "$OUT/rollout_10.json" | tee -a "$OUT/table.txt"
