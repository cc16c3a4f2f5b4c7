#!/bin/bash
# round 4, session 4 (first of the re-entered session): GPU suite, where the kernels of the tree stand on the bench
# workloads (default window, the driver's window, C3, V1, C2, floor, rollouts), stamps of C4 and C3
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r4_s4
mkdir -p "$OUT"
cd "$ROOT"
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); rv = d.get("roofline_valu", {}); a = rv.get("activity", {})
    print(sys.argv[1].split("/")[-1], round(d["value"] / 1e6, 1), "M/s", round(d["ms_per_step"] * 1e3, 2), "us/step kernel", round(d["roofline"]["kernel_ms_per_launch"] * 1e3, 2),
          "valu frac", rv.get("frac"), "| sweeps", round(a.get("phase2_sweeps_per_wave_iteration", 0), 2), "solves", round(a.get("exact_solves_per_wave_iteration", 0), 2),
          "envs/solve", round(a.get("envs_per_exact_solve", 0), 2), "live/sweep", round(a.get("live_envs_per_sweep", 0), 1))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1; echo "pytest rc $?"; tail -5 "$OUT/pytest.log"
B="timeout -k 10 200 python bench.py --no-cpu-baseline"
$B > "$OUT/bench_C4.json" 2> "$OUT/bench_C4.err"; show "$OUT/bench_C4.json" | tee -a "$OUT/table.txt"
$B --steps 20 --warmup 5 > "$OUT/bench_C4_20_5.json" 2>/dev/null; show "$OUT/bench_C4_20_5.json" | tee -a "$OUT/table.txt"
for w in C3 V1 C2; do $B --workload $w > "$OUT/bench_$w.json" 2>/dev/null; show "$OUT/bench_$w.json" | tee -a "$OUT/table.txt"; done
$B --pgs-tol 1e-3 --no-count > "$OUT/bench_floor.json" 2>/dev/null; show "$OUT/bench_floor.json" | tee -a "$OUT/table.txt"
$B --steps 20 --warmup 5 --no-count > "$OUT/bench_C4_20_5b.json" 2>/dev/null; show "$OUT/bench_C4_20_5b.json" | tee -a "$OUT/table.txt"
for K in 10 50; do
  $B --rollout $K --steps 1000 > "$OUT/rollout_$K.json" 2>"$OUT/rollout_$K.err"; show "$OUT/rollout_$K.json" | tee -a "$OUT/table.txt"
done
$B --workload C3 --rollout 10 --steps 1000 > "$OUT/rollout_C3_10.json" 2>/dev/null; show "$OUT/rollout_C3_10.json" | tee -a "$OUT/table.txt"
make -C gym-os2r_amd/csrc stamps -j16 > "$OUT/make.log" 2>&1 || { tail -5 "$OUT/make.log"; exit 1; }
for w in C4 C3; do
  timeout -k 10 300 python tools/dbg/stamps.py $w 1200 > "$OUT/stamps_$w.txt" 2>&1 || { tail -5 "$OUT/stamps_$w.txt"; exit 1; }
  grep -v "^  *dyn\|amdgpu.ids" "$OUT/stamps_$w.txt" | head -40
done
