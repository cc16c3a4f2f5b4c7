#!/bin/bash
# Round-end measurement session: the bench lines, the rocprofv3 profile of the default command (trace + PMC passes),
# the flop model, the in-kernel clock.  Summaries are written under gpurun_out/final/ and copied to profiles/ by hand.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/final
mkdir -p "$OUT/flopmodel"
export TMPDIR=/tmp
cd "$ROOT"
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); rv = d.get("roofline_valu", {}); cb = d.get("cpu_baseline", {})
    print(sys.argv[1].split("/")[-1], round(d["value"] / 1e6, 1), "M/s", round(d["ms_per_step"] * 1e3, 2), "us/step kernel", round(d["roofline"]["kernel_ms_per_launch"] * 1e3, 2), "valu frac", rv.get("frac"), "useful", rv.get("frac_useful"), "flops/env-step", rv.get("issued_lane_flops_per_env_step"), "cpu", cb.get("value"), "| sweeps", round(rv.get("activity", {}).get("phase2_sweeps_per_wave_iteration", 0), 2), "solves", round(rv.get("activity", {}).get("exact_solves_per_wave_iteration", 0), 2), "envs/solve", round(rv.get("activity", {}).get("envs_per_exact_solve", 0), 2))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
# the stamp build does not travel (.gpurunignore): build it here, on the box's cores
( time make -C gym-os2r_amd/csrc STAMPS=1 -j16 ../libos2r_stamps.so > "$OUT/make_stamps.log" 2>&1 ) 2>&1 | grep real
timeout -k 10 300 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; show "$OUT/bench_default.json"
B="timeout -k 10 200 python bench.py --no-cpu-baseline"
$B --steps 20 --warmup 5 > "$OUT/bench_20_5.json" 2>/dev/null; show "$OUT/bench_20_5.json"
$B --splits 4 > "$OUT/bench_splits4.json" 2>/dev/null; show "$OUT/bench_splits4.json"
$B --splits 2 > "$OUT/bench_splits2.json" 2>/dev/null; show "$OUT/bench_splits2.json"
$B --pgs-exact 0 --pgs-iters 20 > "$OUT/bench_sweeps_only.json" 2>/dev/null; show "$OUT/bench_sweeps_only.json"
$B --pgs-exact 0 --pgs-iters 20 --steps 20 --warmup 5 > "$OUT/bench_sweeps_only_20_5.json" 2>/dev/null; show "$OUT/bench_sweeps_only_20_5.json"
$B --pgs-tol 1e-3 > "$OUT/bench_floor.json" 2>/dev/null; show "$OUT/bench_floor.json"
for w in C2 C3 V1; do $B --workload $w > "$OUT/bench_$w.json" 2>/dev/null; show "$OUT/bench_$w.json"; done
$B --runtime-model > "$OUT/bench_rt.json" 2>/dev/null; show "$OUT/bench_rt.json"
$B --envs-per-gpu 131072 --steps 500 > "$OUT/bench_128k.json" 2>/dev/null; show "$OUT/bench_128k.json"
$B --envs-per-gpu 524288 --steps 200 --preroll 600 > "$OUT/bench_512k.json" 2>/dev/null; show "$OUT/bench_512k.json"
$B --dtype f32 > "$OUT/bench_f32_64k.json" 2>/dev/null; show "$OUT/bench_f32_64k.json"
$B --dtype f32 --envs-per-gpu 131072 --steps 500 > "$OUT/bench_f32_128k.json" 2>/dev/null; show "$OUT/bench_f32_128k.json"
$B --dtype f32 --envs-per-gpu 262144 --steps 300 > "$OUT/bench_f32_256k.json" 2>/dev/null; show "$OUT/bench_f32_256k.json"
OS2R_CLOCK_JSON="$OUT/clock.json" timeout -k 10 200 python tools/dbg/stamps.py C4 1100 > "$OUT/stamps.txt" 2>&1; grep -E "clock|stamp build|per-wave" "$OUT/stamps.txt"
rm -rf gpurun_out/prof_r03
bash tools/profile.sh r03 --no-count > "$OUT/profile.log" 2>&1; echo "profile rc=$?"
bash tools/profile_issue.sh r03 --no-count > "$OUT/profile_issue.log" 2>&1; echo "profile_issue rc=$?"
OS2R_TIMED_STEPS=1000 python tools/summarize_profile.py gpurun_out/prof_r03 "$OUT/r03_step_kernel_f64_C4" step_kernel "$OUT/traffic.json" "$OUT/clock.json" > /dev/null
python tools/issue_breakdown.py "$OUT/r03_step_kernel_f64_C4.json" "$OUT/r03_issue_breakdown" > /dev/null 2>&1; echo "issue rc=$?"
timeout -k 10 300 python tools/flop_model.py counts --workload C4 --steps 1200 --out "$OUT/flopmodel/counts_C4.json"; echo "counts rc=$?"
cd /tmp
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU SQ_WAVES --output-format csv -d "$OUT/flopmodel/pmc_C4" -- python3 "$ROOT/tools/flop_model.py" run --workload C4 --steps 1200 > "$OUT/flopmodel/pmc.log" 2>&1; echo "pmc rc=$?"
cd "$ROOT"; python tools/flop_model.py fit "$OUT/flopmodel" "$OUT/flop_model" | tail -12
rm -rf "$OUT/flopmodel/pmc_C4"   # (large; the fit is what is kept)
sed -n 1,30p "$OUT/r03_step_kernel_f64_C4.md"
# the raw rocprofv3 output is large (8560 dispatches per pass) and is not kept: the summaries above are
rm -rf gpurun_out/prof_r03
du -sh gpurun_out
