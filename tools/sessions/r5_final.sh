#!/bin/bash
# Round-5 measurement session: the GPU suite, the bench lines, stamps (phases, the passes of the exact solve), wave start / end
# times, the rocprofv3 profile of the default command (trace + PMC passes), the flop model.  Summaries go under gpurun_out/r5_final/
# and are copied to profiles/ by hand.  tools/sessions/r5_final.sh [part ...]   parts: test bench stamps waves profile flop c2
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r5_final
mkdir -p "$OUT/flopmodel"
export TMPDIR=/tmp
cd "$ROOT"
PARTS=${*:-test bench host stamps waves profile flop}
has() { [[ " $PARTS " == *" $1 "* ]]; }
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); rv = d.get("roofline_valu", {}); cb = d.get("cpu_baseline", {}); a = rv.get("activity", {})
    print(sys.argv[1].split("/")[-1], round(d["value"] / 1e6, 1), "M/s", round(d["ms_per_step"] * 1e3, 2), "us/step kernel", round(d["roofline"]["kernel_ms_per_launch"] * 1e3, 2),
          "valu frac", rv.get("frac"), "useful", rv.get("frac_useful"), "flops/env-step", rv.get("issued_lane_flops_per_env_step"), "cpu", cb.get("value"),
          "| sweeps", round(a.get("phase2_sweeps_per_wave_iteration", 0), 2), "solves", round(a.get("exact_solves_per_wave_iteration", 0), 2), "envs/solve", round(a.get("envs_per_exact_solve", 0), 2))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
if has test; then
  timeout -k 10 900 python -m pytest tests -m gpu -q > "$OUT/pytest.log" 2>&1; echo "pytest rc $?"; grep -E "^FAILED|passed|failed" "$OUT/pytest.log" | tail -8
fi
if has bench; then
  timeout -k 10 300 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; show "$OUT/bench_default.json" | tee -a "$OUT/table.txt"
  B="timeout -k 10 200 python bench.py --no-cpu-baseline --no-gym-level"
  run() { n=$1; shift; $B "$@" > "$OUT/bench_$n.json" 2>/dev/null; show "$OUT/bench_$n.json" | tee -a "$OUT/table.txt"; }
  run 20_5 --steps 20 --warmup 5
  run splits4 --splits 4
  run sweeps_only --pgs-exact 0 --pgs-iters 20
  run floor --pgs-tol 1e-3
  for w in C2 C3 V1; do run $w --workload $w; done
  run C3_splits4 --workload C3 --splits 4
  run rt --runtime-model
  run 128k --envs-per-gpu 131072 --steps 500
  run 128k_splits4 --envs-per-gpu 131072 --steps 500 --splits 4
  run 512k_splits4 --envs-per-gpu 524288 --steps 200 --preroll 600 --splits 4
  run f32_64k --dtype f32
  run f32_128k --dtype f32 --envs-per-gpu 131072 --steps 500
  run rollout10 --rollout 10 --steps 1000
  run rollout50 --rollout 50 --steps 1000
  run C3_rollout10 --workload C3 --rollout 10 --steps 1000
  run V1_rollout10 --workload V1 --rollout 10 --steps 1000
  run 20_5_again --steps 20 --warmup 5
  timeout -k 10 200 python tools/dbg/all_modes.py > "$OUT/all_modes.txt" 2>&1; tail -8 "$OUT/all_modes.txt"
fi
if has host; then
  timeout -k 10 400 python tools/dbg/host_overhead.py 65536 300 > "$OUT/host_surface.txt" 2> "$OUT/host_surface.err"; cat "$OUT/host_surface.txt"
  timeout -k 10 300 python tools/dbg/host_profile.py 64 400 > "$OUT/host_profile_64.txt" 2>&1; head -14 "$OUT/host_profile_64.txt"
fi
if has c2; then
  # C2 (4096 environments = 64 waves on 64 of 1024 SIMDs) is one wave's latency: the same launch time at 4, 16 and 64 times the batch
  B="timeout -k 10 200 python bench.py --no-cpu-baseline --no-count --workload C2"
  for n in 4096 16384 65536 262144; do
    $B --envs-per-gpu $n --steps 500 > "$OUT/bench_C2_$n.json" 2>/dev/null; show "$OUT/bench_C2_$n.json" | tee -a "$OUT/table_c2.txt"
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/dbg/ubench_rcp.hip -o /tmp/ubench_rcp && /tmp/ubench_rcp | tee "$OUT/ubench_rcp.txt"
fi
if has stamps; then
  ( time make -C gym-os2r_amd/csrc stamps -j16 > "$OUT/make_stamps.log" 2>&1 ) 2>&1 | grep real
  for w in C4 C3; do
    OS2R_CLOCK_JSON="$OUT/clock_$w.json" timeout -k 10 300 python tools/dbg/stamps.py $w 1200 > "$OUT/stamps_$w.txt" 2>&1; grep -E "clock|stamp build|per-wave|a launch" "$OUT/stamps_$w.txt"
  done
  cp "$OUT/clock_C4.json" "$OUT/clock.json"
fi
if has waves; then
  ( time make -C gym-os2r_amd/csrc stamps_light -j16 > "$OUT/make_light.log" 2>&1 ) 2>&1 | grep real
  for w in C4 C3 V1; do
    timeout -k 10 300 python tools/dbg/wave_times.py $w 1200 > "$OUT/wave_times_$w.txt" 2>&1; grep -E "per launch|launch span|wave life" "$OUT/wave_times_$w.txt"
  done
fi
if has profile; then
  rm -rf gpurun_out/prof_r05
  bash tools/profile.sh r05 --no-count --no-gym-level > "$OUT/profile.log" 2>&1; echo "profile rc=$?"
  bash tools/profile_issue.sh r05 --no-count --no-gym-level > "$OUT/profile_issue.log" 2>&1; echo "profile_issue rc=$?"
  OS2R_TIMED_STEPS=1000 python tools/summarize_profile.py gpurun_out/prof_r05 "$OUT/r05_step_kernel_f64_C4" step_kernel "$OUT/traffic.json" "$OUT/clock.json" > /dev/null
  python tools/issue_breakdown.py "$OUT/r05_step_kernel_f64_C4.json" "$OUT/r05_issue_breakdown" > /dev/null 2>&1; echo "issue rc=$?"
  for f in $(find gpurun_out/prof_r05/trace -name "*kernel_stats.csv" | head -1); do cp "$f" "$OUT/r05_kernel_stats.csv"; done
  sed -n 1,30p "$OUT/r05_step_kernel_f64_C4.md"
  rm -rf gpurun_out/prof_r05
fi
if has flop; then
  for w in C4 C3 V1; do
    cd "$ROOT"
    timeout -k 10 300 python tools/flop_model.py counts --workload $w --steps 1200 --out "$OUT/flopmodel/counts_$w.json"; echo "counts $w rc=$?"
    cd /tmp
    timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 --output-format csv -d "$OUT/flopmodel/pmc_$w" -- python3 "$ROOT/tools/flop_model.py" run --workload $w --steps 1200 > "$OUT/flopmodel/pmc_$w.log" 2>&1; echo "pmc $w rc=$?"
  done
  cd "$ROOT"; python tools/flop_model.py fit "$OUT/flopmodel" "$OUT/flop_model" | tail -40
  rm -rf "$OUT"/flopmodel/pmc_C4 "$OUT"/flopmodel/pmc_C3 "$OUT"/flopmodel/pmc_V1
fi
du -sh "$OUT"
