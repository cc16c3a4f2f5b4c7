#!/bin/bash
# GPU session 1 of round 2: parity tests, window-independence of the bench line, gain of the stopping rule,
# per-launch work counters from the reset, PMC flop counts for the flop model, kernel-trace of the default run.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/s1
mkdir -p "$OUT/flopmodel"
export TMPDIR=/tmp
cd "$ROOT"
timeout -k 10 600 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest.log"
B="timeout -k 10 200 python bench.py --no-cpu-baseline"
$B > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "default rc=$?"
$B --steps 20 --warmup 5 > "$OUT/bench_20_5.json" 2>/dev/null; echo "20/5 rc=$?"
$B --pgs-tol 0 > "$OUT/bench_tol0.json" 2>/dev/null; echo "tol0 rc=$?"
$B --preroll 0 --steps 20 --warmup 5 > "$OUT/bench_nopreroll_20_5.json" 2>/dev/null
$B --workload C3 > "$OUT/bench_C3.json" 2>/dev/null
$B --workload C2 > "$OUT/bench_C2.json" 2>/dev/null
$B --workload V1 > "$OUT/bench_V1.json" 2>/dev/null
for f in default 20_5 tol0 nopreroll_20_5 C3 C2 V1; do python - "$OUT/bench_$f.json" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    rv = d.get("roofline_valu", {})
    print(sys.argv[1].split("/")[-1], round(d["value"] / 1e6, 1), "M/s", round(d["ms_per_step"] * 1e3, 2), "us/step kernel", round(d["roofline"]["kernel_ms_per_launch"] * 1e3, 2), rv.get("activity"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
timeout -k 10 300 python tools/flop_model.py counts --workload C4 --steps 1200 --out "$OUT/flopmodel/counts_C4.json"; echo "counts rc=$?"
cd /tmp
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU SQ_WAVES --output-format csv -d "$OUT/flopmodel/pmc_C4" -- python3 "$ROOT/tools/flop_model.py" run --workload C4 --steps 1200 > "$OUT/flopmodel/pmc.log" 2>&1; echo "pmc rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-count > "$OUT/trace.log" 2>&1; echo "trace rc=$?"
cd "$ROOT"
python tools/flop_model.py fit "$OUT/flopmodel" "$OUT/flop_model" | tail -15
find "$OUT/trace" -name "*kernel_stats.csv" -exec head -5 {} \;
