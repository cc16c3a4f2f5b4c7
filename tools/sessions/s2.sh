#!/bin/bash
# GPU session 2: wave-level work counters, stopping rule at larger batches, phase stamps, flop model refit
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/s2
mkdir -p "$OUT/flopmodel"
export TMPDIR=/tmp
cd "$ROOT"
B="timeout -k 10 200 python bench.py --no-cpu-baseline"
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); rv = d.get("roofline_valu", {}); a = rv.get("activity") or {}
    print(sys.argv[1].split("/")[-1], round(d["value"] / 1e6, 1), "M/s", round(d["ms_per_step"] * 1e3, 2), "us/step; sweeps/wave-iter", round(a.get("phase2_sweeps_per_wave_iteration", 0), 2), "live envs", round(a.get("live_envs_per_sweep", 0), 1), "frac", rv.get("frac"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
$B > "$OUT/b_64k.json" 2>/dev/null; show "$OUT/b_64k.json"
$B --pgs-tol 0 > "$OUT/b_64k_tol0.json" 2>/dev/null; show "$OUT/b_64k_tol0.json"
$B --envs-per-gpu 131072 --steps 500 > "$OUT/b_128k.json" 2>/dev/null; show "$OUT/b_128k.json"
$B --envs-per-gpu 131072 --steps 500 --pgs-tol 0 > "$OUT/b_128k_tol0.json" 2>/dev/null; show "$OUT/b_128k_tol0.json"
$B --envs-per-gpu 524288 --steps 200 --preroll 600 > "$OUT/b_512k.json" 2>/dev/null; show "$OUT/b_512k.json"
$B --envs-per-gpu 524288 --steps 200 --preroll 600 --pgs-tol 0 > "$OUT/b_512k_tol0.json" 2>/dev/null; show "$OUT/b_512k_tol0.json"
$B --pgs-iters 12 > "$OUT/b_64k_cap12.json" 2>/dev/null; show "$OUT/b_64k_cap12.json"
timeout -k 10 200 python tools/dbg/stamps.py C4 1100 > "$OUT/stamps.txt" 2>&1; tail -32 "$OUT/stamps.txt"
timeout -k 10 300 python tools/flop_model.py counts --workload C4 --steps 1200 --out "$OUT/flopmodel/counts_C4.json"; echo "counts rc=$?"
cp -r gpurun_out/s1/flopmodel/pmc_C4 "$OUT/flopmodel/" 2>/dev/null || (cd /tmp && timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU SQ_WAVES --output-format csv -d "$OUT/flopmodel/pmc_C4" -- python3 "$ROOT/tools/flop_model.py" run --workload C4 --steps 1200 > "$OUT/flopmodel/pmc.log" 2>&1; echo "pmc rc=$?")
cd "$ROOT"; python tools/flop_model.py fit "$OUT/flopmodel" "$OUT/flop_model" | tail -12
