#!/bin/bash
# GPU session 3: full GPU suite with the round-2 tests, f32 builds, profile of the default bench (trace + PMC), clock
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/s3
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest.log"; grep -E "^\[" "$OUT/pytest.log" | tail -30
B="timeout -k 10 200 python bench.py --no-cpu-baseline"
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); rv = d.get("roofline_valu", {}); a = rv.get("activity") or {}
    print(sys.argv[1].split("/")[-1], round(d["value"] / 1e6, 1), "M/s", round(d["ms_per_step"] * 1e3, 2), "us/step; frac", rv.get("frac"), "flops/env-step", rv.get("flops_per_env_step"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
$B > "$OUT/b_f64.json" 2>/dev/null; show "$OUT/b_f64.json"
$B --steps 20 --warmup 5 > "$OUT/b_f64_20_5.json" 2>/dev/null; show "$OUT/b_f64_20_5.json"
$B --dtype f32 > "$OUT/b_f32_64k.json" 2>/dev/null; show "$OUT/b_f32_64k.json"
$B --dtype f32 --envs-per-gpu 131072 --steps 500 > "$OUT/b_f32_128k.json" 2>/dev/null; show "$OUT/b_f32_128k.json"
$B --dtype f32 --envs-per-gpu 262144 --steps 300 > "$OUT/b_f32_256k.json" 2>/dev/null; show "$OUT/b_f32_256k.json"
$B --envs-per-gpu 131072 --steps 500 > "$OUT/b_f64_128k.json" 2>/dev/null; show "$OUT/b_f64_128k.json"
OS2R_CLOCK_JSON="$OUT/clock.json" timeout -k 10 200 python tools/dbg/stamps.py C4 1100 > "$OUT/stamps.txt" 2>&1; grep -E "clock|stamp build|per-wave" "$OUT/stamps.txt"
bash tools/profile.sh r02 --no-count > "$OUT/profile.log" 2>&1; echo "profile rc=$?"
python tools/summarize_profile.py gpurun_out/prof_r02 "$OUT/r02_step_kernel_f64_C4" step_kernel "$OUT/traffic.json" "$OUT/clock.json" > /dev/null; cat "$OUT/r02_step_kernel_f64_C4.md" | head -40
