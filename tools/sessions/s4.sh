#!/bin/bash
# GPU session 4: fp32 kernels after the fast reciprocal / rsqrt / sincos
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/s4
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "f32" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest.log"; grep -E "^\[|^\.\[" "$OUT/pytest.log"
B="timeout -k 10 200 python bench.py --no-cpu-baseline"
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[1].split("/")[-1], round(d["value"] / 1e6, 1), "M/s", round(d["ms_per_step"] * 1e3, 2), "us/step")
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
OS2R_F32_WAVES=1 $B --dtype f32 > "$OUT/b_f32_64k_w1.json" 2>/dev/null; show "$OUT/b_f32_64k_w1.json"
OS2R_F32_WAVES=2 $B --dtype f32 > "$OUT/b_f32_64k_w2.json" 2>/dev/null; show "$OUT/b_f32_64k_w2.json"
$B --dtype f32 --pgs-tol 1e-13 > "$OUT/b_f32_64k_tol.json" 2>/dev/null; show "$OUT/b_f32_64k_tol.json"
$B --dtype f32 --envs-per-gpu 131072 --steps 500 > "$OUT/b_f32_128k.json" 2>/dev/null; show "$OUT/b_f32_128k.json"
$B --dtype f32 --envs-per-gpu 131072 --steps 500 --pgs-tol 1e-13 > "$OUT/b_f32_128k_tol.json" 2>/dev/null; show "$OUT/b_f32_128k_tol.json"
$B --dtype f32 --envs-per-gpu 262144 --steps 300 > "$OUT/b_f32_256k.json" 2>/dev/null; show "$OUT/b_f32_256k.json"
$B > "$OUT/b_f64.json" 2>/dev/null; show "$OUT/b_f64.json"
