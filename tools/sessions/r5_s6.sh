#!/bin/bash
# Round-5 session 6: the dual solve of small free sets -- parity suite, then same-box A/B against the library without it
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r5_s6
mkdir -p "$OUT"
cd "$ROOT"
bash tools/sessions/ab3.sh r5_s6/ab "base=gym-os2r_amd/ab/libos2r_base.so small=gym-os2r_amd/libos2r.so" "--workload C4 --steps 20 --warmup 5" "--workload C4" "--workload C3" "--workload V1" "--pgs-tol 1e-3"
timeout -k 10 300 python bench.py --no-cpu-baseline > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; python - <<PY
import json
d=json.load(open("$OUT/bench_default.json")); a=d.get("roofline_valu",{}).get("activity",{})
print("default", round(d["value"]/1e6,1), "M/s kernel us", round(d["roofline"]["kernel_ms_per_launch"]*1e3,2), {k: round(v,3) for k,v in a.items()}, d["roofline_valu"]["counters"] if "roofline_valu" in d else None)
print("gym_level", d.get("gym_level",{}).get("frac_of_value"), d.get("gym_level",{}).get("us_per_step"), "rollout", d.get("rollout",{}).get("value"))
PY

timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1; echo "pytest rc $?"; grep -E "^FAILED|^ERROR|passed|failed" "$OUT/pytest.log" | tail -8
