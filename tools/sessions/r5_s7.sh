#!/bin/bash
# Round-5 session 7: where the dual solve's time goes -- phase stamps of the small-path build (on the box), its counters
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r5_s7
mkdir -p "$OUT"
cd "$ROOT"
( time make -C gym-os2r_amd/csrc stamps -j16 > "$OUT/make_stamps.log" 2>&1 ) 2>&1 | grep real
timeout -k 10 300 python tools/dbg/stamps.py C4 1200 > "$OUT/stamps_C4.txt" 2>&1; head -36 "$OUT/stamps_C4.txt"
timeout -k 10 300 python bench.py --no-cpu-baseline --no-gym-level --steps 200 > "$OUT/bench.json" 2> "$OUT/bench.err"; python - <<PY
import json
d=json.load(open("$OUT/bench.json")); a=d.get("roofline_valu",{}).get("activity",{})
print("bench", round(d["value"]/1e6,1), "M/s kernel us", round(d["roofline"]["kernel_ms_per_launch"]*1e3,2), {k: round(v,3) for k,v in a.items()})
PY
