#!/bin/bash
# Round-5 session 1: the surface users call (tools/dbg/host_overhead.py) and this round's baseline lines on this box.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r5_s1
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 400 python tools/dbg/host_overhead.py 65536 300 > "$OUT/host_surface.txt" 2> "$OUT/host_surface.err"; echo "host rc $?"; cat "$OUT/host_surface.txt"; tail -3 "$OUT/host_surface.err"
B="timeout -k 10 200 python bench.py --no-cpu-baseline --no-count"
for w in "--steps 20 --warmup 5" "" "--workload C3" "--pgs-tol 1e-3"; do
  $B $w > "$OUT/b.json" 2>"$OUT/b.err" && python -c "import json;d=json.load(open('$OUT/b.json'));print('[$w]', round(d['value']/1e6,1), 'M/s', round(d['ms_per_step']*1e3,2), 'us/step kernel', round(d['roofline']['kernel_ms_per_launch']*1e3,2))" | tee -a "$OUT/table.txt"
done
