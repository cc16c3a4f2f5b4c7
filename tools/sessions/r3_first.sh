#!/bin/bash
# round 3, first GPU session of the exact finish: smoke against the oracle, then same-box A/B of the bench workload
# with the exact finish (default) and with the round-2 solver (--pgs-exact 0 --pgs-iters 20)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r3_first
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1 || { echo "smoke failed"; tail -20 "$OUT/smoke.log"; exit 1; }
tail -1 "$OUT/smoke.log"
for r in 1 2; do
  for v in exact legacy; do
    if [ $v = legacy ]; then EXTRA="--pgs-exact 0 --pgs-iters 20"; else EXTRA=""; fi
    timeout -k 10 300 python bench.py --no-cpu-baseline $EXTRA > "$OUT/bench_$v.json" 2> "$OUT/bench_$v.err" || { echo "bench $v failed"; tail -5 "$OUT/bench_$v.err"; exit 1; }
    python -c "import json;d=json.load(open('$OUT/bench_$v.json'));a=d.get('roofline_valu',{}).get('activity',{});print('$v', round(d['value']/1e6,1), 'M/s', round(d['roofline']['kernel_ms_per_launch']*1e3,2), 'us', {k:round(v,2) for k,v in a.items()})"
  done
done
