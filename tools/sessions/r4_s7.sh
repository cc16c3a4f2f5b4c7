#!/bin/bash
# round 4, session 7: what the rare many-solve environment-iterations cost a launch: the cap on the solves per physics
# iteration (a run-time value of the same kernel) swept on C3 and C4
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r4_s7
mkdir -p "$OUT"
cd "$ROOT"
for w in C3 C4; do
  for k in 12 6 4 3 2 1; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-count --workload $w --pgs-exact $k --steps 500 > "$OUT/b.json" 2>"$OUT/b.err" || { tail -3 "$OUT/b.err"; continue; }
    python -c "import json;d=json.load(open('$OUT/b.json'));print('$w pgs_exact $k', round(d['value']/1e6,1), 'M/s', round(d['roofline']['kernel_ms_per_launch']*1e3,2), 'us')" | tee -a "$OUT/table.txt"
  done
done
