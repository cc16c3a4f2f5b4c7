#!/bin/bash
# same-box A/B of two builds on the bench workload: A = gym-os2r_amd/libos2r.so, B = gym-os2r_amd/libos2r_ab.so (OS2R_LIBRARY);
# the default window and the driver's 20-step window, alternating
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/ab2
mkdir -p "$OUT"
cd "$ROOT"
ARGS="${BENCH_ARGS:-}"
for r in 1 2 3; do for v in A B; do
  if [ $v = B ]; then export OS2R_LIBRARY=$ROOT/gym-os2r_amd/libos2r_ab.so; else unset OS2R_LIBRARY; fi
  for w in "" "--steps 20 --warmup 5"; do
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-count $ARGS $w > "$OUT/bench_$v.json" 2>/dev/null
    python -c "import json;d=json.load(open('$OUT/bench_$v.json'));print('$v [$w]', round(d['value']/1e6,1), 'M/s', round(d['ms_per_step']*1e3,2), 'us/step')"
  done
done; done
