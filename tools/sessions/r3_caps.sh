#!/bin/bash
# round 3: the exact finish on the bench workload with different caps on the solves / tolerances (all on the kernels
# with the default solver settings: the cap and the tolerance are run-time values there)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r3_caps
mkdir -p "$OUT"
cd "$ROOT"
for v in "" "--pgs-exact 6" "--pgs-exact 3" "--pgs-exact 2" "--pgs-exact 1" "--pgs-tol 1e-3" "--pgs-tol 1e-16" "--pgs-tol 1e-20"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline $v > "$OUT/bench.json" 2> "$OUT/bench.err" || { echo "bench $v failed"; tail -5 "$OUT/bench.err"; exit 1; }
  python -c "import json;d=json.load(open('$OUT/bench.json'));a=d.get('roofline_valu',{}).get('activity',{});print('[$v]', round(d['value']/1e6,1), 'M/s', round(d['roofline']['kernel_ms_per_launch']*1e3,2), 'us', 'sweeps', round(a.get('phase2_sweeps_per_wave_iteration',0),2), 'solves', round(a.get('exact_solves_per_wave_iteration',0),2), 'envs/solve', round(a.get('envs_per_exact_solve',0),2))"
done
