#!/bin/bash
# round 3 timing experiment: sweeps before the first check of the exact finish (default 7; variants built with
# -DOS2R_EXACT_FIRST=k as libos2r_k<k>.so) on the workloads where the mean wave counts more than the slowest one
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r3_first_sweeps
mkdir -p "$OUT"
cd "$ROOT"
for k in ${FIRST_SWEEPS:-7 4 5 6}; do
  if [ $k = 7 ]; then unset OS2R_LIBRARY; else export OS2R_LIBRARY=$ROOT/gym-os2r_amd/libos2r_k$k.so; fi
  for w in "--workload C4" "--workload V1" "--workload C3" "--envs-per-gpu 131072 --steps 500 --splits 4" "--envs-per-gpu 524288 --steps 200 --preroll 600" "--workload C4 --splits 4"; do
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-count $w > "$OUT/bench.json" 2> "$OUT/bench.err" || { echo "k=$k failed"; tail -3 "$OUT/bench.err"; continue; }
    python -c "import json;d=json.load(open('$OUT/bench.json'));print('first sweeps $k [$w]:', round(d['value']/1e6,1), 'M/s', round(d['ms_per_step']*1e3,2), 'us/step')"
  done
done
