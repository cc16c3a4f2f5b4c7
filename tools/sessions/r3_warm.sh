#!/bin/bash
# round 3: the warm start between the physics iterations of an env-step -- sweeps before the first check after a warm start
# (default library: 3; libos2r_w2.so: 2), GPU suite first
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r3_warm
mkdir -p "$OUT"
cd "$ROOT"
python -m pytest tests -x -q -m gpu > "$OUT/pytest_gpu.log" 2>&1; tail -3 "$OUT/pytest_gpu.log"
for r in 1 2; do for k in 3 2; do
  if [ $k = 3 ]; then unset OS2R_LIBRARY; else export OS2R_LIBRARY=$ROOT/gym-os2r_amd/libos2r_w$k.so; fi
  for w in "--workload C4" "--workload C4 --steps 20 --warmup 5" "--workload V1" "--workload C3" "--envs-per-gpu 131072 --steps 500 --splits 4"; do
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-count $w > "$OUT/bench.json" 2> "$OUT/bench.err" || { echo "w=$k failed"; tail -3 "$OUT/bench.err"; continue; }
    python -c "import json;d=json.load(open('$OUT/bench.json'));print('warm first $k [$w]:', round(d['value']/1e6,1), 'M/s', round(d['ms_per_step']*1e3,2), 'us/step')"
  done
done; done
