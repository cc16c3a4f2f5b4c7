#!/bin/bash
# round 4, session 1: GPU tests of the new ABI (solver state, rollout), then round 3's kernels against the new ones on one box
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/${SESSION:-r4_s1}
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1; echo "pytest rc $?"; tail -5 "$OUT/pytest.log"
tools/sessions/ab3.sh ${SESSION:-r4_s1} "r3=gym-os2r_amd/ab/libos2r_r3.so new=gym-os2r_amd/libos2r.so" "--workload C4" "--workload C4 --steps 20 --warmup 5" "--workload C3" "--workload V1"
for K in 10 50; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --rollout $K --steps 1000 > "$OUT/rollout_$K.json" 2>"$OUT/rollout_$K.err" && python -c "import json;d=json.load(open('$OUT/rollout_$K.json'));print('rollout $K', round(d['value']/1e6,1), 'M/s', round(d['ms_per_step']*1e3,2), 'us/step')" | tee -a "$OUT/table.txt"
done
