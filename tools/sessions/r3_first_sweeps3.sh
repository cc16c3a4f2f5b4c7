#!/bin/bash
# round 3 timing experiment: sweeps before the first check of the exact finish for the robots with fewer dof
# (variants built with -DOS2R_EXACT_FIRST=k as libos2r_k<k>.so; the default library has 6)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r3_first_sweeps
mkdir -p "$OUT"
cd "$ROOT"
for k in ${FIRST_SWEEPS:-6 3 4 5}; do
  if [ $k = 6 ]; then unset OS2R_LIBRARY; else export OS2R_LIBRARY=$ROOT/gym-os2r_amd/libos2r_k$k.so; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-count --workload V1 > "$OUT/bench.json" 2> "$OUT/bench.err" && python -c "import json;d=json.load(open('$OUT/bench.json'));print('first sweeps $k V1:', round(d['value']/1e6,1), 'M/s', round(d['ms_per_step']*1e3,2), 'us/step')"
  OS2R_SKIP=free_hip,simple timeout -k 10 300 python tools/dbg/all_modes.py 2>/dev/null | grep "f64" | sed "s/^/first sweeps $k: /"
done
