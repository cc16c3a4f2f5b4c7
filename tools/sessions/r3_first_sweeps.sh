#!/bin/bash
# round 3 timing experiment: ordinary sweeps before the first check of the exact finish (kExactFirst = 3, 4, 5), same box.
# The variants are built beforehand:  make -C gym-os2r_amd/csrc BUILD=build_k3 OUT=../libos2r_k3.so CXXFLAGS="... -DOS2R_EXACT_FIRST=3" ../libos2r_k3.so
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r3_first_sweeps
mkdir -p "$OUT"
cd "$ROOT"
for r in 1 2; do for k in ${FIRST_SWEEPS:-4 3 5}; do
  if [ $k = 4 ]; then unset OS2R_LIBRARY; else export OS2R_LIBRARY=$ROOT/gym-os2r_amd/libos2r_k$k.so; fi
  for w in C4 C3; do
    timeout -k 10 300 python bench.py --no-cpu-baseline --workload $w > "$OUT/bench.json" 2> "$OUT/bench.err" || { echo "k=$k failed"; tail -3 "$OUT/bench.err"; continue; }
    python -c "import json;d=json.load(open('$OUT/bench.json'));a=d.get('roofline_valu',{}).get('activity',{});print('first sweeps $k, $w:', round(d['value']/1e6,1), 'M/s', round(d['ms_per_step']*1e3,2), 'us/step; sweeps', round(a.get('phase2_sweeps_per_wave_iteration',0),2), 'solves', round(a.get('exact_solves_per_wave_iteration',0),2))"
  done
done; done
