#!/bin/bash
# round 4, session 15: timing experiment -- the pair fix in the first phase-2 sweep (kernel only: parity tests are not run), against the tree's kernels
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r4_s15
mkdir -p "$OUT"
cd "$ROOT"
for lv in base=gym-os2r_amd/ab/libos2r_base.so pair=gym-os2r_amd/libos2r.so; do
  label=${lv%%=*}; lib=${lv#*=}
  for w in C4 C3 V1; do
    OS2R_LIBRARY=$ROOT/$lib timeout -k 10 300 python bench.py --no-cpu-baseline --workload $w --steps 500 > "$OUT/c_${label}_$w.json" 2>/dev/null
    python -c "import json;d=json.load(open('$OUT/c_${label}_$w.json'));a=d['roofline_valu']['activity'];print('$label $w', round(d['value']/1e6,1),'M/s', round(d['roofline']['kernel_ms_per_launch']*1e3,2),'us | sweeps',round(a['phase2_sweeps_per_wave_iteration'],2),'solves',round(a['exact_solves_per_wave_iteration'],3),'lanes/solve',round(a['envs_per_exact_solve'],2))" | tee -a "$OUT/counts.txt"
  done
done
tools/sessions/ab3.sh r4_s15 "base=gym-os2r_amd/ab/libos2r_base.so pair=gym-os2r_amd/libos2r.so" "--workload C4" "--workload C3 --steps 500" "--workload V1 --steps 500" "--workload C4 --steps 20 --warmup 5"
