#!/bin/bash
# same-box A/B of builds of the same ABI on bench workloads, alternating:
#   tools/sessions/ab3.sh OUTDIR "LABEL=path/to/lib.so ..." "WORKLOAD ARGS" ["WORKLOAD ARGS" ...]
# e.g. tools/sessions/ab3.sh ab_r4 "r3=gym-os2r_amd/ab/libos2r_r3.so new=gym-os2r_amd/libos2r.so" "--workload C4" "--workload C4 --steps 20 --warmup 5" "--workload C3"
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/$1; shift
LIBS=$1; shift
mkdir -p "$OUT"
cd "$ROOT"
for r in 1 2 3; do
  for w in "$@"; do
    for lv in $LIBS; do
      label=${lv%%=*}; lib=${lv#*=}
      OS2R_LIBRARY=$ROOT/$lib timeout -k 10 300 python bench.py --no-cpu-baseline --no-count --no-gym-level $w > "$OUT/bench_$label.json" 2>"$OUT/err_$label.log" || { echo "FAILED $label [$w]"; tail -3 "$OUT/err_$label.log"; exit 1; }
      python -c "import json;d=json.load(open('$OUT/bench_$label.json'));print('$label [$w]', round(d['value']/1e6,1), 'M/s', round(d['ms_per_step']*1e3,2), 'us/step', 'kernel', round(d['roofline']['kernel_ms_per_launch']*1e3,2), 'us')" | tee -a "$OUT/table.txt"
    done
  done
done
