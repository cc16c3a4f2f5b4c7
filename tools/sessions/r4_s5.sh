#!/bin/bash
# round 4, session 5: (1) launch time against the step count for C3 and C4 (is C3 stationary where the bench window lies?),
# (2) does rocprofv3's PC sampling work on this pool (first look at its output format)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r4_s5
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 300 python tools/dbg/regime.py C3 4000 > "$OUT/regime_C3.txt" 2>&1; tail -45 "$OUT/regime_C3.txt"
timeout -k 10 300 python tools/dbg/regime.py C4 3000 > "$OUT/regime_C4.txt" 2>&1; tail -35 "$OUT/regime_C4.txt"
export TMPDIR=/tmp
cd /tmp
for m in host_trap stochastic; do
  u=time; i=200
  [ $m = stochastic ] && { u=cycles; i=1048576; }
  timeout -k 10 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit $u --pc-sampling-method $m --pc-sampling-interval $i --kernel-trace --output-format csv -d "$OUT/pcs_$m" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-count --steps 50 --warmup 5 --preroll 300 > "$OUT/pcs_$m.log" 2>&1
  echo "pcs $m rc $?"; tail -3 "$OUT/pcs_$m.log"
  find "$OUT/pcs_$m" -type f | head; for f in $(find "$OUT/pcs_$m" -name "*pc_sampling*"); do wc -l $f; head -5 $f; done
done
du -sh "$OUT"
