#!/bin/bash
# same-box A/B: A = gym-os2r_amd/libos2r.so, B = gym-os2r_amd/libos2r_ab.so (OS2R_LIBRARY); parity tests on B first
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/ab
mkdir -p "$OUT"
cd "$ROOT"
ARGS="${BENCH_ARGS:-}"
OS2R_LIBRARY=$ROOT/gym-os2r_amd/libos2r_ab.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "${AB_TESTS:-one_step or fallen or trajectory_1000 or random_action or stopping or permut or shard}" > "$OUT/pytest_b.log" 2>&1; echo "pytest(B) rc=$?"; tail -2 "$OUT/pytest_b.log"
for r in 1 2 3; do for v in A B; do
  if [ $v = B ]; then export OS2R_LIBRARY=$ROOT/gym-os2r_amd/libos2r_ab.so; else unset OS2R_LIBRARY; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-count $ARGS > "$OUT/bench_$v.json" 2>/dev/null
  python -c "import json;d=json.load(open('$OUT/bench_$v.json'));print('$v', round(d['value']/1e6,1), 'M/s', round(d['roofline']['kernel_ms_per_launch']*1e3,2), 'us')"
done; done
