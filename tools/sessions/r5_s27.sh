#!/bin/bash
# Round-5 session 27: two normal sweeps instead of three fix the friction box (default configuration; compiled-in standard solver) --
# GPU suite, then the same-box A/B.  A = the kernels of commit 0f-c3profile run with --pgs-normal-iters 3 (their standard solver),
# B = the working tree with its default (2)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r5_s27
mkdir -p "$OUT/ab"
cd "$ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -q -x > "$OUT/pytest.log" 2>&1; rc=$?; echo "pytest rc $rc"; grep -E "^FAILED|passed|failed|Error" "$OUT/pytest.log" | tail -8
[ $rc -eq 0 ] || exit $rc
for r in 1 2 3; do
  for w in "--workload C4" "--workload C3" "--workload V1" "--workload C4 --pgs-tol 1e-3" "--workload C4 --steps 20 --warmup 5"; do
    OS2R_LIBRARY=$ROOT/gym-os2r_amd/ab/libos2r_head.so timeout -k 10 300 python bench.py --no-cpu-baseline --no-count --no-gym-level --pgs-normal-iters 3 $w > "$OUT/ab/bench_three.json" 2>"$OUT/ab/err_three.log" || { echo "FAILED three [$w]"; tail -3 "$OUT/ab/err_three.log"; exit 1; }
    python -c "import json;d=json.load(open('$OUT/ab/bench_three.json'));print('three [$w]', round(d['value']/1e6,1), 'M/s', round(d['ms_per_step']*1e3,2), 'us/step', 'kernel', round(d['roofline']['kernel_ms_per_launch']*1e3,2), 'us')" | tee -a "$OUT/ab/table.txt"
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-count --no-gym-level $w > "$OUT/ab/bench_two.json" 2>"$OUT/ab/err_two.log" || { echo "FAILED two [$w]"; tail -3 "$OUT/ab/err_two.log"; exit 1; }
    python -c "import json;d=json.load(open('$OUT/ab/bench_two.json'));print('two   [$w]', round(d['value']/1e6,1), 'M/s', round(d['ms_per_step']*1e3,2), 'us/step', 'kernel', round(d['roofline']['kernel_ms_per_launch']*1e3,2), 'us')" | tee -a "$OUT/ab/table.txt"
  done
done
