#!/bin/bash
# round 3: bench.py --splits S (shards of the batch on S streams, C-side enqueueing) with the default and with more hardware queues
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r3_splits
mkdir -p "$OUT"
cd "$ROOT"
run() {
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-count "$@" > "$OUT/bench.json" 2> "$OUT/bench.err" || { echo "bench $* failed"; tail -5 "$OUT/bench.err"; return; }
  python -c "import json;d=json.load(open('$OUT/bench.json'));print('[$*] queues=${GPU_MAX_HW_QUEUES:-default}', round(d['value']/1e6,1), 'M/s', round(d['ms_per_step']*1e3,2), 'us/step; launch', round(d['roofline']['kernel_ms_per_launch']*1e3,2), 'us')"
}
for s in 1 2 3 4 6 8; do run --splits $s; done
export GPU_MAX_HW_QUEUES=8
for s in 4 6 8 12; do run --splits $s; done
unset GPU_MAX_HW_QUEUES
for s in 1 4; do run --splits $s --pgs-exact 0 --pgs-iters 20; done
