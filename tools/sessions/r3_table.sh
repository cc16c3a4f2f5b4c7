#!/bin/bash
# round 3: the shards column of DESIGN.md 7's table (bench.py --splits 4 on the other workloads and sizes), every task mode
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r3_table
mkdir -p "$OUT"
cd "$ROOT"
run() {
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-count "$@" > "$OUT/bench.json" 2> "$OUT/bench.err" || { echo "bench $* failed"; tail -5 "$OUT/bench.err"; return; }
  python -c "import json;d=json.load(open('$OUT/bench.json'));print('[$*]', round(d['value']/1e6,1), 'M/s', round(d['ms_per_step']*1e3,2), 'us/step; launch', round(d['roofline']['kernel_ms_per_launch']*1e3,2), 'us')"
}
run --splits 4 --steps 20 --warmup 5
run --splits 4 --envs-per-gpu 131072 --steps 500
run --splits 4 --envs-per-gpu 524288 --steps 200 --preroll 600
run --splits 4 --workload C3
run --splits 4 --workload V1
run --splits 4 --runtime-model
run --splits 4 --dtype f32
run --splits 4 --dtype f32 --envs-per-gpu 131072 --steps 500
run --splits 4 --dtype f32 --envs-per-gpu 262144 --steps 300
python tools/dbg/all_modes.py > "$OUT/all_modes.txt" 2>&1; cat "$OUT/all_modes.txt"
