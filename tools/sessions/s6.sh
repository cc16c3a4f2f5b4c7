#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/s6; mkdir -p "$OUT"; cd "$ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -2 "$OUT/pytest.log"
for a in "--workload V1" "--workload C4" "--workload C3" "--workload C2" "--workload V1 --dtype f32 --envs-per-gpu 131072"; do timeout -k 10 200 python bench.py --no-cpu-baseline --no-count $a > /tmp/b.json 2>/dev/null; python -c "import json;d=json.load(open('/tmp/b.json'));print('r02 $a', round(d['value']/1e6,1), round(d['ms_per_step']*1e3,2))"; done
cd build_dbg/r01 && for a in "--workload V1" "--workload V1 --warmup 1100" "--workload C4 --warmup 1100" "--workload C2 --warmup 1100"; do timeout -k 10 200 python bench.py --no-cpu-baseline $a > /tmp/b.json 2>/dev/null; python -c "import json;d=json.load(open('/tmp/b.json'));print('r01 $a', round(d['value']/1e6,1), round(d['ms_per_step']*1e3,2))"; done
