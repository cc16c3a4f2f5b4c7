#!/bin/bash
# Round-5 session 24: -disable-machine-sink for every kernel -- GPU suite, then A/B on the other workloads
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r5_s24
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -q -x > "$OUT/pytest.log" 2>&1; rc=$?; echo "pytest rc $rc"; grep -E "^FAILED|passed|failed|Error" "$OUT/pytest.log" | tail -8
[ $rc -eq 0 ] || exit $rc
bash tools/sessions/ab3.sh r5_s24/ab "head=gym-os2r_amd/ab/libos2r_head.so nosink=gym-os2r_amd/libos2r.so" "--workload C4" "--workload V1" "--workload C2" "--dtype f32" "--workload C4 --pgs-tol 1e-3" "--workload C4 --steps 20 --warmup 5"
