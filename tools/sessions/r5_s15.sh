#!/bin/bash
# Round-5 session 15: the remaining wave votes of the exact finish as scalar arithmetic (liveness from data, touch masks) -- GPU suite + A/B
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r5_s15
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -q -x > "$OUT/pytest.log" 2>&1; rc=$?; echo "pytest rc $rc"; grep -E "^FAILED|passed|failed|Error" "$OUT/pytest.log" | tail -8
[ $rc -eq 0 ] || exit $rc
bash tools/sessions/ab3.sh r5_s15/ab "prox2=gym-os2r_amd/ab/libos2r_prox2.so votes=gym-os2r_amd/libos2r.so" "--workload C4" "--workload C3" "--workload V1"
