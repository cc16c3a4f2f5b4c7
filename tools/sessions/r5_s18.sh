#!/bin/bash
# Round-5 session 18: the exact-finish loop rotated by hand (vote at the bottom) -- GPU suite + A/B against commit e17cb2b
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r5_s18
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -q -x > "$OUT/pytest.log" 2>&1; rc=$?; echo "pytest rc $rc"; grep -E "^FAILED|passed|failed|Error" "$OUT/pytest.log" | tail -8
[ $rc -eq 0 ] || exit $rc
bash tools/sessions/ab3.sh r5_s18/ab "touch=gym-os2r_amd/ab/libos2r_touch.so rotated=gym-os2r_amd/libos2r.so" "--workload C4" "--workload C3" "--workload V1"
