#!/bin/bash
# Round-5 session 19: candidate scan in chunks of six (scalar loads: 144 cycles of arithmetic per request instead of 96) -- GPU suite + A/B against beb33cb
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r5_s19
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -q -x > "$OUT/pytest.log" 2>&1; rc=$?; echo "pytest rc $rc"; grep -E "^FAILED|passed|failed|Error" "$OUT/pytest.log" | tail -8
[ $rc -eq 0 ] || exit $rc
bash tools/sessions/ab3.sh r5_s19/ab "rot=gym-os2r_amd/ab/libos2r_rot.so ch6=gym-os2r_amd/libos2r.so" "--workload C4" "--workload C3" "--workload V1" "--workload C4 --pgs-tol 1e-3"
