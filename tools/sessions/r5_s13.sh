#!/bin/bash
# Round-5 session 13: two proximal iterations (specification) + the solve's flags as scalar lane masks -- GPU suite, then same-box A/B
# against the kernels of the previous commit (three proximal iterations)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r5_s13
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -q -x > "$OUT/pytest.log" 2>&1; rc=$?; echo "pytest rc $rc"; grep -E "^FAILED|passed|failed|Error" "$OUT/pytest.log" | tail -8
[ $rc -eq 0 ] || exit $rc
bash tools/sessions/ab3.sh r5_s13/ab "prox3=gym-os2r_amd/ab/libos2r_base.so prox2=gym-os2r_amd/libos2r.so" "--workload C4" "--workload C3" "--workload V1"
