#!/bin/bash
# same-box A/B of the tree's library against gym-os2r_amd/ab/libos2r_base.so on the bench workloads, then the GPU suite (-x):
#   tools/sessions/r4_ab.sh OUTDIR
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/${1:-r4_ab}
mkdir -p "$OUT"
cd "$ROOT"
tools/sessions/ab3.sh ${1:-r4_ab} "base=gym-os2r_amd/ab/libos2r_base.so new=gym-os2r_amd/libos2r.so" "--workload C4" "--workload C3 --steps 500" "--workload V1 --steps 500" "--workload C4 --steps 20 --warmup 5" "--workload C4 --pgs-tol 1e-3 --steps 500"
timeout -k 10 900 python -m pytest tests -m gpu -q -x > "$OUT/pytest.log" 2>&1; echo "pytest rc $?"; grep -E "^FAILED|passed|failed" "$OUT/pytest.log" | tail -5
