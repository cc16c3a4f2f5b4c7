#!/bin/bash
# round 4, session 12: the exact solve's row mask (one scalar bit mask per solve, whole bodies / the joint rows skipped with one branch)
# against the base kernels on one box; the GPU suite
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/${SESSION:-r4_s12}
mkdir -p "$OUT"
cd "$ROOT"
tools/sessions/ab3.sh ${SESSION:-r4_s12} "base=gym-os2r_amd/ab/libos2r_base.so new=gym-os2r_amd/libos2r.so" "--workload C4" "--workload C3 --steps 500" "--workload V1 --steps 500" "--workload C4 --steps 20 --warmup 5"
timeout -k 10 900 python -m pytest tests -m gpu -q -x > "$OUT/pytest.log" 2>&1; echo "pytest rc $?"; grep -E "^FAILED|passed|failed" "$OUT/pytest.log" | tail -15
