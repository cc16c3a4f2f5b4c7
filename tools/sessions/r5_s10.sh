#!/bin/bash
# Round-5 session 10: row-equilibrated solve and the kernel-top reorder against the round-4 structure (same box), then the GPU suite
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
mkdir -p gpurun_out/r5_s10
bash tools/sessions/ab3.sh r5_s10/ab "base=gym-os2r_amd/ab/libos2r_base.so equil=gym-os2r_amd/ab/libos2r_equil.so equil_top=gym-os2r_amd/ab/libos2r_equil_top.so" "--workload C4" "--workload C4 --steps 20 --warmup 5" "--workload C3" "--workload V1" "--pgs-tol 1e-3"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r5_s10/pytest.log 2>&1; echo "pytest rc $?"; grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r5_s10/pytest.log | tail -8
