#!/bin/bash
# Round-5 session 11: the GPU suite with the row-equilibrated solve (all failures, not the first)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
mkdir -p gpurun_out/r5_s11
timeout -k 10 1000 python -m pytest tests -m gpu -q -s > gpurun_out/r5_s11/pytest.log 2>&1; echo "pytest rc $?"; grep -E "^FAILED|^ERROR|passed|failed|traj|rel err|one step|steady" gpurun_out/r5_s11/pytest.log | cut -c1-300 | tail -40
