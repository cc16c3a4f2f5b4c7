#!/bin/bash
# round 4, session 10: lagged friction box / phase-1 skip against the base kernels on one box (floor, C4, C3, V1), the GPU suite without -x,
# the stamps of the new kernels
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r4_s10
mkdir -p "$OUT"
cd "$ROOT"
tools/sessions/ab3.sh r4_s10 "base=gym-os2r_amd/ab/libos2r_base.so new=gym-os2r_amd/libos2r.so" "--workload C4 --pgs-tol 1e-3 --steps 500" "--workload C4" "--workload C3 --steps 500" "--workload V1 --steps 500"
timeout -k 10 900 python -m pytest tests -m gpu -q > "$OUT/pytest.log" 2>&1; echo "pytest rc $?"; grep -E "^FAILED|passed|failed" "$OUT/pytest.log" | tail -15
make -C gym-os2r_amd/csrc stamps -j16 > "$OUT/make.log" 2>&1 || { tail -5 "$OUT/make.log"; exit 1; }
timeout -k 10 300 python tools/dbg/stamps.py C4 1200 > "$OUT/stamps_C4.txt" 2>&1; grep -v "^  *dyn\|amdgpu.ids" "$OUT/stamps_C4.txt" | head -24
