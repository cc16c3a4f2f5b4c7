#!/bin/bash
# Round-5 session 3: ABI 5 on the box -- the runtime GPU tests, then the host surface in the stationary regime
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r5_s3
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_runtime.py -m gpu -x -q > "$OUT/pytest_runtime.log" 2>&1; echo "pytest rc $?"; tail -5 "$OUT/pytest_runtime.log"
timeout -k 10 300 python tools/dbg/host_profile.py 65536 300 > "$OUT/host_profile_65536.txt" 2>&1; head -14 "$OUT/host_profile_65536.txt"
timeout -k 10 300 python tools/dbg/host_profile.py 64 400 > "$OUT/host_profile_64.txt" 2>&1; head -14 "$OUT/host_profile_64.txt"
timeout -k 10 400 python tools/dbg/host_overhead.py 65536 300 > "$OUT/host_surface.txt" 2> "$OUT/host_surface.err"; cat "$OUT/host_surface.txt"
