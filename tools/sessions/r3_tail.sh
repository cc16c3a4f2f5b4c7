#!/bin/bash
# round 3: what the slowest waves of a launch do differently (stamp build; exact finish and sweeps-only solver)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r3_tail
mkdir -p "$OUT"
cd "$ROOT"
make -C gym-os2r_amd/csrc stamps > "$OUT/make.log" 2>&1 || { tail -5 "$OUT/make.log"; exit 1; }
timeout -k 10 300 python tools/dbg/stamps.py C4 1200 > "$OUT/stamps_exact.txt" 2>&1 || { tail -5 "$OUT/stamps_exact.txt"; exit 1; }
grep -v "^  *dyn\|amdgpu.ids" "$OUT/stamps_exact.txt"
OS2R_PGS_ITERS=20 OS2R_PGS_EXACT=0 timeout -k 10 300 python tools/dbg/stamps.py C4 1200 > "$OUT/stamps_legacy.txt" 2>&1 || exit 1
grep -A14 "per-wave" "$OUT/stamps_legacy.txt"
