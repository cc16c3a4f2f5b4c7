#!/bin/bash
# stamp build of the current tree and the per-phase ticks of the given workloads: tools/sessions/stamps_only.sh OUT "C4 C3"
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/$1
mkdir -p "$OUT"
cd "$ROOT"
make -C gym-os2r_amd/csrc stamps -j16 > "$OUT/make.log" 2>&1 || { tail -5 "$OUT/make.log"; exit 1; }
for w in ${2:-C4}; do
  timeout -k 10 300 python tools/dbg/stamps.py $w 1200 > "$OUT/stamps_$w.txt" 2>&1 || { tail -5 "$OUT/stamps_$w.txt"; exit 1; }
  grep -v "^  *dyn\|amdgpu.ids" "$OUT/stamps_$w.txt" | head -36
done
