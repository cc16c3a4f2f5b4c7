#!/bin/bash
# round 4, session 11: where the cycles of an exact solve go (stamps inside it), C4 and C3; sanity: the tree's kernels = the base kernels
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r4_s11
mkdir -p "$OUT"
cd "$ROOT"
make -C gym-os2r_amd/csrc stamps -j16 > "$OUT/make.log" 2>&1 || { tail -5 "$OUT/make.log"; exit 1; }
for w in C4 C3; do
  timeout -k 10 300 python tools/dbg/stamps.py $w 1200 > "$OUT/stamps_$w.txt" 2>&1; grep -v "^  *dyn\|amdgpu.ids" "$OUT/stamps_$w.txt" | head -42
done
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 300 > "$OUT/bench_C4.json" 2>/dev/null; python -c "import json;d=json.load(open('$OUT/bench_C4.json'));print('C4', round(d['value']/1e6,1), 'M/s', d['roofline_valu']['activity'])"
