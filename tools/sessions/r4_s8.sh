#!/bin/bash
# round 4, session 8: when the waves of a launch start and end and where they run (light stamp build = the shipped code), C3 and C4
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r4_s8
mkdir -p "$OUT"
cd "$ROOT"
make -C gym-os2r_amd/csrc stamps_light -j16 > "$OUT/make.log" 2>&1 || { tail -5 "$OUT/make.log"; exit 1; }
for w in C3 C4; do
  timeout -k 10 300 python tools/dbg/wave_times.py $w 1200 > "$OUT/wave_times_$w.txt" 2>&1 || { tail -5 "$OUT/wave_times_$w.txt"; exit 1; }
  grep -v "amdgpu.ids" "$OUT/wave_times_$w.txt"
done
OS2R_PGS_TOL=1e-3 timeout -k 10 300 python tools/dbg/wave_times.py C3 1200 > "$OUT/wave_times_C3_floor.txt" 2>&1; grep -v "amdgpu.ids" "$OUT/wave_times_C3_floor.txt"
