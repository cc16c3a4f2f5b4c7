#!/bin/bash
# round 4, session 3: instruction counts and waits of round 3's kernels against the new ones (same box), stamps of the new ones
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r4_s3
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 300 python -m pytest tests/test_gpu_runtime.py -x -q -k "vec_env" > "$OUT/pytest_vec.log" 2>&1; echo "pytest vec rc $?"; tail -3 "$OUT/pytest_vec.log"
tools/sessions/pmc_ab.sh r4_s3 "r3=gym-os2r_amd/ab/libos2r_r3.so new=gym-os2r_amd/libos2r.so" --workload C3
tools/sessions/pmc_ab.sh r4_s3 "r3=gym-os2r_amd/ab/libos2r_r3.so new=gym-os2r_amd/libos2r.so" --workload C4
cd "$ROOT"
make -C gym-os2r_amd/csrc stamps -j16 > "$OUT/make.log" 2>&1 || { tail -5 "$OUT/make.log"; exit 1; }
for w in C3 C4; do
  timeout -k 10 300 python tools/dbg/stamps.py $w 1200 > "$OUT/stamps_$w.txt" 2>&1 || { tail -5 "$OUT/stamps_$w.txt"; exit 1; }
  grep -v "^  *dyn\|amdgpu.ids" "$OUT/stamps_$w.txt" | head -40
done
