#!/bin/bash
# Round-5 session 23: three code-generation switches on the last kernels (no tail duplication / no machine sinking / -O2)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
bash tools/sessions/ab3.sh r5_s23/ab "head=gym-os2r_amd/ab/libos2r_head.so notaildup=gym-os2r_amd/ab/libos2r_notaildup.so nosink=gym-os2r_amd/ab/libos2r_nosink.so o2=gym-os2r_amd/ab/libos2r_o2.so" "--workload C4" "--workload C3"
