#!/bin/bash
# Round-5 session 12: scheduler strategies of the compiler for the C4 / C3 unit, same box
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
L=""
for v in equil ilp trk bias0 hrp lowocc memc; do L="$L $v=gym-os2r_amd/ab/libos2r_$v.so"; done
bash tools/sessions/ab3.sh r5_s12/ab "$L" "--workload C4" "--workload C3"
