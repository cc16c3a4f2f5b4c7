#!/bin/bash
# Round-5 session 26: register-allocator switches (regclass priority over globalness / reverse local assignment)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
bash tools/sessions/ab3.sh r5_s26/ab "head=gym-os2r_amd/ab/libos2r_head.so rcprio=gym-os2r_amd/ab/libos2r_r1.so revlocal=gym-os2r_amd/ab/libos2r_r2.so" "--workload C4" "--workload C3"
