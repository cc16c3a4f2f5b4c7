#!/bin/bash
# Round-5 session 16: the three scalar-vote changes of session 15 one at a time (touch masks in the sweep / liveness from data / carried warm start)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
bash tools/sessions/ab3.sh r5_s16/ab "prox2=gym-os2r_amd/ab/libos2r_prox2.so touch=gym-os2r_amd/ab/libos2r_va.so live=gym-os2r_amd/ab/libos2r_vb.so carried=gym-os2r_amd/ab/libos2r_vc.so all=gym-os2r_amd/ab/libos2r_votes.so" "--workload C4" "--workload C3"
