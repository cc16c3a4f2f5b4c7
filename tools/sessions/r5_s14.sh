#!/bin/bash
# Round-5 session 14: what moved the clock in session 13 -- two proximal iterations or the lane masks -- and the residual kept from pass 1
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
bash tools/sessions/ab3.sh r5_s14/ab "prox3=gym-os2r_amd/ab/libos2r_base.so masks3=gym-os2r_amd/ab/libos2r_m3.so prox2only=gym-os2r_amd/ab/libos2r_p2.so prox2=gym-os2r_amd/ab/libos2r_prox2.so keepw=gym-os2r_amd/libos2r.so" "--workload C4" "--workload C3"
