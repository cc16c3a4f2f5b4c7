#!/bin/bash
# Round-5 session 20: on the rotated loop -- its votes read off the data (moved, sweeps) / the free-row mask skipping bodies that no lane at work touches
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
bash tools/sessions/ab3.sh r5_s20/ab "head=gym-os2r_amd/ab/libos2r_head.so datavote=gym-os2r_amd/ab/libos2r_vb.so uskip=gym-os2r_amd/ab/libos2r_vc.so" "--workload C4" "--workload C3" "--workload V1"
