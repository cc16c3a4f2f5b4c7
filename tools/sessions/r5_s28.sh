#!/bin/bash
# Round-5 session 28: scheduler switches on the final kernels (max-ilp strategy / no post-RA scheduler / no pre-RA machine scheduler)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
bash tools/sessions/ab3.sh r5_s28/ab "head=gym-os2r_amd/ab/libos2r_head.so maxilp=gym-os2r_amd/ab/libos2r_s1.so nopostsched=gym-os2r_amd/ab/libos2r_s2.so nomisched=gym-os2r_amd/ab/libos2r_s3.so" "--workload C4" "--workload C3"
