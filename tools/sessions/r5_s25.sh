#!/bin/bash
# Round-5 session 25: IR-level code-generation switches (SimplifyCFG sink / hoist of common code, if-conversion threshold)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
bash tools/sessions/ab3.sh r5_s25/ab "head=gym-os2r_amd/ab/libos2r_head.so nosinkcommon=gym-os2r_amd/ab/libos2r_w1.so nohoistcommon=gym-os2r_amd/ab/libos2r_w2.so nofold=gym-os2r_amd/ab/libos2r_w4.so" "--workload C4" "--workload C3"
