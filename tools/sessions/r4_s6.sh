#!/bin/bash
# round 4, session 6: PMC counters per wave and env-step of the production kernels, C3 against C4 (why is C3's launch 194 us
# when its stamp build's is 168?)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
tools/sessions/pmc_ab.sh r4_s6 "C3=gym-os2r_amd/libos2r.so" --workload C3
tools/sessions/pmc_ab.sh r4_s6 "C4=gym-os2r_amd/libos2r.so" --workload C4
