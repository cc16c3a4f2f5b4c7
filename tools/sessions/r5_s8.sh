#!/bin/bash
# Round-5 session 8: same-box A/B -- round-4 structure / + dual solve (four copies of the finish) / two copies of the finish without and with the dual solve
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
bash tools/sessions/ab3.sh r5_s8/ab "base=gym-os2r_amd/ab/libos2r_base.so small=gym-os2r_amd/ab/libos2r_small.so fin2_nosmall=gym-os2r_amd/ab/libos2r_fin2_nosmall.so fin2=gym-os2r_amd/ab/libos2r_fin2.so" "--workload C4" "--workload C3" "--workload V1"
timeout -k 10 300 python bench.py --no-cpu-baseline --no-gym-level --steps 200 > gpurun_out/r5_s8/bench.json 2> gpurun_out/r5_s8/bench.err; python - <<PY
import json
d=json.load(open("gpurun_out/r5_s8/bench.json")); a=d.get("roofline_valu",{}).get("activity",{})
print("bench", round(d["value"]/1e6,1), "M/s kernel us", round(d["roofline"]["kernel_ms_per_launch"]*1e3,2), {k: round(v,3) for k,v in a.items()})
PY
