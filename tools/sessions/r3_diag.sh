#!/bin/bash
# round 3 diagnostics: why the counting replay fails; phase shares of the exact-finish kernel (stamp build)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r3_diag
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 300 python - > "$OUT/count.log" 2>&1 <<'PY'
import bench, torch
class A:
    workload="C4"; envs_per_gpu=65536; dtype="f64"; seed=42; pgs_iters=None; pgs_normal_iters=3; pgs_tol=None; pgs_exact=None; runtime_model=False
cfg,model,spec=bench.build_config(A,0,1)
from gym_os2r_amd.sim import HipSim
sim=HipSim(cfg)
sim.bench_steps(1200)
sim.count_work(True)
try:
    sim.bench_steps(100)
    c=sim.work_counters()
    wi=c["wave_iterations"]
    print({k:v/wi for k,v in c.items()})
except Exception as e:
    print("ERR", e)
PY
cat "$OUT/count.log" | tail -3
timeout -k 10 300 python tools/dbg/stamps.py C4 1200 > "$OUT/stamps_exact.txt" 2>&1; cat "$OUT/stamps_exact.txt" | grep -v "^  *dyn\|amdgpu.ids"
OS2R_PGS_ITERS=20 OS2R_PGS_EXACT=0 timeout -k 10 300 python tools/dbg/stamps.py C4 1200 > "$OUT/stamps_legacy.txt" 2>&1; grep "PGS\|per-wave\|stamp build" "$OUT/stamps_legacy.txt"
