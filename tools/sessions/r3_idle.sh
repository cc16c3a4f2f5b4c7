#!/bin/bash
# round 3: where does a wave wait?  Issue counters of the default run against the same kernel with the first four sweeps only
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r3_idle
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for v in default floor; do
  if [ $v = floor ]; then EXTRA="--pgs-tol 1e-3"; else EXTRA=""; fi
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_WAVES --output-format csv -d "$OUT/pmc_$v" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-count --steps 200 --warmup 20 $EXTRA > "$OUT/$v.log" 2>&1
  python3 - "$OUT/pmc_$v" $v <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "step_kernel" in r["Kernel_Name"]]
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})[-200:]
    for r in rows:
        if int(r["Dispatch_Id"]) in ids:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
w = m.get("SQ_WAVES", 1024)
print(sys.argv[2], {k: round(v / w, 1) for k, v in m.items() if k != "SQ_WAVES"})
PY
  rm -rf "$OUT/pmc_$v"
done
