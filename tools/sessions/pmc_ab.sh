#!/bin/bash
# same-box PMC comparison of builds of the same ABI on one bench workload (instruction counts and waits per wave and env-step):
#   tools/sessions/pmc_ab.sh OUTDIR "LABEL=lib.so ..." [bench args]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/$1; shift
LIBS=$1; shift
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for lv in $LIBS; do
  label=${lv%%=*}; lib=${lv#*=}
  export OS2R_LIBRARY=$ROOT/$lib
  for grp in "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_WAVES" \
             "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_WAVES" \
             "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_WAVES" \
             "SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES"; do
    rm -rf "$OUT/pmc_tmp"
    timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc_tmp" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-count --no-gym-level --steps 200 --warmup 20 "$@" > "$OUT/$label.log" 2>&1
    python3 - "$OUT/pmc_tmp" "$label" <<'PY' | tee -a "$OUT/pmc_table.txt"
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "step_kernel" in r["Kernel_Name"]]
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})[-200:]
    keep = set(ids)
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in rows:
        if int(r["Dispatch_Id"]) in keep:
            per[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    for k, v in per.items():
        acc[k] += list(v.values())
m = {k: sum(v) / len(v) for k, v in acc.items()}
w = m.get("SQ_WAVES", 1024) or 1024
print(sys.argv[2], {k: round(v / w, 1) for k, v in sorted(m.items()) if k != "SQ_WAVES"})
PY
    rm -rf "$OUT/pmc_tmp"
  done
done
