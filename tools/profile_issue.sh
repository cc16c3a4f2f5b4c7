#!/bin/bash
# Runs on the GPU box (via gpurun): extra rocprofv3 PMC passes that break down instruction issue
# (instruction cache, scalar / LDS / branch issue, measured fp64 operation mix, LDS conflicts).
# Usage: tools/profile_issue.sh <tag> [bench args...]      outputs under gpurun_out/prof_<tag>/
set -u
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
BENCH="python3 $ROOT/bench.py --no-cpu-baseline $*"
timeout -k 10 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_BRANCH --output-format csv -d "$OUT/pmc_icache" -- $BENCH > "$OUT/pmc_icache.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS --output-format csv -d "$OUT/pmc_issue" -- $BENCH > "$OUT/pmc_issue.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT --output-format csv -d "$OUT/pmc_mix" -- $BENCH > "$OUT/pmc_mix.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE --output-format csv -d "$OUT/pmc_lds" -- $BENCH > "$OUT/pmc_lds.log" 2>&1
find "$OUT" -name "*counter_collection.csv" | head
