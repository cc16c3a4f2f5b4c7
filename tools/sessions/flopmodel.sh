#!/bin/bash
# Calibrates profiles/flop_model.json for the workloads that have a counting kernel variant (C4, C3, V1):
# per-launch work counters and rocprofv3 PMC flop counts of the same 1200 launches from the reset, then the fit.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/flopmodel
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
for w in ${FLOP_WORKLOADS:-C4 C3 V1}; do
  cd "$ROOT"
  timeout -k 10 300 python tools/flop_model.py counts --workload $w --steps 1200 --out "$OUT/counts_$w.json"; echo "counts $w rc=$?"
  cd /tmp
  timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 --output-format csv -d "$OUT/pmc_$w" -- python3 "$ROOT/tools/flop_model.py" run --workload $w --steps 1200 > "$OUT/pmc_$w.log" 2>&1; echo "pmc $w rc=$?"
done
cd "$ROOT"; python tools/flop_model.py fit "$OUT" "$OUT/flop_model" | tail -40
rm -rf "$OUT"/pmc_C4 "$OUT"/pmc_C3 "$OUT"/pmc_V1
