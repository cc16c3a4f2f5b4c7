#!/bin/bash
# Compile one instantiation unit and print the resource usage of its step kernels.
# usage: tools/kstat.sh [double|float] [UNIT] [extra hipcc flags...]   UNIT: 0-3 static model, 12-15 run-time nq
REAL=${1:-double}; NQ=${2:-0}; shift; shift
cd "$(dirname "$0")/../gym-os2r_amd/csrc"
mkdir -p /tmp/kstat
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DOS2R_REAL=$REAL -DOS2R_UNIT=$NQ -mllvm -disable-machine-licm -mllvm -disable-machine-sink "$@" \
  -Rpass-analysis=kernel-resource-usage -c os2r_inst.hip -o /tmp/kstat/inst.o 2>&1 \
  | grep -E "Function Name|VGPRs:|AGPRs:|ScratchSize|VGPRs Spill|SGPRs Spill|Occupancy" \
  | sed -e 's/.*remark: [^ ]* *//' -e 's/\[-Rpass.*//' | paste - - - - - - - | sed -e 's/Function Name: _ZN4os2r//' | grep step_kernel | awk '{print}' | cut -c1-260
