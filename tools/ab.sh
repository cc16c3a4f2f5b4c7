#!/bin/bash
# Same-box A/B of the working tree against the installed library (boxes of the pool differ by +-4 %):
#   tools/ab.sh            builds the working tree into gym-os2r_amd/libos2r_ab.so and prints the gpurun command
# Run the printed command; A = gym-os2r_amd/libos2r.so as built earlier, B = the working tree.
set -e
cd "$(dirname "$0")/../gym-os2r_amd/csrc"
make -j8 BUILD=build_ab OUT=../libos2r_ab.so ../libos2r_ab.so 2>&1 | grep -E "error|warning" || true
ls -la ../libos2r_ab.so
cat <<'CMD'
/usr/local/graft/bin/gpurun --timeout 900 -- 'for r in 1 2 3; do for v in A B; do if [ $v = B ]; then export OS2R_LIBRARY=$GRAFT_REPO_ROOT/gym-os2r_amd/libos2r_ab.so; else unset OS2R_LIBRARY; fi; timeout -k 10 300 python bench.py --no-cpu-baseline $BENCH_ARGS > gpurun_out/bench_ab.json 2>/dev/null; python -c "import json;d=json.load(open(\"gpurun_out/bench_ab.json\"));print(\"$v\", round(d[\"value\"]/1e6,1), round(d[\"ms_per_step\"]*1e3,2))"; done; done'
CMD
