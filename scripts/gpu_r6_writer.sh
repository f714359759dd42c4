#!/bin/bash
# round 6: GPU parity suite, the contract bench (config-3 leg with its tail/write thread time), then config 4's first 24 blocks
# in the default overlapped mode (lead leg wall and phases).           gpurun --timeout 1200 -- bash scripts/gpu_r6_writer.sh
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r6w_tests.log 2>&1
rc=$?; tail -3 gpurun_out/r6w_tests.log; echo "pytest rc $rc"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python bench.py > gpurun_out/r6w_bench.json 2> gpurun_out/r6w_bench.err
rc=$?; echo "bench rc $rc"; cut -c1-1500 gpurun_out/r6w_bench.json
[ $rc -ne 0 ] && { tail -5 gpurun_out/r6w_bench.err; exit $rc; }
P=$PWD
W=$(mktemp -d /dev/shm/c4w.XXXX)
damar_amd/bin/simdb $W SIM 248 -c80 -m15000 -s3000 -e.15 -r4 -S78 -N24 > /dev/null || exit 1
python3 - "$W" 24 <<'PY'
import sys
w, n = sys.argv[1], int(sys.argv[2])
open(w + "/plan.txt", "w").write("".join("daligner -k14 -j8 SIM.%d %s\n" % (a, " ".join("SIM.%d" % b for b in range(a, 0, -1))) for a in range(1, n + 1)))
open(w + "/keep.txt", "w").write("nothing-is-kept\n")
PY
for i in 1 2; do
  ( cd $W && DAMAR_LAS_KEEP=$W/keep.txt DAMAR_PLAN_TIDY=1 DAMAR_PLAN_STATS=$P/gpurun_out/r6w_c4lead_$i.json timeout -k 10 120 $P/damar_amd/bin/daligner -P plan.txt > $P/gpurun_out/r6w_c4lead_$i.log 2>&1 ) || { echo "lead $i failed"; tail -5 gpurun_out/r6w_c4lead_$i.log; exit 1; }
  cut -c1-900 gpurun_out/r6w_c4lead_$i.json; echo
done
rm -rf $W
