#!/bin/bash
# Everything profiles/ records for a round, in one GPU call: rocprofv3 kernel stats of one bench
# step, SQ instruction / activity counters and TCC traffic of report_kernel (separate --pmc
# passes), and the plain bench line with the CPU baseline.  Run from the repo root on the GPU box.
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/round
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-trace > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
echo "stats done"
cd $ROOT
bash scripts/gpu_pmc.sh "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" > $OUT/pmc_summary.txt 2>&1
echo "pmc done"
bash scripts/gpu_traffic.sh > $OUT/traffic_summary.txt 2>&1
echo "traffic done"
timeout -k 10 400 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench done"
tail -c 600 $OUT/bench.json
