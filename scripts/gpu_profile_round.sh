#!/bin/bash
# Everything profiles/ records for a round, in one GPU call (run from the repo root on the GPU box):
#   1. rocprofv3 --kernel-trace --stats of one bench step                      -> gpurun_out/round/stats
#   2. SQ instruction / activity / LDS counters, separate --pmc passes         -> pmc_summary.txt
#   3. TCC FETCH_SIZE / WRITE_SIZE of the report kernel, separate passes        -> traffic_summary.txt
#   4. the issue-rate calibration (tools/roofcal.hip) under the same counters   -> roofcal*.txt
#   5. the plain bench line (with cpu_baseline, end_to_end)                     -> bench.json
# usage: bash scripts/gpu_profile_round.sh [packed]
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/round
rm -rf $OUT; mkdir -p $OUT
if [ "$1" = packed ]; then export DAMAR_PACKED=1; fi
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-trace --no-e2e --no-legs > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
echo "stats done"
cd $ROOT
bash scripts/gpu_pmc.sh "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVES" \
                        "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" \
                        "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS GRBM_GUI_ACTIVE" > $OUT/pmc_summary.txt 2>&1
echo "pmc done"
# a pass with counters this ROCm may not know (kept apart: a refused name must not cost the passes above)
bash scripts/gpu_pmc.sh "SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS" > $OUT/pmc_extra.txt 2>&1 || echo "extra pmc pass refused"
bash scripts/gpu_traffic.sh > $OUT/traffic_summary.txt 2>&1
echo "traffic done"
if [ -f build/prof/libdamar_hip.so ]; then
  timeout -k 10 300 python3 scripts/prof_la.py > $OUT/prof_counters.txt 2> $OUT/prof.err || echo "prof variant failed"
  echo "prof done"
fi
timeout -k 10 500 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench done"
tail -c 900 $OUT/bench.json
