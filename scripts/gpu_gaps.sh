#!/bin/bash
# Idle time of the device between consecutive kernels of a bench step (kernel trace), by the kernel that precedes the gap.
# usage: gpurun -- bash scripts/gpu_gaps.sh [DAMAR_OVERLAP value, default 2]
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/gaps
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export DAMAR_OVERLAP=${1:-2}
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o r -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-trace --no-e2e --no-legs > $OUT/bench.json 2> $OUT/err.txt
cd $ROOT
T=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 scripts/trace_gaps.py $T | tee $OUT/gaps.txt
python3 scripts/trace_streams.py $T | tee $OUT/streams.txt
python3 scripts/trace_phase.py $T | tee $OUT/phase.txt
