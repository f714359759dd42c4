#!/bin/bash
# The HIP calls the host makes while the device idles before a report launch / before a comparison's first kernel.
# usage: gpurun -- bash scripts/gpu_gaps_api.sh [DAMAR_OVERLAP value, default 2]
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/gaps_api
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export DAMAR_OVERLAP=${1:-2}
timeout -k 10 300 rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $OUT/trace -o r -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-trace --no-e2e --no-legs > $OUT/bench.json 2> $OUT/err.txt
cd $ROOT
ls $OUT/trace
python3 scripts/trace_gaps_api.py $OUT/trace | tee $OUT/gaps_api.txt
