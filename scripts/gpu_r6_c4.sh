#!/bin/bash
# round 6: the whole config-4 plan on one GPU (scripts/gpu_c4_whole.py), after a rehearsal on its first 24 blocks in tiles of 8
mkdir -p gpurun_out
DAMAR_PLAN_TILE=8 timeout -k 10 300 python3 scripts/gpu_c4_whole.py 24 > gpurun_out/c4_rehearsal.log 2>&1
rc=$?; tail -3 gpurun_out/c4_rehearsal.log | cut -c1-600; cp gpurun_out/c4_whole.json gpurun_out/c4_rehearsal.json 2>/dev/null
echo "rehearsal rc $rc"
[ $rc -ne 0 ] && { tail -20 gpurun_out/c4_whole.err; exit $rc; }
timeout -k 10 900 python3 scripts/gpu_c4_whole.py 2>&1 | tee gpurun_out/c4_whole.log | cut -c1-400
rc=${PIPESTATUS[0]}
echo "whole rc $rc"; tail -5 gpurun_out/c4_whole.err
exit $rc
