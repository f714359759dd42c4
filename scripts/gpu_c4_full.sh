#!/bin/bash
# BASELINE config 4 at full size: the whole 19.8 Gbp database (255 blocks) is generated on the box (2-3 min) and a
# seeded random sample of 8 block pairs from ALL blocks is compared with the reference's md5s
# (tests/golden/config4_ref_md5.txt, sample "full").  Run from the repo root on the GPU box.
set -e
mkdir -p gpurun_out
( while true; do sleep 60; echo "[c4 full] still running $(date +%T)"; done ) &
KEEP=$!
DAMAR_C4_FULL=1 timeout -k 10 900 python -m pytest tests/test_gpu_configs.py -q -k "all_255_blocks" 2>&1 | tee gpurun_out/c4_full.log | tail -5
kill $KEEP
