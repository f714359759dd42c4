#!/bin/bash
# round 6, final tree: the stress sweeps against the oracle (other seeds than the earlier run of the round), then config 4's
# first 24 blocks kernel by kernel.                              gpurun --timeout 1200 -- bash scripts/gpu_r6_final.sh
mkdir -p gpurun_out
( timeout -k 10 500 python3 scripts/stress_options.py 120 611 | tail -3
  timeout -k 10 300 python3 scripts/stress_dbs.py 12 612 | tail -2
  timeout -k 10 300 python3 scripts/stress_tandem.py 40 613 | tail -2 ) > gpurun_out/r6_stress_final.txt 2>&1
cat gpurun_out/r6_stress_final.txt
grep -q "bad [1-9]" gpurun_out/r6_stress_final.txt && exit 1
bash scripts/gpu_c4_kernels.sh > gpurun_out/r6_c4k_final.txt 2>&1; grep "pair_work\|merge_tiles\|DAMAR_OVERLAP" gpurun_out/r6_c4k_final.txt | cut -c1-200
