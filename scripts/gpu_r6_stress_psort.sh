#!/bin/bash
# round 6: the stress sweeps against the oracle with the seed sort over the read pair only FORCED on (DAMAR_SORT_PAIR=1: on small
# databases the rule of shim.hip match_front would leave it off).     gpurun --timeout 1200 -- bash scripts/gpu_r6_stress_psort.sh
mkdir -p gpurun_out
export DAMAR_SORT_PAIR=1
( timeout -k 10 500 python3 scripts/stress_options.py 120 621 | tail -3
  timeout -k 10 300 python3 scripts/stress_dbs.py 12 622 | tail -2
  DAMAR_TEST_RUN_MAX=5 timeout -k 10 300 python3 scripts/stress_dbs.py 8 623 | tail -2 ) > gpurun_out/r6_stress_psort.txt 2>&1
cat gpurun_out/r6_stress_psort.txt
grep -q "bad [1-9]" gpurun_out/r6_stress_psort.txt && exit 1
exit 0
