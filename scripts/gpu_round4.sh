#!/bin/bash
# Round-4 evidence in one GPU call: the round profile (kernel stats under the default overlapped mode, PMC passes, TCC
# traffic of the report kernel, the plain bench line with cpu_baseline / contract / legs), kernel stats with every kernel
# alone on the machine (DAMAR_OVERLAP=0), and the radix sort on its own.
cd "$(dirname "$0")/.."
bash scripts/gpu_profile_round.sh > gpurun_out/round4_profile.log 2>&1; tail -3 gpurun_out/round4_profile.log
bash scripts/gpu_stats_seq.sh > gpurun_out/round4_seqstats.txt 2>&1; head -20 gpurun_out/round4_seqstats.txt
bash scripts/gpu_sortbench.sh > gpurun_out/round4_sortbench.txt 2>&1; tail -25 gpurun_out/round4_sortbench.txt
