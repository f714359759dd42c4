#!/bin/bash
# Round-6 evidence in one GPU call: the round profile (kernel stats under the default overlapped mode, PMC passes, TCC
# traffic of the report kernel, the plain bench line with cpu_baseline / contract / legs), kernel stats with every kernel
# alone on the machine (DAMAR_OVERLAP=0), and the radix sort on its own.
cd "$(dirname "$0")/.."
bash scripts/gpu_profile_round.sh > gpurun_out/round6_profile.log 2>&1; tail -3 gpurun_out/round6_profile.log
bash scripts/gpu_stats_seq.sh > gpurun_out/round6_seqstats.txt 2>&1; head -20 gpurun_out/round6_seqstats.txt
bash scripts/gpu_sortbench.sh > gpurun_out/round6_sortbench.txt 2>&1; tail -25 gpurun_out/round6_sortbench.txt
# datander (config 5) kernel by kernel: one command over the four blocks, in-process under the profiler
W=$(mktemp -d /dev/shm/dtan.XXXX)
damar_amd/bin/simdb $W SIM 27 -c20 -r2 -e.15 -S135 -T.3 > /dev/null
P=$PWD
( cd /tmp && export TMPDIR=/tmp && cd $W && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $P/gpurun_out/tandem_stats -o r -- $P/damar_amd/bin/datander -j16 SIM.1 SIM.2 SIM.3 SIM.4 > $P/gpurun_out/tandem_stats.log 2>&1 ) || echo "tandem profile failed"
rm -rf $W
python3 - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/tandem_stats/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:14]:
        print('%-60s calls %5s total %9.3f ms avg %9.1f us  %5s%%' % (r['Name'][:60], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, r['Percentage']))
PY
