#!/bin/bash
# config 4's first 24 blocks (300 block pairs) kernel by kernel: one `daligner -P` command in-process under rocprofv3,
# every kernel alone on the machine (DAMAR_OVERLAP=0) and in the default overlapped mode.   -> gpurun_out/c4k_*.txt
P=$PWD
mkdir -p gpurun_out
W=$(mktemp -d /dev/shm/c4k.XXXX)
damar_amd/bin/simdb $W SIM 248 -c80 -m15000 -s3000 -e.15 -r4 -S78 -N${1:-24} > /dev/null || exit 1
python3 - "$W" ${1:-24} <<'PY'
import sys
w, n = sys.argv[1], int(sys.argv[2])
open(w + "/plan.txt", "w").write("".join("daligner -k14 -j8 SIM.%d %s\n" % (a, " ".join("SIM.%d" % b for b in range(a, 0, -1))) for a in range(1, n + 1)))
open(w + "/keep.txt", "w").write("nothing-is-kept\n")
PY
for mode in 0 1; do
  ( cd /tmp && export TMPDIR=/tmp && cd $W && DAMAR_OVERLAP=$mode DAMAR_LAS_KEEP=$W/keep.txt DAMAR_PLAN_TIDY=1 DAMAR_PLAN_STATS=$P/gpurun_out/c4k_stats_$mode.json \
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/gpurun_out/c4k_$mode -o r -- $P/damar_amd/bin/daligner -P plan.txt > $P/gpurun_out/c4k_$mode.log 2>&1 ) || { echo "mode $mode failed"; tail -5 gpurun_out/c4k_$mode.log; }
  python3 - $mode <<'PY'
import csv, glob, sys, json
m = sys.argv[1]
out = open('gpurun_out/c4k_kernels_%s.txt' % m, 'w')
for f in glob.glob('gpurun_out/c4k_%s/**/*kernel_stats.csv' % m, recursive=True):
    for r in list(csv.DictReader(open(f)))[:24]:
        ln = '%-64s calls %6s total %9.2f ms avg %9.1f us  %5s%%' % (r['Name'][:64], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, r['Percentage'])
        print(ln); out.write(ln + '\n')
try:
    st = json.load(open('gpurun_out/c4k_stats_%s.json' % m))
    ln = 'DAMAR_OVERLAP=%s wall %.0f ms phases %s' % (m, st['wall_ms'], st['phase_ms'])
    print(ln); out.write(ln + '\n')
except Exception as e:
    print('no stats', e)
PY
done
rm -rf $W
