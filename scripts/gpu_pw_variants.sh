#!/bin/bash
# prices the phases of pair_work_mark: config 4's first 8 blocks (36 block pairs), every kernel alone, one library per variant
# (build/exp_pw_*: results of the cut-down variants are wrong on purpose).      -> gpurun_out/pw_variants.txt
P=$PWD
mkdir -p gpurun_out
W=$(mktemp -d /dev/shm/pwv.XXXX)
damar_amd/bin/simdb $W SIM 248 -c80 -m15000 -s3000 -e.15 -r4 -S78 -N8 > /dev/null || exit 1
python3 - "$W" 8 <<'PY'
import sys
w, n = sys.argv[1], int(sys.argv[2])
open(w + "/plan.txt", "w").write("".join("daligner -k14 -j8 SIM.%d %s\n" % (a, " ".join("SIM.%d" % b for b in range(a, 0, -1))) for a in range(1, n + 1)))
open(w + "/keep.txt", "w").write("nothing-is-kept\n")
PY
: > gpurun_out/pw_variants.txt
for dir in "$@"; do
  tag=$(basename $dir)
  ( cd /tmp && export TMPDIR=/tmp && cd $W && LD_LIBRARY_PATH=$P/$dir:$LD_LIBRARY_PATH DAMAR_OVERLAP=0 DAMAR_LAS_KEEP=$W/keep.txt DAMAR_PLAN_TIDY=1 \
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $P/gpurun_out/pwv_$tag -o r -- $P/damar_amd/bin/daligner -P plan.txt > $P/gpurun_out/pwv_$tag.log 2>&1 ) || { echo "$tag failed"; tail -5 gpurun_out/pwv_$tag.log; exit 1; }
  python3 - $tag <<'PY' | tee -a gpurun_out/pw_variants.txt
import csv, glob, sys
for f in glob.glob('gpurun_out/pwv_%s/**/*kernel_stats.csv' % sys.argv[1], recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r['Name'] for k in ('pair_work', 'pair_heads_mark', 'pair_screen', 'merge_fast', 'merge_emit')):
            print('%-12s %-40s calls %5s avg %8.1f us' % (sys.argv[1], r['Name'][:40], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
rm -rf $W
