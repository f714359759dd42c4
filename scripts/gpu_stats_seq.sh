#!/bin/bash
# kernel stats of one bench step with the report launch NOT overlapped (every kernel alone on the machine)
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/seqstats
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export DAMAR_OVERLAP=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-trace --no-e2e --no-legs "$@" > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
cd $ROOT
python3 - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/seqstats/stats/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:28]:
        print('%-60s calls %5s total %9.3f ms avg %9.1f us  %5s%%' % (r['Name'][:60], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, r['Percentage']))
PY
