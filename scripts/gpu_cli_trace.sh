#!/bin/bash
# rocprofv3 kernel stats + idle-gap analysis of one command-line plan line (config-3-like blocks).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=/dev/shm/damar_clit
rm -rf $W && mkdir -p $W && cd $W
$ROOT/damar_amd/bin/simdb . SIM ${GENOME:-4.6} -c${COV:-87} -r3 -e.15 -S${BLOCK:-25} > nblocks.txt
A=${A:-9}; bs=""; for b in $(seq $A -1 1); do bs="$bs $W/SIM.$b"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/clit -o c --output-format csv -- $ROOT/damar_amd/bin/daligner -k14 -j16 $W/SIM.$A $bs > $ROOT/gpurun_out/clit.log 2>&1
cd $ROOT
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/clit/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]:
        print("  %-50s calls %5s total %8.2f ms avg %7.3f ms" % (r["Name"][:50], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
python3 scripts/gap_analysis.py $(find gpurun_out/clit -name "*kernel_trace.csv" | head -1) | head -12
rm -rf $W
