#!/bin/bash
# SQ / TCC counters of the trace-expansion kernel on one config-2 block pair (separate --pmc passes).
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/trace_pmc
mkdir -p $OUT
W=/dev/shm/damar_trp
rm -rf $W && mkdir -p $W && cd $W
$ROOT/damar_amd/bin/simdb . SIM 27 -c20 -r2 -e.15 -S135 > /dev/null
timeout -k 10 300 $ROOT/damar_amd/bin/daligner -k14 -j16 SIM.1 SIM.1 > dal.log 2>&1
LAS=$W/$(ls d001_*/SIM.1.SIM.1.las | head -1)
cd /tmp && export TMPDIR=/tmp
if [ $# -eq 0 ]; then set -- "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; fi
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/stats -o tr --output-format csv -- $ROOT/damar_amd/bin/lastrace -v $W/SIM.1 $W/SIM.1 $LAS $W/o.bin > $OUT/stats.log 2>&1
python3 - $OUT/stats <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]:
        print("  %-60s calls %s avg %.3f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
for set in "$@"; do
  tag=$(echo $set | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace -d $OUT/$tag -o pmc --output-format csv -- $ROOT/damar_amd/bin/lastrace $W/SIM.1 $W/SIM.1 $LAS $W/o.bin > $OUT/$tag.log 2>&1
  echo "== $set"
  python3 - "$OUT/$tag" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][:40]][r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(acc, key=lambda k: -sum(acc[k].values()))[:3]:
    print("  %-40s" % k, {c: "%.4g" % v for c, v in acc[k].items()})
PY
done
rm -rf $W
