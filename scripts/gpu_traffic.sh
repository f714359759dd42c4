#!/bin/bash
# HBM traffic of report_kernel for one bench step (TCC counters, report_kernel only).
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/traffic
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout -k 10 400 rocprofv3 --pmc $set --kernel-trace --kernel-include-regex "report" -d $OUT/$set -o pmc --output-format csv -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-trace --no-e2e --no-legs > $OUT/$set.log 2>&1
  python3 - "$OUT/$set" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])] += float(r["Counter_Value"]); n[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])] += 1
for k, v in acc.items(): print(k, "%.6g" % v, n[k], "launches")
PY
done
