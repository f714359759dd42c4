#!/bin/bash
# SQ instruction counters and HBM traffic of one bench step (separate --pmc passes, kernel-trace only).
# usage (on the GPU box, from the repo root): bash scripts/gpu_pmc.sh
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ $# -eq 0 ]; then set -- "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS"; fi
for set in "$@"; do
  tag=$(echo $set | tr ' ' '_')
  timeout -k 10 280 rocprofv3 --pmc $set --kernel-trace -d $OUT/$tag -o pmc --output-format csv -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-trace --no-e2e --no-legs > $OUT/$tag.log 2>&1
  echo "== $set"
  python3 - "$OUT/$tag" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
for k in sorted(acc, key=lambda k: -sum(acc[k].values()))[:9]:
    print("  %-40s" % k, {c: "%.4g (%d launches)" % (v, n[(k, c)]) for c, v in acc[k].items()})
PY
done
