#!/bin/bash
# Radix sort on its own (tools/sortbench.hip): correctness sweep of the default variant, timings of the variants
# built by scripts/build_sortbench.sh, kernel stats and LDS bank-conflict counters of the default variant.
# Run on the GPU box: gpurun -- bash scripts/gpu_sortbench.sh
set -e
cd "$(dirname "$0")/.."
OUT=gpurun_out/sortbench
mkdir -p $OUT
B=damar_amd/bin
timeout -k 10 300 $B/sortbench check > $OUT/check.txt 2>&1 || { cat $OUT/check.txt; exit 1; }
cat $OUT/check.txt
for v in $B/sortbench $B/sortbench_*; do
  [ -x "$v" ] || continue
  echo "== $v" | tee -a $OUT/time.txt
  timeout -k 10 200 $v time 5 2>&1 | tee -a $OUT/time.txt
done
export TMPDIR=/tmp
P=$PWD
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/$OUT/stats -- $P/$B/sortbench time 3 > $P/$OUT/stats.log 2>&1 || true
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $P/$OUT/pmc -- $P/$B/sortbench time 1 > $P/$OUT/pmc.log 2>&1 || true
cd $P
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob('gpurun_out/sortbench/stats/**/*kernel_stats.csv', recursive=True):
    print(open(f).read()[:3000])
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob('gpurun_out/sortbench/pmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'][:60]][r['Counter_Name']] += float(r['Counter_Value'])
for k, v in agg.items():
    bc, ia = v.get('SQ_LDS_BANK_CONFLICT', 0), v.get('SQ_LDS_IDX_ACTIVE', 1)
    print(k, dict(v), 'conflict/active = %.3f' % (bc / max(ia, 1)))
PY
