#!/bin/bash
# Prices parts of the report kernel with the experiment builds (scripts/build_exp.sh): for each build/exp_* directory
# one bench step with DAMAR_OVERLAP=0, three passes: kernel trace (time), FETCH_SIZE, WRITE_SIZE -- report kernel only.
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/exp
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export DAMAR_OVERLAP=0
for dir in "$@"; do
  tag=$(basename $dir)
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --kernel-include-regex "report" --output-format csv -d $OUT/$tag/t -o r -- python3 $ROOT/scripts/exp_la.py $dir > $OUT/$tag.t.log 2>&1 || { echo "$tag: trace failed"; exit 1; }
  for set in FETCH_SIZE WRITE_SIZE "TCC_EA0_ATOMIC_sum TCC_WRITE_sum TCC_READ_sum"; do
    st=$(echo $set | tr ' ' '_')
    timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --kernel-include-regex "report" --output-format csv -d $OUT/$tag/$st -o pmc -- python3 $ROOT/scripts/exp_la.py $dir > $OUT/$tag.$st.log 2>&1 || { echo "$tag: $set failed"; exit 1; }
  done
  python3 - "$OUT/$tag" "$tag" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]] += float(r["Counter_Value"])
ms = 0.0
for f in glob.glob(sys.argv[1] + "/t/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "report" in r["Name"]:
            ms += float(r["TotalDurationNs"]) / 1e6
print("%-16s report kernels %8.1f ms  " % (sys.argv[2], ms) + "  ".join("%s %.4g" % kv for kv in sorted(acc.items())))
PY
done
