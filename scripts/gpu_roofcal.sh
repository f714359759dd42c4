#!/bin/bash
# Issue-rate calibration: pure VALU / SALU / mixed / bpermute streams at 1-8 waves per
# SIMD, plain and under the SQ counters report_kernel is read with.  Run from the repo root on the GPU box.
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/roof
mkdir -p $OUT
$ROOT/damar_amd/bin/roofcal 20000 > $OUT/roofcal.txt
cat $OUT/roofcal.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1 || true
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"; do
  tag=cal_$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace -d $OUT/$tag -o pmc --output-format csv -- $ROOT/damar_amd/bin/roofcal 5000 > $OUT/$tag.log 2>&1
  python3 - "$OUT/$tag" <<'PY' | tee $OUT/roofcal_pmc.txt
import csv, glob, sys, collections
rows = collections.defaultdict(dict)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows[(int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0], int(r["Grid_Size"]))][r["Counter_Name"]] = float(r["Counter_Value"])
for (d, k, g), c in sorted(rows.items()):
    print(d, k, "grid", g, " ".join("%s=%.4g" % kv for kv in sorted(c.items())))
PY
done
