#!/bin/bash
# report kernel A/B on one box: the library of the tree against experiment builds (build/exp_*), every kernel alone on the
# machine (DAMAR_OVERLAP=0), kernel trace only.   gpurun -- bash scripts/gpu_ab_report.sh build/exp_noscalar ...
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/abrep
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export DAMAR_OVERLAP=0
for rep in 1 2; do
for dir in damar_amd "$@"; do
  tag=$(basename $dir)_$rep
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --kernel-include-regex "report" --output-format csv -d $OUT/$tag/t -o r -- python3 $ROOT/scripts/exp_la.py $dir > $OUT/$tag.log 2>&1 || { echo "$tag: trace failed"; tail -5 $OUT/$tag.log; exit 1; }
  python3 - "$OUT/$tag" "$tag" <<'PY'
import csv, glob, sys, json
ms = n = 0
for f in glob.glob(sys.argv[1] + "/t/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "report" in r["Name"]:
            ms += float(r["TotalDurationNs"]) / 1e6; n += int(r["Calls"])
line = [l for l in open(sys.argv[1] + ".log") if l.startswith("{")]
par = json.loads(line[-1])["parity"] if line else None
print("%-22s report kernels %8.1f ms in %d launches   parity %s" % (sys.argv[2], ms, n, par and par.get("identical")))
PY
done
done
