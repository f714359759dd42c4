#!/bin/bash
# HIP API + kernel timeline of the cold contract command (daligner -P, config 2): which runtime calls the host threads
# spend the first 150 ms in.   gpurun -- bash scripts/gpu_hiptrace.sh
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/hiptrace
rm -rf $OUT; mkdir -p $OUT
python3 - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import bench
from damar_amd import api
work = "/dev/shm/ht_db"
os.makedirs(work, exist_ok=True)
cfg = bench.CONFIGS[2]
nb = api.sim_write_db(work, "SIM", cfg["genome"], coverage=cfg["coverage"], seed=cfg["seed"], block_mbp=cfg["block"])
open(os.path.join(work, "plan.txt"), "w").write(bench.plan_text("SIM", nb))
PY
cd /dev/shm/ht_db && export TMPDIR=/tmp
$ROOT/damar_amd/bin/daligner -P plan.txt > /dev/null 2>&1; sleep 1; rm -rf d0*      # warm the page cache
DAMAR_CLIPROF=1 DAMAR_INITPROF=1 timeout -k 10 300 rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --output-format csv -d $OUT/t -o r -- $ROOT/damar_amd/bin/daligner -P plan.txt > $OUT/run.log 2>&1
cd $ROOT
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/hiptrace/t/**/*hip_api_trace.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
t0 = min(int(r['Start_Timestamp']) for r in rows)
per = collections.defaultdict(lambda: [0, 0.0])
early = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    s = (int(r['Start_Timestamp']) - t0) / 1e6
    per[r['Function']][0] += 1; per[r['Function']][1] += d
    if s < 120:
        early[(r['Function'], r['Thread_Id'])][0] += 1; early[(r['Function'], r['Thread_Id'])][1] += d
print("whole command, by total ms:")
for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])[:14]:
    print("  %-40s calls %6d  %9.1f ms" % (k, v[0], v[1]))
print("first 120 ms after the first HIP call, by (function, thread):")
for k, v in sorted(early.items(), key=lambda kv: -kv[1][1])[:24]:
    print("  %-36s thr %-8s calls %5d  %8.1f ms" % (k[0], k[1], v[0], v[1]))
print("calls longer than 3 ms in the first 150 ms:")
for r in rows:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    s = (int(r['Start_Timestamp']) - t0) / 1e6
    if s < 150 and d > 3:
        print("  +%7.1f ms  %-34s thr %-8s %7.1f ms" % (s, r['Function'], r['Thread_Id'], d))
PY
grep -E "^cli|^init" $OUT/run.log
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/hiptrace/t/**/*memory_copy_trace.csv', recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    api = list(csv.DictReader(open(glob.glob('gpurun_out/hiptrace/t/**/*hip_api_trace.csv', recursive=True)[0])))
    t0 = min(int(r['Start_Timestamp']) for r in api)
    print("memory copies longer than 1 ms or larger than 4 MB:", rows[0].keys())
    for r in rows:
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
        s = (int(r['Start_Timestamp']) - t0) / 1e6
        if d > 1:
            print("  +%7.1f ms  %-28s %7.2f ms" % (s, r.get('Direction', r.get('Name', '?')), d))
PY
