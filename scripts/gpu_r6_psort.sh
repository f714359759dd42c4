#!/bin/bash
# round 6: the seed sort over the read pair only (DAMAR_SORT_PAIR=1, kernels/seed_merge.hip order_runs) against the sort over all the key bits: its tests,
# config 4 first 24 blocks and the contract bench, each both ways.           gpurun --timeout 1200 -- bash scripts/gpu_r6_psort.sh
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "seed_sort_over or work_list or sparse" > gpurun_out/r6p_tests.log 2>&1
rc=$?; tail -3 gpurun_out/r6p_tests.log; echo "pytest rc $rc"
[ $rc -ne 0 ] && { tail -40 gpurun_out/r6p_tests.log; exit $rc; }
P=$PWD
W=$(mktemp -d /dev/shm/c4w.XXXX)
damar_amd/bin/simdb $W SIM 248 -c80 -m15000 -s3000 -e.15 -r4 -S78 -N24 > /dev/null || exit 1
python3 - "$W" 24 <<'PY'
import sys
w, n = sys.argv[1], int(sys.argv[2])
open(w + "/plan.txt", "w").write("".join("daligner -k14 -j8 SIM.%d %s\n" % (a, " ".join("SIM.%d" % b for b in range(a, 0, -1))) for a in range(1, n + 1)))
open(w + "/keep.txt", "w").write("nothing-is-kept\n")
PY
for i in 1 2; do
  for two in 1 0; do
    ( cd $W && DAMAR_SORT_PAIR=$two DAMAR_LAS_KEEP=$W/keep.txt DAMAR_PLAN_TIDY=1 DAMAR_PLAN_STATS=$P/gpurun_out/r6p_c4lead_${two}_$i.json timeout -k 10 120 $P/damar_amd/bin/daligner -P plan.txt > $P/gpurun_out/r6p_c4lead_${two}_$i.log 2>&1 ) || { echo "lead $two $i failed"; tail -5 gpurun_out/r6p_c4lead_${two}_$i.log; exit 1; }
    python3 -c "
import json,sys
d=json.load(open('gpurun_out/r6p_c4lead_${two}_$i.json'))
print('c4 lead psort=$two run $i: wall %.1f ms work_items %d records %d phases %s' % (d['wall_ms'], d['work_items'], d['records'], d['phase_ms']))"
  done
done
rm -rf $W
for two in 1 0; do
  DAMAR_SORT_PAIR=$two timeout -k 10 400 python bench.py > gpurun_out/r6p_bench_$two.json 2> gpurun_out/r6p_bench_$two.err
  rc=$?; echo "bench psort=$two rc $rc"
  [ $rc -ne 0 ] && { tail -5 gpurun_out/r6p_bench_$two.err; exit $rc; }
  python3 -c "
import json
d=json.load(open('gpurun_out/r6p_bench_$two.json'))
print('ms_per_step %.1f synced %.1f' % (d['ms_per_step'], d['ms_per_step_synced']))
for k in ('config3','config4_lead'):
    l=d['legs'][k]; print(k, 'wall_s %.3f' % l['wall_s'], l['phase_ms'])"
done
