#!/bin/bash
# round 6: the seed stage's waits for a count from the device, polling (default) against sleeping (DAMAR_SYNC_SPIN=0): config 4's
# first 24 blocks and the contract bench, each both ways.           gpurun --timeout 1200 -- bash scripts/gpu_r6_spin.sh
mkdir -p gpurun_out
P=$PWD
W=$(mktemp -d /dev/shm/c4w.XXXX)
damar_amd/bin/simdb $W SIM 248 -c80 -m15000 -s3000 -e.15 -r4 -S78 -N24 > /dev/null || exit 1
python3 - "$W" 24 <<'PY'
import sys
w, n = sys.argv[1], int(sys.argv[2])
open(w + "/plan.txt", "w").write("".join("daligner -k14 -j8 SIM.%d %s\n" % (a, " ".join("SIM.%d" % b for b in range(a, 0, -1))) for a in range(1, n + 1)))
open(w + "/keep.txt", "w").write("nothing-is-kept\n")
PY
for i in 1 2 3; do
  for spin in 1 0; do
    ( cd $W && DAMAR_SYNC_SPIN=$spin DAMAR_LAS_KEEP=$W/keep.txt DAMAR_PLAN_TIDY=1 DAMAR_PLAN_STATS=$P/gpurun_out/r6s_c4lead_${spin}_$i.json timeout -k 10 120 $P/damar_amd/bin/daligner -P plan.txt > $P/gpurun_out/r6s_c4lead_${spin}_$i.log 2>&1 ) || { echo "lead $spin $i failed"; tail -5 gpurun_out/r6s_c4lead_${spin}_$i.log; exit 1; }
    python3 -c "
import json,sys
d=json.load(open('gpurun_out/r6s_c4lead_${spin}_$i.json'))
print('c4 lead spin=$spin run $i: wall %.1f ms work_items %d records %d phases %s' % (d['wall_ms'], d['work_items'], d['records'], d['phase_ms']))"
  done
done
rm -rf $W
for spin in 1 0; do
  DAMAR_SYNC_SPIN=$spin timeout -k 10 400 python bench.py > gpurun_out/r6s_bench_$spin.json 2> gpurun_out/r6s_bench_$spin.err
  rc=$?; echo "bench spin=$spin rc $rc"
  [ $rc -ne 0 ] && { tail -5 gpurun_out/r6s_bench_$spin.err; exit $rc; }
  python3 -c "
import json
d=json.load(open('gpurun_out/r6s_bench_$spin.json'))
print('ms_per_step %.1f synced %.1f contract %s' % (d['ms_per_step'], d['ms_per_step_synced'], {k: round(v, 3) for k, v in d['contract'].items() if k.endswith('_s')}))
for k in ('config3','config4_lead','config5_datander'):
    l=d['legs'][k]; print(k, 'wall_s %.3f' % l['wall_s'], l.get('phase_ms'))"
done
