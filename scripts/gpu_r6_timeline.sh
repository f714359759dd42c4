#!/bin/bash
# round 6: kernel traces of a few config-2 steps with the seed sort over all the bits (DAMAR_SORT_PAIR=0) and over the read pair
# only (1): which stream waits for which (scripts analysis: per queue busy time, report-stream idle time).
P=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for ps in 0 1; do
  DAMAR_SORT_PAIR=$ps timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $P/gpurun_out/tl_$ps -o r -- python3 $P/bench.py --steps 3 --warmup 1 --no-cpu --no-e2e --no-legs --no-trace > $P/gpurun_out/tl_$ps.json 2> $P/gpurun_out/tl_$ps.err || { echo "trace $ps failed"; tail -3 $P/gpurun_out/tl_$ps.err; exit 1; }
  python3 - $P/gpurun_out/tl_$ps $ps <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [(r['Queue_Id'], r['Kernel_Name'], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: r[2])
t0, t1 = rows[0][2], max(r[3] for r in rows)
byq = collections.defaultdict(list)
for q, n, s, e in rows: byq[q].append((s, e, n))
print('sort_pair=%s: span %.1f ms' % (sys.argv[2], (t1 - t0) / 1e6))
for q, l in sorted(byq.items()):
    busy = 0; cur_s, cur_e = l[0][0], l[0][1]
    for s, e, n in l[1:]:
        if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    names = collections.Counter(n.split('(')[0][:28] for _, _, n in l).most_common(2)
    print('  queue %s: %5d kernels, busy %.1f ms, first %.1f last %.1f  %s' % (q, len(l), busy / 1e6, (l[0][0] - t0) / 1e6, (l[-1][1] - t0) / 1e6, names))
# time where the report kernel runs and a seed kernel runs too / alone
rep = [(s, e) for q, n, s, e in rows if 'report2' in n]
seed = [(s, e) for q, n, s, e in rows if 'report2' not in n]
def union(iv):
    iv = sorted(iv); out = []
    for s, e in iv:
        if out and s <= out[-1][1]: out[-1][1] = max(out[-1][1], e)
        else: out.append([s, e])
    return out
def inter(a, b):
    i = j = 0; tot = 0
    while i < len(a) and j < len(b):
        s = max(a[i][0], b[j][0]); e = min(a[i][1], b[j][1])
        if e > s: tot += e - s
        if a[i][1] < b[j][1]: i += 1
        else: j += 1
    return tot
ur, us = union(rep), union(seed)
tr, ts, both = sum(e - s for s, e in ur), sum(e - s for s, e in us), inter(ur, us)
print('  report running %.1f ms, seed kernels running %.1f ms, both at once %.1f ms, neither %.1f ms' % (tr / 1e6, ts / 1e6, both / 1e6, ((t1 - t0) - tr - ts + both) / 1e6))
PY
done
