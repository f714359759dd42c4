#!/usr/bin/env python3
"""Two-stream timeline of the LAST bench step of a rocprofv3 kernel trace: the report launches (start, length, pause before
the next one) and what the seed stream did meanwhile (busy time inside / outside report launches)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-40:], r['Queue_Id']) for r in rows)
rep = [(s, e) for s, e, n, q in ev if 'report2_kernel' in n or n.endswith('report_kernel')]
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rep = rep[-nl:]
t0, t1 = rep[0][0], rep[-1][1]
seed = [(s, e) for s, e, n, q in ev if not ('report' in n) and e > t0 and s < t1]
def union(iv):
    out = []
    for s, e in sorted(iv):
        if out and s <= out[-1][1]: out[-1][1] = max(out[-1][1], e)
        else: out.append([s, e])
    return out
su = union(seed)
def overlap(a0, a1, iv): return sum(max(0, min(a1, e) - max(a0, s)) for s, e in iv)
print('window %.1f ms: report busy %.1f ms, seed stream busy %.1f ms' % ((t1 - t0) / 1e6, sum(e - s for s, e in rep) / 1e6, overlap(t0, t1, su) / 1e6))
for i, (s, e) in enumerate(rep):
    nxt = rep[i + 1][0] if i + 1 < len(rep) else None
    print('launch %2d  at %7.1f ms  runs %6.1f ms  seed stream busy %5.1f ms inside it   pause to next %s' %
          (i, (s - t0) / 1e6, (e - s) / 1e6, overlap(s, e, su) / 1e6, ('%.2f ms, seed busy %.2f' % ((nxt - e) / 1e6, overlap(e, nxt, su) / 1e6)) if nxt else '-'))
