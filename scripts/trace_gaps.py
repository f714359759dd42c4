#!/usr/bin/env python3
"""Device idle time between kernels, from a rocprofv3 kernel trace: the union of the kernels' intervals is taken, every
gap shorter than 20 ms (longer ones are the pauses between bench steps) is charged to the kernel that ended last before it
and to the kernel that starts after it."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-48:]) for r in rows))
busy_end, last = ev[0][1], ev[0][2]
before, after = collections.Counter(), collections.Counter()
nb, gaps, busy, span0 = collections.Counter(), 0, 0, ev[0][0]
cur0 = ev[0][0]
for s, e, n in ev[1:]:
    if s > busy_end:
        g = s - busy_end
        if g < 20e6:
            before[last] += g; after[n] += g; nb[(last, n)] += 1; gaps += g
        busy += busy_end - cur0; cur0 = s
    if e > busy_end:
        busy_end, last = e, n
busy += busy_end - cur0
print('kernels %d, device busy %.1f ms, idle in gaps < 20 ms: %.1f ms' % (len(ev), busy / 1e6, gaps / 1e6))
print('-- idle by the kernel BEFORE the gap')
for k, v in before.most_common(14): print('%8.2f ms  %s' % (v / 1e6, k))
print('-- idle by the kernel AFTER the gap')
for k, v in after.most_common(14): print('%8.2f ms  %s' % (v / 1e6, k))
