#!/usr/bin/env python3
"""For the LAST bench step of a rocprofv3 kernel + HIP runtime trace: every device-idle gap longer than 100 us, with the
HIP calls the host made inside it (name, duration)."""
import csv, sys, glob, collections
d = sys.argv[1]
kt = list(csv.DictReader(open(glob.glob(d + '/*kernel_trace.csv')[0])))
at = list(csv.DictReader(open(glob.glob(d + '/*hip_api_trace.csv')[0])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-40:]) for r in kt)
api = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Function']) for r in at)
t_end = ev[-1][1]
gaps = []
busy_end, last = ev[0][1], ev[0][2]
for s, e, n in ev[1:]:
    if s > busy_end and s - busy_end < 20e6 and s > t_end - 430e6:
        gaps.append((busy_end, s, last, n))
    if e > busy_end:
        busy_end, last = e, n
tot = collections.Counter(); cnt = collections.Counter()
for g0, g1, a, b in gaps:
    tot[(a, b)] += g1 - g0; cnt[(a, b)] += 1
print('last step: %d gaps, %.1f ms idle' % (len(gaps), sum(tot.values()) / 1e6))
for k, v in tot.most_common(12):
    print('%7.2f ms in %3d gaps   %s  ->  %s' % (v / 1e6, cnt[k], k[0], k[1]))
shown = collections.Counter()
for g0, g1, a, b in gaps:
    if g1 - g0 < 100e3 or shown[(a, b)] >= 2: continue
    shown[(a, b)] += 1
    print('\ngap %.0f us: %s -> %s' % ((g1 - g0) / 1e3, a, b))
    for s, e, f in api:
        if e > g0 and s < g1:
            print('   +%7.0f us  %-34s %7.0f us' % ((s - g0) / 1e3, f, (e - s) / 1e3))
