#!/usr/bin/env python3
"""Duration of the seed-sort passes (big grids) by where in a report launch they start: rocprofv3 kernel trace."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], int(r['Grid_Size_X'])) for r in rows)
rep = [(s, e) for s, e, n, g in ev if 'report2_kernel' in n][-10:]
for pat, name in (('onesweep_pass<unsigned long long, unsigned int, false, false', 'seed sort pass'), ('merge_sweep', 'merge_sweep'), ('merge_emit', 'merge_emit')):
    bins = collections.defaultdict(list); out = []
    for s, e, n, g in ev:
        if pat in n and g > 1000000 and s > rep[0][0]:
            for rs, re in rep:
                if rs <= s < re:
                    bins[int(5 * (s - rs) / (re - rs))].append((e - s) / 1e3); break
            else:
                out.append((e - s) / 1e3)
    print(name, ' '.join('ph%d: n=%d avg %.0f us |' % (k, len(v), sum(v) / len(v)) for k, v in sorted(bins.items())), 'outside: n=%d avg %.0f us' % (len(out), sum(out) / max(1, len(out))))
