#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV, attributed to the pair
(kernel before the gap -> kernel after it).  usage: gap_analysis.py <kernel_trace.csv> [skip-fraction]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
cut = t0 + skip * (t1 - t0)
rows = [r for r in rows if int(r["Start_Timestamp"]) >= cut]
gaps = collections.defaultdict(lambda: [0, 0])
busy = tot = 0
prev = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - s
    if prev is not None and s > prev[1]:
        k = (prev[0].split("(")[0][:34], r["Kernel_Name"].split("(")[0][:34])
        gaps[k][0] += s - prev[1]
        gaps[k][1] += 1
        tot += s - prev[1]
    if prev is None or e > prev[1]:
        prev = (r["Kernel_Name"], e)
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print("span %.1f ms, kernels %.1f ms, idle %.1f ms" % (span / 1e6, busy / 1e6, tot / 1e6))
for k, v in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:22]:
    print("%8.2f ms %5d x  %s -> %s" % (v[0] / 1e6, v[1], k[0], k[1]))
