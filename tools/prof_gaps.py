#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 kernel trace (csv): where a launch-bound loop loses time.
usage: prof_gaps.py <dir with *_kernel_trace.csv> [last N kernels to analyse, default 600]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:60]


path = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Kernel_Name") or r.get("Name")) for r in csv.DictReader(open(path))]
rows.sort()
rows = rows[-(int(sys.argv[2]) if len(sys.argv) > 2 else 600):]
gaps = defaultdict(list)
busy = 0
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    gaps[(short(n0), short(n1))].append(max(0, s1 - e0))
    busy += e0 - s0
span = rows[-1][1] - rows[0][0]
print("last %d kernels: span %.3f ms, kernels busy %.3f ms (%.1f%%), idle %.3f ms" %
      (len(rows), span / 1e6, busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6))
print("%-62s -> %-62s %6s %10s %10s" % ("after", "before", "n", "avg_us", "total_us"))
for (a, b), v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:25]:
    print("%-62s -> %-62s %6d %10.2f %10.1f" % (a, b, len(v), sum(v) / len(v) / 1e3, sum(v) / 1e3))
