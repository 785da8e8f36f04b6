#!/usr/bin/env python3
"""Timeline of the last N kernels of a rocprofv3 kernel trace (csv): start offset, duration, queue, name -- to see
which launches of a multi-stream iteration overlap and where the device idles.
usage: prof_timeline.py <dir with *_kernel_trace.csv> [N=60] [skip the last M=0 kernels]"""
import csv
import glob
import os
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    head = n.split("(")[0]
    return head[:70]


path = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
if skip:
    rows = rows[:-skip]
rows = rows[-(int(sys.argv[2]) if len(sys.argv) > 2 else 60):]
t0 = int(rows[0]["Start_Timestamp"])
qs = {}
print("%10s %10s %9s  %-5s %s" % ("start_us", "end_us", "dur_us", "queue", "kernel"))
for r in rows:
    q = r.get("Queue_Id", "?")
    qs.setdefault(q, len(qs))
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%10.1f %10.1f %9.1f  q%-4d %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, qs[q], short(r.get("Kernel_Name") or r.get("Name"))))
