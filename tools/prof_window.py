#!/usr/bin/env python3
"""Per-kernel totals of a WINDOW of a rocprofv3 kernel trace (csv): from the end of the last launch whose name contains
<start-marker> to the end of the last launch whose name contains <end-marker> -- e.g. the epoch loop of the last fit of
tools/svi_c5.py = after its last `uniform_rows_kernel` (the initial tables) up to its last `svi_side_kernel`.  Answers
"which copies sit INSIDE the loop" (VERDICT r03): a whole-run summary mixes the loop with the uploads and downloads.
usage: prof_window.py <dir with *_kernel_trace.csv> <start-marker> <end-marker> [divide totals by N (epochs)]"""
import csv
import glob
import os
import sys


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:64]


path = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r.get("Kernel_Name") or r.get("Name")      # noqa: E731
starts = [i for i, r in enumerate(rows) if sys.argv[2] in name(r)]
ends = [i for i, r in enumerate(rows) if sys.argv[3] in name(r)]
lo, hi = starts[-1] + 1, ends[-1]
div = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
win = rows[lo: hi + 1]
span = (int(win[-1]["End_Timestamp"]) - int(win[0]["Start_Timestamp"])) / 1e6
tot = {}
for r in win:
    d = tot.setdefault(short(name(r)), [0, 0.0])
    d[0] += 1
    d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
busy = sum(v[1] for v in tot.values())
print("window: %d launches over %.2f ms (%.2f ms of kernel time summed); per %s:" % (len(win), span, busy, "unit (/%g)" % div))
print("%-64s %8s %10s %8s" % ("kernel", "calls", "ms", "share"))
for n, (c, ms) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("%-64s %8.1f %10.3f %7.1f%%" % (n, c / div, ms / div, 100 * ms / busy))
