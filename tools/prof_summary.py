#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel trace and/or PMC counter collection) into a small
per-kernel summary (markdown-ish text) that can be committed under profiles/.

usage: prof_summary.py <dir with rocprofv3 csv files> [name filter substring ...]
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    return name if len(name) < 110 else name[:107] + "..."


def kernel_trace(path, filt):
    rows = list(csv.DictReader(open(path)))
    by = defaultdict(list)
    for r in rows:
        n = r.get("Kernel_Name") or r.get("Name")
        if filt and not any(f in n for f in filt):
            continue
        by[n].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    print("## kernel trace: %s" % os.path.basename(path))
    print("%-110s %7s %12s %12s %12s" % ("kernel", "calls", "avg_us", "min_us", "total_ms"))
    tot = sum(sum(e - s for s, e in v) for v in by.values())
    for n, v in sorted(by.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
        d = [(e - s) / 1e3 for s, e in v]
        print("%-110s %7d %12.1f %12.1f %12.3f  (%.1f%%)" % (short(n), len(d), sum(d) / len(d), min(d), sum(d) / 1e3,
                                                            100.0 * sum(d) * 1e3 / max(tot, 1)))
        if "sweep_kernel" in n and len(d) >= 4:
            v2 = sorted(v)
            ev = [(e - s) / 1e3 for s, e in v2[0::2]]
            od = [(e - s) / 1e3 for s, e in v2[1::2]]
            print("    launches alternate user-side/item-side: even avg %.1f us, odd avg %.1f us" %
                  (sum(ev) / len(ev), sum(od) / len(od)))


def counters(path, filt):
    rows = list(csv.DictReader(open(path)))
    by = defaultdict(lambda: defaultdict(list))
    for r in rows:
        n = r["Kernel_Name"]
        if filt and not any(f in n for f in filt):
            continue
        by[n][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    print("## counters: %s" % os.path.basename(path))
    for n, cs in by.items():
        for c, v in cs.items():
            vals = [x for _, x in sorted(v)]
            line = "%-90s %-14s calls=%d avg=%.6g" % (short(n)[:90], c, len(vals), sum(vals) / len(vals))
            if "sweep_kernel" in n and len(vals) >= 4:
                ev, od = vals[0::2], vals[1::2]
                line += "  [user-side avg=%.6g item-side avg=%.6g]" % (sum(ev) / len(ev), sum(od) / len(od))
            print(line)


def main():
    d = sys.argv[1]
    filt = sys.argv[2:]
    for p in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
        kernel_trace(p, filt)
    for p in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        counters(p, filt)
    for p in sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)):
        print("## rocprofv3 --stats: %s" % os.path.basename(p))
        for i, line in enumerate(open(p)):
            if i < 14:
                print(line.rstrip()[:260])


if __name__ == "__main__":
    main()
