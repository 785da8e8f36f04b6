#!/usr/bin/env python3
"""Per-LAUNCH fabric-side rate of the two sweeps of the SVI epoch loop at C5: the kernel trace of one run of
`tools/svi_c5.py E` (durations) joined, dispatch by dispatch in launch order, with the FETCH_SIZE and WRITE_SIZE passes of two
more runs of the same command (the runs issue the same launches in the same order).  Prints the launches of the last item epoch
and the last user epoch: bytes, microseconds, TB/s.
usage: svi_launch_rates.py <trace dir> <FETCH_SIZE dir> <WRITE_SIZE dir>"""
import csv
import glob
import os
import sys


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def trace(d):
    p = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(p)), key=lambda r: int(r["Start_Timestamp"]))
    return [(short(r.get("Kernel_Name") or r.get("Name")), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows]


def counters(d, name):
    out = []
    for p in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(p)) if r["Counter_Name"] == name]
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        out += [(short(r["Kernel_Name"]), float(r["Counter_Value"])) for r in rows]
    return out


tr, fe, wr = trace(sys.argv[1]), counters(sys.argv[2], "FETCH_SIZE"), counters(sys.argv[3], "WRITE_SIZE")
for mode, label in (("sweep_kernel<64, 1, 3", "batch-side sweep (MODE 3)"), ("sweep_kernel<64, 1, 2", "other-side sweep (MODE 2)")):
    t = [u for n, u in tr if n.startswith(mode)]
    f = [v for n, v in fe if n.startswith(mode)]
    w = [v for n, v in wr if n.startswith(mode)]
    n = min(len(t), len(f), len(w))
    assert n > 0 and len(t) == len(f) == len(w), (len(t), len(f), len(w))
    # the last two epochs of the run: 22 launches (16 of a user epoch + 6 of an item epoch, in whichever order they came)
    print("%s: last 22 launches [read GB (FETCH_SIZE x2) | written GB | us | TB/s]" % label)
    tot_b = tot_t = 0.0
    for i in range(n - 22, n):
        rd, wb = 2.0 * f[i] * 1024 / 1e9, w[i] * 1024 / 1e9
        print("   %6.2f | %5.2f | %7.1f | %.2f" % (rd, wb, t[i], (rd + wb) / t[i] * 1e3))
        tot_b += rd + wb
        tot_t += t[i]
    print("   total %.1f GB in %.2f ms = %.2f TB/s" % (tot_b, tot_t / 1e3, tot_b / tot_t * 1e3))
