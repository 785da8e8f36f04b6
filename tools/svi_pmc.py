#!/usr/bin/env python3
"""Fabric-side bytes of the SVI epoch loop at C5 from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs,
collected as MI355X_MICROARCH.md prescribes: FETCH_SIZE x2 on gfx950 for wide coalesced reads, KB units) of
`python tools/svi_c5.py <E>`: per kernel of the epoch loop, calls and bytes summed over the run, divided by the run's
epochs (tools/svi_c5.py fits 2 + 2 + (2+E) epochs).  Kernels of the uploads / layout / initial draws are left out by name.
usage: svi_pmc.py <dir FETCH_SIZE> <dir WRITE_SIZE> <E>"""
import csv
import glob
import os
import sys

LOOP = ("sweep_kernel<64", "svi_", "expect_kernel", "colsum_reduce", "segsum_desc", "llk_sweep")


def totals(d, counter):
    out = {}
    for p in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(p)):
            n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            if r["Counter_Name"] == counter and any(t in n for t in LOOP):
                c = out.setdefault(n, [0, 0.0])
                c[0] += 1
                c[1] += float(r["Counter_Value"])
    return out


fetch, write, E = totals(sys.argv[1], "FETCH_SIZE"), totals(sys.argv[2], "WRITE_SIZE"), int(sys.argv[3])
epochs = 2 + 2 + 2 + E
print("SVI at C5, fabric-side bytes per epoch (FETCH_SIZE x2 + WRITE_SIZE, KB -> GB; %d epochs in the run):" % epochs)
print("%-44s %9s %10s %10s" % ("kernel", "calls/ep", "read GB", "write GB"))
tr = tw = 0.0
for n in sorted(fetch, key=lambda k: -fetch[k][1]):
    rd = 2.0 * fetch[n][1] * 1024 / 1e9 / epochs
    wr = write.get(n, [0, 0.0])[1] * 1024 / 1e9 / epochs
    tr, tw = tr + rd, tw + wr
    print("%-44s %9.1f %10.2f %10.2f" % (n[:44], fetch[n][0] / epochs, rd, wr))
print("%-44s %9s %10.2f %10.2f   = %.1f GB per epoch moved on the fabric side; algorithmic bytes of the reference's "
      "statements: 131.0 GB per epoch" % ("total", "", tr, tw, tr + tw))
