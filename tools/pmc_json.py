#!/usr/bin/env python3
"""profiles/pmc_<workload>_n1.json from the two rocprofv3 --pmc passes of tools/gpu_profile.sh (FETCH_SIZE and
WRITE_SIZE, separate runs): HBM-side bytes per launch of the dominant kernel, tied to the kernel source it was
measured on (bench.py emits roofline.traffic = null when the library running is built from another source).

usage: pmc_json.py <dir with FETCH_SIZE csv> <dir with WRITE_SIZE csv> <workload> <out.json> [kernel substring
                   [dir with RDREQ + RDREQ_DRAM csv, dir with WRREQ + WRREQ_DRAM csv]]
The optional last two passes split the L2's memory-side requests into those destined for DRAM and the rest (VERDICT r03
item 7): `hbm_bytes` = the corrected bytes x that fraction.  The Infinity Cache is memory-side -- a DRAM-destined request can
still be served by it -- and rocprofv3 lists no counter for it on gfx950, so `mall_hit_rate` is null."""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def avg_counter(d, counter, kern):
    vals = {}
    for p in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(p)):
            if r["Counter_Name"] == counter and kern in r["Kernel_Name"]:
                vals.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    if not vals:
        raise SystemExit("no %s rows for kernel %r under %s" % (counter, kern, d))
    name = max(vals, key=lambda n: sum(vals[n]))          # the instantiation that moves the most bytes
    return name, sum(vals[name]) / len(vals[name]), len(vals[name])


fetch_dir, write_dir, workload, out = sys.argv[1:5]
kern = sys.argv[5] if len(sys.argv) > 5 else "sweep_kernel"
name, fetch_kb, nf = avg_counter(fetch_dir, "FETCH_SIZE", kern)
inst = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")    # e.g. sweep_kernel<16, 1, 1, 8>
_, write_kb, nw = avg_counter(write_dir, "WRITE_SIZE", inst)
doc = {
    "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes of `python bench.py --workload %s --steps 3 "
              "--warmup 1 --no-cpu-baseline --no-events --no-extras` (tools/gpu_profile.sh)" % workload,
    "kernel": name.replace("(anonymous namespace)::", "").split("(")[0],
    "launches_averaged": [nf, nw],
    "fetch_size_kb_per_launch": fetch_kb,
    "write_size_kb_per_launch": write_kb,
    "correction": "gfx950: FETCH_SIZE counts 128-B requests at 64 B for wide coalesced reads -> x2 (MI355X_MICROARCH.md "
                  "section HBM); WRITE_SIZE taken as reported (uncalibrated)",
    "sweep_kernel_hbm_bytes_per_launch": int((2.0 * fetch_kb + write_kb) * 1024),
    "kernel_source_sha16": bench.kernel_source_sha16(),
}
if len(sys.argv) > 7:
    rd_dir, wr_dir = sys.argv[6], sys.argv[7]
    try:
        _, rd_all, _ = avg_counter(rd_dir, "TCC_EA0_RDREQ_sum", inst)
        _, rd_dram, _ = avg_counter(rd_dir, "TCC_EA0_RDREQ_DRAM_sum", inst)
        _, wr_all, _ = avg_counter(wr_dir, "TCC_EA0_WRREQ_sum", inst)
        _, wr_dram, _ = avg_counter(wr_dir, "TCC_EA0_WRREQ_DRAM_sum", inst)
        fr, fw = rd_dram / max(rd_all, 1.0), wr_dram / max(wr_all, 1.0)
        doc.update({
            "read_requests_per_launch": {"all": rd_all, "destined_for_dram": rd_dram, "fraction": fr},
            "write_requests_per_launch": {"all": wr_all, "destined_for_dram": wr_dram, "fraction": fw},
            "hbm_bytes": int((2.0 * fetch_kb * fr + write_kb * fw) * 1024),
            "hbm_bytes_note": "the corrected FETCH_SIZE / WRITE_SIZE bytes x the fraction of the L2's memory-side requests "
                              "destined for DRAM (TCC_EA0_RDREQ_DRAM / TCC_EA0_WRREQ_DRAM): an UPPER bound on what HBM itself "
                              "moved -- the Infinity Cache sits on the memory side and serves part of them",
            "mall_hit_rate": None,
            "mall_hit_rate_note": "rocprofv3 (ROCm 7.2, gfx950) lists no Infinity-Cache / MALL counter "
                                  "(profiles/r04_rocprofv3_memory_counters_available.txt)"})
    except SystemExit as e:
        doc["hbm_bytes"], doc["hbm_bytes_note"] = None, "DRAM-destined request passes missing: %s" % e
json.dump(doc, open(out, "w"), indent=2)
print(json.dumps(doc))
