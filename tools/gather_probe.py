#!/usr/bin/env python3
"""Gather-bandwidth ceiling for the sweep's access pattern (256-byte rows, 8 gathers in flight per wave, no
arithmetic): uniform and power-law row ids over tables of several sizes.  Run on the GPU box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hpfrec_amd import _lib  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda", 0)
n = 48 * 1024 * 1024
sink = torch.zeros(1, device=dev)
g = torch.Generator(device=dev)
g.manual_seed(0)
print("rows  table_MB  distribution  GB/s(256B per gather)")
for rows in (16_384, 380_000, 1_000_000, 4_000_000):
    tab = torch.rand((rows, 64), device=dev)
    for dist in ("uniform", "power2.5"):
        r = torch.rand(n, generator=g, device=dev, dtype=torch.float64)
        if dist != "uniform":
            r = r.pow(2.5)
        idx = torch.clamp((rows * r).to(torch.int32), max=rows - 1)
        best = 0.0
        for rep in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.check(L.hpf_hip_gather_probe_f32(idx.data_ptr(), n, tab.data_ptr(), sink.data_ptr(), 2048,
                                                  torch.cuda.current_stream().cuda_stream), "probe")
            e1.record()
            torch.cuda.synchronize()
            best = max(best, n * 256 / (e0.elapsed_time(e1) * 1e-3) / 1e9)
        print("%8d %8.1f  %-10s %8.0f" % (rows, rows * 256 / 2 ** 20, dist, best))
