#!/usr/bin/env python3
"""The apply half of the item finalizer (hpf_hip_item_apply_rows_f32) ALONE on an idle GPU, at the shapes of an 8-rank C3
iteration, over launch grids, warm (back to back) and cold (after a 1 GB copy): what the kernel itself costs against its
place in the iteration's timeline (47 us there for 176 MB of traffic at C3's 380000 items: 3.8 TB/s).  Also colsum_reduce
over that many partial rows (the launch that follows it).

    python tools/apply_probe.py [world] [nI] [k]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hpfrec_amd import _lib, ops_hip  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nI = int(sys.argv[2]) if len(sys.argv) > 2 else 380000
k = int(sys.argv[3]) if len(sys.argv) > 3 else 50
dev = torch.device("cuda", 0)
ops = ops_hip.HipOps(dev)
ld, sld = _lib.ld_for_k(k), ops.gather_payload_ld(k)
step = 4 * world
hi = -(-nI // step) * step
half = (hi // 2 // step) * step
ranges = [(0, half), (half, hi)]
total = sum((b - a) // world for a, b in ranges)
g = torch.Generator(device=dev).manual_seed(1)
recv = torch.rand((world * total, sld), device=dev, generator=g) + 0.5
recv[:, k + 1:] = 0
shp_own = torch.rand((total, ld), device=dev, generator=g) + 0.3
eB = torch.zeros((hi, ld), device=dev)
rs = torch.zeros(hi, device=dev)
cs = torch.rand(ld, device=dev, generator=g) * 100
out = torch.zeros(ld, device=dev)


def timed(fn, reps=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


base = ops.finalize_grid(nI)
print("world %d, %d items (%d rows per rank), k %d: bytes moved %.1f MB; finalize_grid %d"
      % (world, nI, total, k, (world * total * sld + hi * ld) * 4 / 1e6, base))
for mult in (0.25, 0.5, 1, 2, 4):
    grid = world * max(len(ranges), -(-int(mult * base) // world))
    part = torch.zeros((grid, ld), device=dev)
    t = timed(lambda: ops.item_apply_rows(recv, shp_own, eB, None, None, rs, cs, part, 0.3, k, ld, 0, world, nI, ranges))
    t2 = timed(lambda: ops.colsum_reduce(part, out, ld))
    print("  grid %5d (x%.2f): apply %6.1f us back to back (%.2f TB/s)   colsum_reduce over its partials %5.1f us"
          % (grid, mult, t, (world * total * sld + hi * ld) * 4 / t / 1e6, t2))

# the same launches COLD: a 1 GB copy between them evicts L2 and the memory-side cache, as the sweeps of an iteration do
big_a = torch.empty(1 << 28, device=dev)
big_b = torch.empty(1 << 28, device=dev)


def timed_cold(fn, reps=20):
    ts = []
    for _ in range(reps):
        big_b.copy_(big_a)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


empty = timed_cold(lambda: None)
print("cold (after a 1 GB copy), medians; an empty event pair reads %.1f us" % empty)
for mult in (0.5, 1, 2, 4):
    grid = world * max(len(ranges), -(-int(mult * base) // world))
    part = torch.zeros((grid, ld), device=dev)
    t = timed_cold(lambda: ops.item_apply_rows(recv, shp_own, eB, None, None, rs, cs, part, 0.3, k, ld, 0, world, nI, ranges))
    t2 = timed_cold(lambda: ops.colsum_reduce(part, out, ld))
    print("  grid %5d (x%.2f): apply %6.1f us   colsum_reduce %5.1f us" % (grid, mult, t, t2))
