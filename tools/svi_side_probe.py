#!/usr/bin/env python3
"""Whole-table pass of an SVI step (hpf_hip_svi_side_f32) on a [rows][256] table, k = 200: lazy batch-side form (shapes read,
nothing stored) against a plain read of the same table, with no / few / all rows flagged.
usage: python tools/svi_side_probe.py [rows=1000000]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hpfrec_amd.ops_hip import HipOps  # noqa: E402

dev = torch.device("cuda", 0)
ops = HipOps(dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
k, ld = 200, 256
f32 = dict(dtype=torch.float32, device=dev)
shp = torch.rand((n, ld), **f32) + 0.5
shp[:, k:] = 0
rte = torch.rand((n, ld), **f32) + 0.5
fac = torch.zeros((n, ld), **f32)
acc = torch.rand((n, ld), **f32)
e = torch.rand((n, ld), **f32)
rs = torch.rand(n, **f32) + 1
rsp = torch.zeros(n, **f32)
cs = torch.rand(ld, **f32)
part = torch.zeros((ops.svi_side_grid(n) if hasattr(ops, "svi_side_grid") else 4096, ld), **f32)


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


gb = n * ld * 4 / 1e9
t = timed(lambda: shp.sum())
print("torch sum of the table (%.2f GB read): %.3f ms = %.2f TB/s" % (gb, t, gb / t))
for name, frac in (("no flag array", None), ("no row flagged", 0.0), ("6.5 %% flagged", 0.065), ("all flagged", 1.0)):
    flag = None if frac is None else (torch.rand(n, device=dev) < frac).to(torch.uint8)
    t = timed(lambda: ops.svi_side(n, flag, acc, e, shp, None, None, rs, cs, part, 0.3, 1.0, 0.0, 0.3, 0.3, 0.5, 0.5, 0, 1,
                                   k, ld, rs_prev_out=rsp))
    print("lazy batch side, %-16s: %.3f ms = %.2f TB/s of shapes" % (name, t, gb / t))
for g in [int(x) for x in os.environ.get("PROBE_GRIDS", "").split(",") if x]:
    pg = torch.zeros((g, ld), **f32)
    fl65 = (torch.rand(n, device=dev) < 0.065).to(torch.uint8)
    t = timed(lambda: ops.svi_side(n, fl65, acc, e, shp, None, None, rs, cs, pg, 0.3, 1.0, 0.0, 0.3, 0.3, 0.5, 0.5, 0, 1, k, ld,
                                   rs_prev_out=rsp), reps=20)
    print("lazy batch side, 6.5 %% flagged, grid %5d workgroups: %.3f ms = %.2f TB/s of shapes (%.2f TB/s of the sectors read)"
          % (g, t, gb / t, gb * 13 / 16 / t))
t = timed(lambda: ops.svi_side(n, None, acc, e, shp, rte, fac, rs, cs, part, 0.3, 1.0, 0.0, 0.3, 0.3, 0.5, 0.5, 0, 1, k, ld))
print("stored batch side (rate and mean tables written, 3 x %.2f GB): %.3f ms = %.2f TB/s" % (gb, t, 3 * gb / t))

# the OTHER side of a lazy step: rate_mode 1 (rates of the touched rows blended), E rows written, 99 % of the rows touched
fl = (torch.rand(n, device=dev) < 0.99).to(torch.uint8)
t = timed(lambda: ops.svi_side(n, fl, acc, e, shp, rte, None, rs, cs, part, 0.3, 0.4, 0.6, 0.3, 0.3, 0.5, 0.5, 1, 1, k, ld,
                               e_out=e))
moved = n * 0.99 * 7 * k * 4 / 1e9 + n * 0.01 * 2 * k * 4 / 1e9
print("lazy other side, 99 %% of the rows touched (reads shp, rte, acc, e; writes shp, rte, e: %.2f GB of columns): %.3f ms = "
      "%.2f TB/s" % (moved, t, moved / t))
