#!/usr/bin/env python3
"""VERDICT r01 item 5 as an experiment: cut the item-side (CSC) rows at user-range tile boundaries and order the
segments tile-major, so that the gathers of one phase of the launch fall into an eT slice that fits the Infinity
Cache (244 MB table -> T tiles).  Timing of the plain item-side sweep (no finalizer, accumulators to part[]) with
the shipped layout against the tiled ones; layout only, same kernel.  usage: tile_probe.py [workload=c3]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from hpfrec_amd import cavi, layout  # noqa: E402
from hpfrec_amd import cython_loops_float as backend  # noqa: E402
from hpfrec_amd.ops_hip import HipOps  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
dev = torch.device("cuda", 0)
nU, nI, nnz_t, k, _ = bench.WORKLOADS[wl]
iu, ii, y = bench.synth_on_device(nU, nI, nnz_t, dev)
ops = HipOps(dev)
hy = cavi.Hyper(k, 0.3, 0.3, 1.0, 0.3, 0.3, 1.0)
m = cavi.FullBatchCavi(ops, dev, iu, ii, y, nU, nI, hy)
del iu, ii, y
Theta = np.empty((nU, k), np.float32)
Beta = np.empty((nI, k), np.float32)
init = backend.initialize_parameters(Theta, Beta, 123, 0.3, 0.3, 1.0, 0.3, 0.3, 1.0)
m.load_state(init[0], init[1], init[2], init[3], init[4], init[5], Theta, Beta)
m.iterate(True)
ld, it = m.ld, m.items


def timed(side, part, n=20):
    fn = lambda: ops.sweep(side, m.eB, m.eT, part, k, ld, grid_blocks=ops.cu_count * 16)   # noqa: E731
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


class Side:
    pass


def tiled_side(T, min_split, cap=layout.SEG_CAP):
    """Segments of the item side cut at T user-range tile boundaries (rows with >= min_split nonzeros only) and at
    `cap` nonzeros, tile-major order for the cut rows, then the uncut rows."""
    indptr, idx = it.indptr, it.idx.to(torch.int64)
    nnz = idx.shape[0]
    deg = indptr[1:] - indptr[:-1]
    row_of = torch.repeat_interleave(torch.arange(nI, device=dev), deg, output_size=nnz)
    tile = torch.div(idx * T, nU, rounding_mode="floor")
    split_row = deg >= min_split
    first = torch.zeros(nnz, dtype=torch.bool, device=dev)
    first[indptr[:-1][deg > 0]] = True
    cut = first.clone()
    cut[1:] |= (tile[1:] != tile[:-1]) & split_row[row_of[1:]]
    # cap-based cuts inside a piece: position relative to the piece start
    start_pos = torch.where(cut, torch.arange(nnz, device=dev), torch.zeros(1, dtype=torch.int64, device=dev))
    start_pos = torch.cummax(start_pos, 0).values
    cut |= ((torch.arange(nnz, device=dev) - start_pos) % cap) == 0
    begin = torch.nonzero(cut).reshape(-1)
    length = torch.diff(begin, append=torch.tensor([nnz], device=dev))
    rows = row_of[begin]
    seg_tile = torch.where(split_row[rows], tile[begin], torch.full_like(rows, T))   # uncut rows last
    order = torch.argsort(seg_tile * (2 * nnz) + begin)                              # tile-major, then by position
    begin, length, rows = begin[order], length[order], rows[order]
    s = Side()
    s.segs = torch.stack([begin, length | (rows << 32)], dim=1).contiguous()
    s.nseg = int(begin.shape[0])
    s.idx, s.y, s.short_rows = it.idx, it.y, 0
    return s


base = timed(it, m.part_i)
gb = it.nnz * ld * 4 / 1e9
print("%s item-side plain sweep, shipped layout: %d segments, %.1f us (%.2f TB/s gathered)" % (wl, it.nseg, base, gb / base * 1e3))
for T in (2, 4, 8):
    for min_split in (64, 256, 1024):
        s = tiled_side(T, min_split)
        part = torch.empty((s.nseg, ld), dtype=torch.float32, device=dev)
        t = timed(s, part)
        print("T=%d tiles, rows with >= %4d nonzeros cut: %8d segments, %.1f us (%+.1f %%)" % (T, min_split, s.nseg, t, 100 * (t / base - 1)),
              flush=True)
        del s, part
        torch.cuda.empty_cache()
