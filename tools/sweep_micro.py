#!/usr/bin/env python3
"""Isolated launch times of the two sweeps of ONE rank of an N-way run (no other stream active): the floor the
sharded iteration is held against.  usage: sweep_micro.py [N=8] [workload=c3]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from hpfrec_amd import cavi, layout  # noqa: E402
from hpfrec_amd import cython_loops_float as backend  # noqa: E402
from hpfrec_amd.ops_hip import HipOps  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
wl = sys.argv[2] if len(sys.argv) > 2 else "c3"
dev = torch.device("cuda", 0)
nU, nI, nnz_t, k, _ = bench.WORKLOADS[wl]
iu, ii, y = bench.synth_on_device(nU, nI, nnz_t, dev)
lu, li, ly, (u0, u1) = cavi.shard_users(iu, ii, y, nU, 0, world)
del iu, ii, y
ops = HipOps(dev)
hy = cavi.Hyper(k, 0.3, 0.3, 1.0, 0.3, 0.3, 1.0)
m = cavi.FullBatchCavi(ops, dev, lu, li, ly, u1 - u0, nI, hy)
Theta = np.empty((nU, k), np.float32)
Beta = np.empty((nI, k), np.float32)
init = backend.initialize_parameters(Theta, Beta, 123, 0.3, 0.3, 1.0, 0.3, 0.3, 1.0)
s = slice(u0, u1)
m.load_state(init[0][s], init[1][s], init[2], init[3], init[4][s], init[5], Theta[s], Beta)
for _ in range(3):
    m.iterate(True)
ld = m.ld
print("rank 0 of %d, %s: %d users, %d nnz, %d user segments, %d item segments (%d split/empty item rows)"
      % (world, wl, m.nU, m.nnz, m.users.nseg, m.items.nseg, m.items.nmulti))


def timed(fn, n=30):
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


acc = torch.zeros((m.nI, k), device=dev)
gb = m.nnz * ld * 4 / 1e9
for bpc in (4, 8, 16, 32):
    blocks = ops.cu_count * bpc
    for variant in (0, 1):
        m.items.short_rows = variant
        t = timed(lambda: ops.sweep(m.items, m.eB, m.eT, m.part_i, k, ld, acc_rows=acc, acc_ld=k, grid_blocks=blocks))
        print("item sweep  bpc=%2d short_rows=%d: %7.1f us  (%.2f TB/s gathered)" % (bpc, variant, t, gb / t * 1e3))
for bpc in (4, 8, 16, 32):
    blocks = ops.cu_count * bpc
    for variant in (0, 1):
        m.users.short_rows = variant
        t = timed(lambda: ops.sweep(m.users, m.eT, m.eB, m.part_u, k, ld, grid_blocks=blocks))
        print("user sweep (plain) bpc=%2d short_rows=%d: %7.1f us  (%.2f TB/s gathered)" % (bpc, variant, t, gb / t * 1e3))
hyp = m.hy
for bpc in (4, 8, 16):
    g = ops.sweep_grid(m.users.nseg, ops.cu_count * bpc)
    csp = torch.zeros((g, ld), device=dev)
    t = timed(lambda: ops.sweep_finalize(m.users, m.eT, m.eB, m.part_u, m.eT_next, m.Gamma_shp, None, m.Theta, m.k_rte,
                                         m.csB, csp, hyp.a, hyp.k_shp, hyp.add_k_rte, k, ld, rs_prev=m.k_rte_prev))
    print("user sweep+finalize bpc=%2d: %7.1f us  (%.2f TB/s gathered)" % (bpc, t, gb / t * 1e3))
