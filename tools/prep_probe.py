#!/usr/bin/env python3
"""Cost of the index work of one stochastic batch on an otherwise idle GPU (C3 matrix, 65,536-row batches): the
tensor-library path (gather_rows + BatchSide) vs svi.batch_sides, user batches and item batches."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from hpfrec_amd import layout, svi  # noqa: E402
from hpfrec_amd.ops_hip import HipOps  # noqa: E402

nU, nI, nnz_t, _, _ = bench.WORKLOADS["c3"]
dev = torch.device("cuda", 0)
ops = HipOps(dev)
iu, ii, y = bench.synth_on_device(nU, nI, nnz_t, dev)
users, items, _ = layout.build_sides(iu, ii, y, nU, nI)
hp = (users.indptr.cpu().numpy(), items.indptr.cpu().numpy())
staging = svi.PinnedStaging(dev)
rs = np.random.RandomState(0)
for name, side, n, n_other, hptr in (("user batch", users, nU, nI, hp[0]), ("item batch", items, nI, nU, hp[1])):
    chunks = [rs.permutation(n)[:65536].astype(np.uint64) for _ in range(6)]

    def fast():
        for c in chunks:
            svi.batch_sides_finish(svi.batch_sides_start(ops, side, hptr, c, n_other, staging=staging))

    def lib():
        for c in chunks:
            rows = torch.sort(svi._dev_ids(c, dev)).values
            br, bc, by = svi.gather_rows(side, rows)
            svi.BatchSide(br, bc, by, grouped=True)
            svi.BatchSide(bc, br, by)

    for label, f in (("svi.batch_sides", fast), ("tensor-library path", lib)):
        f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        f()
        torch.cuda.synchronize()
        print("%s, %-20s %.2f ms per batch (wall, idle GPU)" % (name, label + ":", (time.perf_counter() - t0) / len(chunks) * 1e3))
