#!/usr/bin/env python3
"""Cost of one stochastic batch's index structures on an idle GPU: svi.BatchWorkspace.prepare (hpf_hip_svi_batch_prepare:
flags, the own side's compacted segment list, the other side's filtered copy -- one host call, 8 launches) for user
batches and item batches of 65,536 rows of the C3 matrix (BASELINE config C5), host issue time and device time."""
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
del iu, ii, y
ld, B = 256, 65536
rs = np.random.RandomState(0)
for name, own, oth, n in (("user batch", users, items, nU), ("item batch", items, users, nI)):
    acc = torch.zeros((n, ld), device=dev)
    ws = svi.BatchWorkspace(ops, own, oth, acc, ld, B)
    perm = torch.from_numpy(rs.permutation(n).astype(np.int64)).to(dev)
    chunks = [perm[b * B: (b + 1) * B] for b in range(min(6, n // B))]
    ws.prepare(ops, chunks[0])
    torch.cuda.synchronize()
    t_issue, t_dev, sizes = [], [], []
    for c in chunks[1:]:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        t0 = time.perf_counter()
        ws.prepare(ops, c)
        t_issue.append((time.perf_counter() - t0) * 1e3)
        e1.record()
        torch.cuda.synchronize()
        t_dev.append(e0.elapsed_time(e1))
        sizes.append(ws.sizes.cpu().tolist())
    s = sizes[-1]
    print("%s of %d rows: host issue %.3f ms, device %.3f ms per batch (runs: %s); last batch: %d nonzeros, %d + %d "
          "segments, %d rows of the other side, workspace bound %d nonzeros"
          % (name, B, np.mean(t_issue), np.mean(t_dev), " ".join("%.3f" % t for t in t_dev), s[4], s[0], s[2], s[5],
             ws.nnz_bound), flush=True)
    del ws, acc
