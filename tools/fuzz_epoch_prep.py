#!/usr/bin/env python3
"""Random shapes through the epoch-level batch preparation (hpf_hip_svi_epoch_prepare, svi.EpochWorkspace) against the
per-batch one (hpf_hip_svi_batch_prepare, svi.BatchWorkspace) on the GPU: for every batch of a random epoch the same
sizes, flags, own-side descriptors, split-row lists and other-side nonzeros in the same order.  Shapes: 1 .. 5000 rows per
side, 0 .. 200k nonzeros, uniform to heavily skewed ids (hub rows), segment caps 1 .. 1024, 1 .. 255 batches, duplicate
pairs, rows without data.  usage: python tools/fuzz_epoch_prep.py [cases=150] [seed=1]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hpfrec_amd import layout, svi  # noqa: E402
from hpfrec_amd.ops_hip import HipOps  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda", 0)
ops = HipOps(dev)
ld = 32
checked = batches = 0
worst = {"nb": 0, "nnz": 0, "rows": 0}
for case in range(cases):
    nU, nI = int(rs.choice([1, 2, 7, 64, 300, 1000, 5000])), int(rs.choice([1, 3, 50, 400, 2000, 5000]))
    nnz = int(rs.choice([0, 1, 5, 100, 3000, 40000, 200000]))
    pu, pi = rs.choice([1.0, 1.5, 3.0, 6.0]), rs.choice([1.0, 2.0, 4.0, 8.0])
    iu = np.minimum((nU * rs.random_sample(nnz) ** pu).astype(np.int64), nU - 1)
    ii = np.minimum((nI * rs.random_sample(nnz) ** pi).astype(np.int64), nI - 1)
    y = (1 + rs.poisson(1.0, size=nnz)).astype(np.float32)
    cap = int(rs.choice([1, 2, 16, 100, 1024]))
    users, items, _ = layout.build_sides(torch.from_numpy(iu).to(dev), torch.from_numpy(ii).to(dev),
                                         torch.from_numpy(y).to(dev), nU, nI, seg_cap=cap)
    for side, other in ((users, items), (items, users)):
        n_rows = side.nrows
        nb_want = int(rs.choice([1, 2, 3, 16, 64, 65, 200, 255]))
        per = max(1, -(-n_rows // nb_want))
        if -(-n_rows // per) > 255:
            continue
        acc = torch.ones((n_rows, ld), dtype=torch.float32, device=dev)
        acc_b = torch.ones((n_rows, ld), dtype=torch.float32, device=dev)
        ews = svi.EpochWorkspace(ops, side, other, acc, ld, per, seg_cap=cap)
        order = rs.permutation(n_rows).astype(np.int64)
        ews.prepare(ops, torch.from_numpy(order).to(dev))
        assert not ews.overflowed()
        if nnz == 0:          # (the per-batch entry wants a segment list; the epoch entry takes a matrix without data)
            assert int(ews.sizes[:, :6].abs().sum()) == 0 and int(ews.flag_oth.sum()) == 0
            assert np.array_equal(ews.flag_own.sum(dim=0).cpu().numpy(), np.ones(n_rows)) and float(acc.abs().sum()) == 0
            checked += 1
            continue
        bws = svi.BatchWorkspace(ops, side, other, acc_b, ld, per, seg_cap=cap)
        base = 0
        for j in range(ews.nb):
            ids = order[j * per: min(n_rows, (j + 1) * per)]
            bws.prepare(ops, torch.from_numpy(ids).to(dev))
            own_e, oth_e, f_own, f_oth = ews.batch(j)
            se, sb = ews.sizes[j].cpu().numpy(), bws.sizes.cpu().numpy()
            ctx = (case, nU, nI, nnz, cap, ews.nb, j)
            assert np.array_equal(se[:6], sb[:6]) and se[7] == 0 and sb[7] == 0, (ctx, se, sb)
            assert torch.equal(f_own, bws.flag_own) and torch.equal(f_oth, bws.flag_oth), ctx
            assert torch.equal(own_e.segs[: se[0]], bws.b_segs[: sb[0]]), ctx
            assert torch.equal(own_e.multi[: se[1]], bws.b_multi[: sb[1]]), ctx
            got = oth_e.segs[: se[2]].clone()
            got[:, 0] -= base
            assert torch.equal(got, bws.o_segs[: sb[2]]), ctx
            assert torch.equal(oth_e.multi[: se[3]], bws.o_multi[: sb[3]]), ctx
            assert torch.equal(oth_e.idx[base: base + se[4]], bws.o_idx[: sb[4]]), ctx
            assert torch.equal(oth_e.y[base: base + se[4]], bws.o_y[: sb[4]]), ctx
            base += int(se[4])
            batches += 1
        assert base == side.nnz
        deg = (side.indptr[1:] - side.indptr[:-1]).cpu().numpy()
        a = acc.cpu().numpy()
        assert np.all(a[deg == 0] == 0) and np.all(a[deg > 0] == 1)
        checked += 1
        worst = {"nb": max(worst["nb"], ews.nb), "nnz": max(worst["nnz"], nnz), "rows": max(worst["rows"], n_rows)}
print("epoch-level preparation == per-batch preparation on %d random epochs (%d batches compared; up to %d batches per epoch, "
      "%d nonzeros, %d rows): sizes, flags, own-side descriptors, split rows, other-side nonzeros and their order all equal"
      % (checked, batches, worst["nb"], worst["nnz"], worst["rows"]))
