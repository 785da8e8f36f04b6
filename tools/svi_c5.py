#!/usr/bin/env python3
"""BASELINE config C5 smoke: stochastic VI on the C3-shaped matrix, users_per_batch = items_per_batch = 65536,
k = 200, through fit_hpf (2 epochs: one item epoch, one user epoch).  Prints wall time and sanity checks."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from hpfrec_amd import cython_loops_float as be  # noqa: E402

nU, nI, nnz_t, _, _ = bench.WORKLOADS["c3"]
k = 200
epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda", 0)
iu, ii, y = bench.synth_on_device(nU, nI, nnz_t, dev)
Y, IU, II = y.cpu().numpy(), iu.cpu().numpy().astype(np.uint64), ii.cpu().numpy().astype(np.uint64)
del iu, ii, y
torch.cuda.empty_cache()


def run(n):
    Theta = np.empty((nU, k), np.float32)
    Beta = np.empty((nI, k), np.float32)
    t0 = time.time()
    i, temp, llk = be.fit_hpf(0.3, 0.3, 1.0, 0.3, 0.3, 1.0, Y, IU, II, Theta, Beta, n, "maxiter", n, 1e-3, 65536, 65536,
                              lambda x: 1 / np.sqrt(x + 2), 0, np.zeros(1, np.uint64), "", 123, 1, 1, 1, 0,
                              np.empty(0, np.float32), np.empty(0, np.uint64), np.empty(0, np.uint64), 0, 1, 0)
    dt = time.time() - t0
    ok = bool(np.isfinite(Theta).all() and np.isfinite(Beta).all() and (Theta > 0).all() and (Beta > 0).all())
    assert ok
    return dt, float(llk)


os.environ["HPF_TIMING"] = "1"
from hpfrec_amd import svi  # noqa: E402

run(2)                      # warm: code objects, allocator
t_a, _ = run(2)
t_b, llk = run(2 + epochs)
loop = dict(svi.SVI_TIMINGS)
print("C5 SVI (C3 matrix, k=%d, %d user batches / %d item batches of 65536 rows per epoch, nnz=%d): %d epochs in %.2f s wall "
      "incl. upload/layout/download; steady state %.1f ms per epoch (%.2f ms per batch); llk=%.6g"
      % (k, -(-nU // 65536), -(-nI // 65536), Y.shape[0], 2 + epochs, t_b, (t_b - t_a) / epochs * 1e3,
         (t_b - t_a) / epochs * 1e3 / ((-(-nU // 65536) + -(-nI // 65536)) / 2.0), llk))
per_epoch = loop["seconds"] / loop["epochs"] * 1e3
print("epoch loop of the last fit alone (device-synchronised; includes the one llk check at the last epoch): %d epochs in "
      "%.3f s = %.1f ms per epoch, %.2f ms per batch; host time in it: %.3f s preparing batches, %.3f s issuing their kernels"
      % (loop["epochs"], loop["seconds"], per_epoch, per_epoch / ((-(-nU // 65536) + -(-nI // 65536)) / 2.0),
         loop["host_prepare_s"], loop["host_issue_s"]))
print("phases of the last fit [s]: %s" % {p: round(v, 3) for p, v in be.FIT_TIMINGS.items()})
# roofline-style line: algorithmic bytes of an epoch over its time.  Per batch of B rows with n nonzeros touching R rows of
# the other side (fp32, int32 ids; every gather counted once): the two sweeps n*(8 + 8k) (ids, counts, one gathered row and
# one accumulated row per nonzero and side); the reference's per-batch whole-table statements (PXI:300,318,322 / 352,370,374):
# the batch side's rates, means and shapes of ALL its rows -- (nU or nI)*12k -- and the other side's means -- *8k; E rows and
# accumulators of the touched rows (B + R)*16k.  Summed over an epoch's batches: sum n = nnz, sum B = rows of the side.
nb_u, nb_i = -(-nU // 65536), -(-nI // 65536)
nnz = Y.shape[0]
R_u, R_i = min(nI, nnz // nb_u), min(nU, nnz // nb_i)       # (upper bounds on the other side's rows per batch)
b_user_epoch = nnz * (8 + 8 * k) + nb_u * (nU * 12 * k + nI * 8 * k + R_u * 16 * k) + nU * 16 * k
b_item_epoch = nnz * (8 + 8 * k) + nb_i * (nI * 12 * k + nU * 8 * k + R_i * 16 * k) + nI * 16 * k
b_epoch = (b_user_epoch + b_item_epoch) / 2.0
print("roofline (algorithmic bytes per epoch, mean of a user and an item epoch: %.1f GB) / %.1f ms per epoch = %.2f TB/s = "
      "%.0f %% of the 8 TB/s HBM peak" % (b_epoch / 1e9, per_epoch, b_epoch / (per_epoch * 1e-3) / 1e12,
                                            100 * b_epoch / (per_epoch * 1e-3) / 8e12))
