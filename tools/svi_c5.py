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
