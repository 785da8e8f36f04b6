#!/usr/bin/env python3
"""Scale probe: the C3 shape multiplied by a factor (default 10: 10M users x 3.8M items, ~480M nonzeros, k=50),
initialised on the device, a few iterations, then the per-row phi-mass identity and timing.  Exercises 64-bit
offsets (nnz*ld and nseg*ld far beyond 2^31) and the memory footprint on one 288 GB GPU."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from hpfrec_amd import cavi  # noqa: E402
from hpfrec_amd.ops_hip import HipOps  # noqa: E402

f = int(sys.argv[1]) if len(sys.argv) > 1 else 10
k = 50
nU, nI, nnz_t = 1_000_000 * f, 380_000 * f, 48_000_000 * f
dev = torch.device("cuda", 0)
t0 = time.time()
iu, ii, y = bench.synth_on_device(nU, nI, nnz_t, dev)
nnz = int(iu.shape[0])
ysum_u = torch.zeros(nU, dtype=torch.float64, device=dev).index_add_(0, iu, y.double())
print("generated %d nonzeros in %.1f s; peak mem %.1f GB" % (nnz, time.time() - t0, torch.cuda.max_memory_allocated() / 2 ** 30))
ops = HipOps(dev)
hy = cavi.Hyper(k, 0.3, 0.3, 1.0, 0.3, 0.3, 1.0)
t0 = time.time()
m = cavi.FullBatchCavi(ops, dev, iu, ii, y, nU, nI, hy)
del iu, ii, y
torch.cuda.empty_cache()
print("layout built in %.1f s: %d user segments, %d item segments" % (time.time() - t0, m.users.nseg, m.items.nseg))
g = torch.Generator(device=dev)
g.manual_seed(123)
for shp, rte, fac, n in ((m.Gamma_shp, m.Gamma_rte, m.Theta, nU), (m.Lambda_shp, m.Lambda_rte, m.Beta, nI)):
    rte[:, :k] = 0.3 + 0.01 * torch.rand((n, k), generator=g, device=dev)
    shp[:, :k] = 0.3 + 0.01 * torch.rand((n, k), generator=g, device=dev)
    fac[:, :k] = shp[:, :k] / rte[:, :k]
m.k_rte.fill_(1.0)
m.t_rte.fill_(1.0)
m.refresh_expectations()
for _ in range(2):
    m.iterate()
torch.cuda.synchronize()
t0 = time.time()
steps = 5
for _ in range(steps):
    m.iterate()
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
b_iter = nnz * (8 + 8 * k) + nU * (12 + 20 * k) + nI * (4 + 24 * k)
gu = m.Gamma_shp[:, :k].double().sum(dim=1) - k * float(hy.a)
err = float(((gu - ysum_u).abs() / ysum_u.clamp_min(1)).max())
ok = bool(torch.isfinite(m.Theta).all() and torch.isfinite(m.Beta).all())
print("x%d: %.1f ms/iteration (%.1f it/s), %.1f%% of the 8 TB/s algorithmic roofline, phi-mass identity max rel err %.1e, "
      "finite=%s, peak mem %.1f GB" % (f, dt * 1e3, 1 / dt, 100 * b_iter / dt / 8e12, err, ok,
                                      torch.cuda.max_memory_allocated() / 2 ** 30))
assert ok and err < 1e-4
