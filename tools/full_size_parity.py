#!/usr/bin/env python3
"""Parity at the north-star size: the CPU port of the reference (oracle/, bit-exact to the reference's arrays on the
golden fixtures) against the HIP path on the FULL synthetic matrix, same initialisation (seed 123), same number of
iterations; Theta, Beta and the train llk.  Two flavours of the port: as is (numpy's float32 row-by-row column sums,
PXI:236,255) and with float64 column sums (the reference's own sums are ~1e-4 off at 1e6 rows).

    python tools/full_size_parity.py [workload=c3] [iterations=3]        (phi for C3: 9.7 GB of host memory)
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from hpfrec_amd import cavi  # noqa: E402
from hpfrec_amd.ops_hip import HipOps  # noqa: E402
from oracle import hpf_oracle as O  # noqa: E402  (the checker)

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
its = int(sys.argv[2]) if len(sys.argv) > 2 else 3
nU, nI, nnz_t, k, label = bench.WORKLOADS[wl]
dev = torch.device("cuda", 0)
iu, ii, y = bench.synth_on_device(nU, nI, nnz_t, dev)
Y, IU, II = O._f32(y.cpu().numpy()), O._ind(iu.cpu().numpy()), O._ind(ii.cpu().numpy())
hy = O.Hyper(k, 0.3, 0.3, 1.0, 0.3, 0.3, 1.0)
st0 = O.State(nU, nI, hy, 123)
init = {n: getattr(st0, n).copy() for n in O.State.names}

# HIP path
m = cavi.FullBatchCavi(HipOps(dev), dev, iu, ii, y, nU, nI, cavi.Hyper(k, 0.3, 0.3, 1.0, 0.3, 0.3, 1.0))
del iu, ii, y
m.load_state(init["Gamma_shp"], init["Gamma_rte"], init["Lambda_shp"], init["Lambda_rte"], init["k_rte"], init["t_rte"],
             init["Theta"], init["Beta"])
for _ in range(its):
    m.iterate(True)
got = {n: m.fetch(n) for n in ("Theta", "Beta")}
t = m.llk_terms(False)
llk_gpu = float(t[0] - m.colsum_dot())
del m
torch.cuda.empty_cache()

cores = O.max_threads()
phi = np.empty((Y.shape[0], k), dtype=np.float32)
print("%s: %d nonzeros, k=%d, %d iterations; port on %d threads" % (label, Y.shape[0], k, its, cores), flush=True)
for name, exact in (("port as is (float32 row-by-row column sums, as numpy does for the reference)", False),
                    ("port with float64 column sums", True)):
    st = O.State(nU, nI, hy, 123)
    t0 = time.time()
    for _ in range(its):
        O.cavi_iteration(st, hy, Y, IU, II, phi, 0, cores, exact_colsums=exact)
    dt = (time.time() - t0) / its
    llk_cpu = float(O.train_llk(st, Y, IU, II, cores)[0])
    dev_rel = {n: float(np.max(np.abs(got[n] - getattr(st, n)) / np.abs(getattr(st, n)))) for n in got}
    frob = {n: float(np.linalg.norm((got[n] - getattr(st, n)).astype(np.float64)) /
                     np.linalg.norm(getattr(st, n).astype(np.float64))) for n in got}
    print("%s: %.1f s/iteration; max rel dev Theta %.2e Beta %.2e; rel Frobenius Theta %.2e Beta %.2e; "
          "train llk port %.9g gpu %.9g (rel %.1e)" % (name, dt, dev_rel["Theta"], dev_rel["Beta"], frob["Theta"],
                                                       frob["Beta"], llk_cpu, llk_gpu, abs(llk_gpu / llk_cpu - 1)),
          flush=True)
