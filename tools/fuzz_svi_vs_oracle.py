#!/usr/bin/env python3
"""Random small problems through the STOCHASTIC epochs of fit_hpf on the HIP path (batches prepared per epoch, the other
side's step fused into its sweep) vs the CPU oracle's restatement of PXI:262-377 (bit-exact to the reference's own SVI
captures): odd shapes -- 1 user, 1 item, k not a multiple of 4, users / items without data, hub rows cut into several
segments, batches of one row, batches larger than the side, a short last batch, batches none of whose rows has data --
user-only, item-only and alternating epochs, 2-4 epochs each.

    python tools/fuzz_svi_vs_oracle.py [cases=60] [seed=0]
"""
import os
import sys

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hpfrec_amd import cython_loops_float as be  # noqa: E402
from hpfrec_amd import layout  # noqa: E402
from oracle import hpf_oracle as O  # noqa: E402

if os.environ.get("FUZZ_STANDIN") == "1":      # (a dry run of this script without a GPU: the tests' numpy stand-in ops)
    import cpu_ops
    be.HipOps = lambda device=None: cpu_ops.CpuOps()
NAMES = ("Theta", "Beta", "Gamma_shp", "Gamma_rte", "Lambda_shp", "Lambda_rte", "k_rte", "t_rte")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = worst64 = 0.0
for c in range(cases):
    nU = int(rs.choice([1, 2, 3, 17, 100, 400, 1500]))
    nI = int(rs.choice([1, 2, 5, 33, 300, 1200]))
    k = int(rs.choice([1, 3, 5, 7, 30, 33, 50, 64, 65, 100, 130, 200]))
    nnz = int(rs.choice([1, 2, 10, 300, 5000, 20000]))
    iu = np.minimum((nU * rs.random_sample(nnz) ** rs.choice([1, 2, 3])).astype(np.int64), max(0, nU - 1 - int(rs.rand() < 0.5)))
    ii = np.minimum((nI * rs.random_sample(nnz) ** rs.choice([1, 2, 4])).astype(np.int64), nI - 1)
    if rs.rand() < 0.3:
        ii[: nnz // 2] = 0                      # a hub item
    df = pd.DataFrame({"u": iu, "i": ii}).drop_duplicates().reset_index(drop=True)      # (item epochs of the reference merge duplicates)
    iu, ii = df["u"].to_numpy().astype(np.uint64), df["i"].to_numpy().astype(np.uint64)
    Y = (rs.gamma(1, rs.choice([1, 10]), size=iu.shape[0]) + 1).astype(np.int64).astype(np.float32)
    mode = rs.choice(["both", "users", "items"])
    upb = int(rs.choice([1, 2, 7, max(1, nU // 3), nU, nU + 5])) if mode != "items" else 0
    ipb = int(rs.choice([1, 3, max(1, nI // 4), nI, nI + 2])) if mode != "users" else 0
    upb, ipb = min(upb, nU), min(ipb, nI)       # (the class replaces larger values, INIT:517-519)
    if (upb and -(-nU // upb) > 60) or (ipb and -(-nI // ipb) > 60):
        upb, ipb = (max(upb, -(-nU // 60)) if upb else 0), (max(ipb, -(-nI // 60)) if ipb else 0)   # keep the oracle quick
    epochs = int(rs.choice([2, 3, 4]))
    layout.SEG_CAP = int(rs.choice([8, 64, 1024]))
    Ys, ius, iis, st = O.svi_inputs_like_reference(Y, iu, ii, nU, nI)
    Theta, Beta = np.empty((nU, k), np.float32), np.empty((nI, k), np.float32)
    i, temp, _ = be.fit_hpf(0.3, 0.3, 1.0, 0.3, 0.3, 1.0, Ys, ius, iis, Theta, Beta, epochs, "maxiter", 0, 1e-3, upb, ipb,
                            lambda x: 1 / np.sqrt(x + 2), 0, st, "", 123, 0, 1, 0, 0, np.empty(0, np.float32),
                            np.empty(0, np.uint64), np.empty(0, np.uint64), 0, 1, 0)
    assert i == epochs - 1
    got = dict(zip(("Gamma_shp", "Gamma_rte", "Lambda_shp", "Lambda_rte", "k_rte", "t_rte"), temp), Theta=Theta, Beta=Beta)
    dev = {}
    for exact in (False, True):
        ref = O.fit_svi(Ys, ius, iis, st, nU, nI, k, epochs, 123, upb, ipb, exact_colsums=exact)
        dev[exact] = max(float(np.max(np.abs(got[n] - getattr(ref, n)) / np.abs(getattr(ref, n)))) for n in NAMES)
    assert all(np.isfinite(got[n]).all() and (got[n] > 0).all() for n in NAMES), c
    worst, worst64 = max(worst, dev[False]), max(worst64, dev[True])
    print("case %2d: nU=%4d nI=%4d k=%3d nnz=%5d %-5s batches %4d/%4d epochs %d cap %4d: HIP vs oracle %.1e | vs oracle with "
          "float64 column sums %.1e%s" % (c, nU, nI, k, iu.shape[0], mode, upb, ipb, epochs, layout.SEG_CAP, dev[False],
                                         dev[True], "" if min(dev.values()) < 1e-4 else "   <-- CHECK"), flush=True)
print("worst over %d cases: HIP vs oracle %.2e, vs oracle with float64 column sums %.2e" % (cases, worst, worst64))
assert min(worst, worst64) < 1e-4
print("FUZZ_SVI_OK")
