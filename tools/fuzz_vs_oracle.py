#!/usr/bin/env python3
"""Random small problems (odd shapes: 1 user, 1 item, k not a multiple of 4, duplicate pairs, rows without data, explicit
zero counts, one hub item) through fit_hpf on the HIP path vs the CPU oracle, 3 iterations each.

    python tools/fuzz_vs_oracle.py [cases=60] [seed=0]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_host_logic import NAMES, _fit, _maxrel  # noqa: E402
from hpfrec_amd import cython_loops_float as be  # noqa: E402
from oracle import hpf_oracle as O  # noqa: E402

from exact_ref import exact_sums_reference  # noqa: E402  (tests/exact_ref.py)

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = worst_x = 0.0
for c in range(cases):
    nU = int(rs.choice([1, 2, 3, 17, 100, 1000, 5000]))
    nI = int(rs.choice([1, 2, 5, 33, 300, 3000]))
    k = int(rs.choice([1, 2, 3, 5, 7, 30, 31, 32, 33, 50, 64, 65, 100, 129, 200, 257]))
    nnz = int(rs.choice([1, 2, 10, 300, 5000, 40000]))
    iu = (nU * rs.random_sample(nnz) ** rs.choice([1, 2, 3])).astype(np.uint64)
    ii = (nI * rs.random_sample(nnz) ** rs.choice([1, 2, 4])).astype(np.uint64)
    if rs.rand() < 0.3:
        ii[: nnz // 2] = 0                      # a hub item: long, split rows
    Y = (rs.gamma(1, rs.choice([1, 10, 1000]), size=nnz) + 1).astype(np.int64).astype(np.float32)
    its = 3
    _, arrs, _ = _fit(be, Y, iu, ii, nU, nI, k, its)
    st, _ = O.fit_full_batch(Y, iu, ii, nU, nI, k, its, 123)
    w = max(_maxrel(arrs[n], getattr(st, n)) for n in NAMES)
    sx = exact_sums_reference(Y, iu, ii, nU, nI, k, its)
    wx = max(_maxrel(arrs[n], getattr(sx, n)) for n in NAMES)
    wo = max(_maxrel(getattr(st, n), getattr(sx, n)) for n in NAMES)     # the reference's own accumulation noise
    worst, worst_x = max(worst, w), max(worst_x, wx)
    print("case %2d: nU=%5d nI=%5d k=%3d nnz=%6d  HIP vs oracle %.1e | HIP vs float64-sums reference %.1e | oracle vs "
          "float64-sums reference %.1e%s" % (c, nU, nI, k, nnz, w, wx, wo, "" if wx < 5e-5 and (wo < 2e-5 or wx < 0.3 * wo) else "   <-- CHECK"), flush=True)
print("worst over %d cases: HIP vs oracle %.2e, HIP vs float64-sums reference %.2e" % (cases, worst, worst_x))
assert worst_x < 5e-5
print("FUZZ_OK")
