#!/usr/bin/env python3
"""topN / predict / predict_factors latency on a C3-sized model (1M users x 380k items, k=50) as a short fit leaves
it -- state and seen-items list on the device: the serving-side numbers of DESIGN.md section 9 (reference: topN
45.8 ms, NB:604-605)."""
import os
import sys
import time

import numpy as np
import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hpfrec_amd import HPF  # noqa: E402

import torch  # noqa: E402
import bench  # noqa: E402

# a C3-sized model as a short fit leaves it: state, seen-items list and all on the device (hpfrec_amd/resident.py)
nU, nI, nnz_t, k, _ = bench.WORKLOADS["c3"]
rs = np.random.RandomState(0)
iu, ii, y = bench.synth_on_device(nU, nI, nnz_t, torch.device("cuda", 0))
df = pd.DataFrame({"UserId": iu.cpu().numpy(), "ItemId": ii.cpu().numpy(), "Count": y.cpu().numpy()})
del iu, ii, y
m = HPF(k=k, reindex=False, verbose=False, maxiter=3, check_every=None, random_seed=1).fit(df)
del df
nU, nI = int(m.nusers), int(m.nitems)


def timeit(f, n=20):
    f()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    return (time.perf_counter() - t0) / n * 1e3


print("topN(n=10, exclude_seen=True):   %.2f ms/query" % timeit(lambda: m.topN(user=12345, n=10)))
print("topN(n=10, exclude_seen=False):  %.2f ms/query" % timeit(lambda: m.topN(user=777, n=10, exclude_seen=False)))
t0 = time.perf_counter()
th, be_ = m._state.peek_host("Theta"), m._state.peek_host("Beta")     # (peek: package-internal read, nothing handed out)
t0 = time.perf_counter()
ref = np.argsort(-(th[777].dot(be_.T)))[:10]
print("  (host numpy GEMV+argsort for the same query: %.1f ms; ids agree: %s)"
      % ((time.perf_counter() - t0) * 1e3, list(ref) == list(m.topN(user=777, n=10, exclude_seen=False))))
pu, pi = rs.randint(nU, size=1000), rs.randint(nI, size=1000)
print("predict(1000 pairs):             %.2f ms" % timeit(lambda: m.predict(pu, pi)))
new = pd.DataFrame({"ItemId": rs.choice(nI, size=40, replace=False), "Count": rs.randint(1, 5, size=40)})
for _ in range(3):
    m.predict_factors(new.copy())
print("predict_factors(40 items):       %.2f ms" % timeit(lambda: m.predict_factors(new.copy()), n=20))
big = pd.DataFrame({"ItemId": rs.choice(nI, size=1000, replace=False), "Count": rs.randint(1, 5, size=1000)})
print("predict_factors(1000 items):     %.2f ms   (reference, 8 CPU threads, k=50, smaller model: 3.5 ms for 40 items, "
      "27 ms for 1000 -- profiles/r02_cpu_calibration.txt)" % timeit(lambda: m.predict_factors(big.copy()), n=20))
print("predict_factors(40 items) again: %.2f ms" % timeit(lambda: m.predict_factors(new.copy()), n=20))
