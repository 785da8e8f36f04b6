#!/usr/bin/env python3
"""topN / predict / predict_factors latency on a C3-sized model (1M users x 380k items, k=50) with random
parameters (no fit needed): the serving-side numbers of DESIGN.md section 9 (reference: topN 45.8 ms, NB:604-605)."""
import os
import sys
import time

import numpy as np
import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hpfrec_amd import HPF  # noqa: E402

nU, nI, k = 1_000_000, 380_000, 50
rs = np.random.RandomState(0)
m = HPF(k=k, reindex=False, verbose=False)
# (as a fitted model holds them: arrays the package created itself and never handed out -- an array assigned through
# the public attribute stays referenced by the caller and is re-uploaded before every device use, hpfrec_amd/resident.py)
for name, arr in (("Theta", rs.gamma(0.3, 1.0, size=(nU, k))), ("Beta", rs.gamma(0.3, 1.0, size=(nI, k))),
                  ("Lambda_shp", rs.uniform(0.3, 5, size=(nI, k))), ("Lambda_rte", rs.uniform(0.3, 5, size=(nI, k)))):
    m._state.set_host(name, arr.astype(np.float32), private=True)
m.nusers, m.nitems, m.is_fitted, m.niter = nU, nI, True, 1
m.seen = np.sort(rs.choice(nI, size=48, replace=False))
m._n_seen_by_user = np.full(nU, 48, dtype=np.int64)
m._st_ix_user = np.zeros(nU, dtype=np.int64)


def timeit(f, n=20):
    f()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    return (time.perf_counter() - t0) / n * 1e3


print("topN(n=10, exclude_seen=True):   %.2f ms/query" % timeit(lambda: m.topN(user=12345, n=10)))
print("topN(n=10, exclude_seen=False):  %.2f ms/query" % timeit(lambda: m.topN(user=777, n=10, exclude_seen=False)))
t0 = time.perf_counter()
ref = np.argsort(-(m._state.host["Theta"][777].dot(m._state.host["Beta"].T)))[:10]
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
