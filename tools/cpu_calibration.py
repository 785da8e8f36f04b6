"""Calibrate the CPU baseline: time the REAL reference (Cython/OpenMP build in a
scratch copy outside this repo, see tests/golden/make_golden.py) and the oracle port side by side, same
matrix, same thread count.  Build-container only (the reference does not travel to the GPU box).

    python tools/cpu_calibration.py [nU nI nnz k iters]
"""
import contextlib
import io
import os
import sys
import time
import warnings

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("HPFREC_REF_BUILD", "/tmp/hpfrec_oracle")
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import datagen  # noqa: E402
from oracle import hpf_oracle as O  # noqa: E402

nU, nI, nnz, k, iters = (int(v) for v in (sys.argv[1:6] if len(sys.argv) >= 6 else (40000, 380000, 2000000, 50, 4)))
iu, ii, Y = datagen.synthetic_hpf_shaped(nU, nI, nnz, seed=1)
cores = O.max_threads()
print("matrix: %d x %d, %d nnz, k=%d; threads=%d" % (nU, nI, Y.shape[0], k, cores))

sys.path.insert(0, REF)
from hpfrec import HPF  # noqa: E402  (the reference itself, from the scratch build)
sys.path.remove(REF)
df = pd.DataFrame({"UserId": iu.astype(np.int64), "ItemId": ii.astype(np.int64), "Count": Y})
Yc, iuc, iic = O._f32(Y), O._ind(iu), O._ind(ii)
hy = O.Hyper(k, 0.3, 0.3, 1.0, 0.3, 0.3, 1.0)
phi = np.empty((Y.shape[0], k), dtype=np.float32)


def time_port(threads):
    st = O.State(nU, nI, hy, 123)
    O.cavi_iteration(st, hy, Yc, iuc, iic, phi, 0, threads)
    best = np.inf
    for _ in range(iters):
        t0 = time.time()
        O.cavi_iteration(st, hy, Yc, iuc, iic, phi, 0, threads)
        best = min(best, time.time() - t0)
    return best


def time_ref(threads):
    def fit(n):
        m = HPF(k=k, maxiter=n, random_seed=123, ncores=threads, reindex=False, verbose=False, stop_crit="maxiter",
                check_every=None, allow_inconsistent_math=False, use_float=True)
        t = time.time()
        with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m.fit(df.copy())
        return time.time() - t
    # per-iteration time = slope between two run lengths (removes input processing + init)
    return (fit(1 + iters) - fit(1)) / iters


# the build container is a shared machine: alternate the two and keep the best of each
for threads in (1, cores):
    tp, tr = np.inf, np.inf
    for _ in range(3 if threads > 1 else 1):
        tp = min(tp, time_port(threads))
        tr = min(tr, time_ref(threads))
    print("threads=%d: reference %.3f s/iter, oracle port %.3f s/iter, port/reference = %.2f"
          % (threads, tr, tp, tp / tr))
