#!/usr/bin/env python3
"""BASELINE config C5 read literally: HPF.partial_fit on one GPU, users_per_batch = items_per_batch = 65536, k = 200, on
the C3-shaped matrix (1M x 380k, 48M nonzeros).  One pass over the users (16 calls of 65,536 users with ALL their
interactions, as the method's contract demands) and one over the items (6 calls), each call timed device-synchronised,
with the host's share (the class's own pandas / numpy statements + uploads) separated from the step's kernels."""
import os
import sys
import time
import warnings

import numpy as np
import pandas as pd
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from hpfrec_amd import HPF  # noqa: E402

nU, nI, nnz_t, _, _ = bench.WORKLOADS["c3"]
k, per = 200, 65536
dev = torch.device("cuda", 0)
iu, ii, y = bench.synth_on_device(nU, nI, nnz_t, dev)
order = torch.argsort(iu, stable=True)
IU, II, Y = iu[order].cpu().numpy(), ii[order].cpu().numpy(), y[order].cpu().numpy()
order_i = torch.argsort(ii, stable=True)
IU2, II2, Y2 = iu[order_i].cpu().numpy(), ii[order_i].cpu().numpy(), y[order_i].cpu().numpy()
del iu, ii, y, order, order_i
torch.cuda.empty_cache()
ptr_u = np.searchsorted(IU, np.arange(0, nU + per, per))
ptr_i = np.searchsorted(II2, np.arange(0, nI + per, per))


def frames(U, I, C, ptr):
    return [pd.DataFrame({"UserId": U[a:b], "ItemId": I[a:b], "Count": C[a:b]}) for a, b in zip(ptr[:-1], ptr[1:]) if b > a]


user_batches, item_batches = frames(IU, II, Y, ptr_u), frames(IU2, II2, Y2, ptr_i)
m = HPF(k=k, reindex=False, keep_data=False, random_seed=7, verbose=False)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    m.partial_fit(user_batches[0], nusers=nU, nitems=nI)        # uploads the 2.2 GB of initial state once
    m.partial_fit(item_batches[0], batch_type="items")
    torch.cuda.synchronize()
    for name, batches, kind in (("user", user_batches, "users"), ("item", item_batches, "items")):
        t_call, t_dev = [], []
        for b in batches:
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            m.partial_fit(b, batch_type=kind)
            e1.record()
            torch.cuda.synchronize()
            t_call.append(time.perf_counter() - t0)
            t_dev.append(e0.elapsed_time(e1) * 1e-3)
        n = np.array([b.shape[0] for b in batches])
        print("%s batches: %d calls of <= %d rows, %.2fM triplets each on average: %.1f ms per call (median; min %.1f, max %.1f); "
              "between the first and the last device operation of a call: %.1f ms"
              % (name, len(batches), per, n.mean() / 1e6, 1e3 * np.median(t_call), 1e3 * min(t_call), 1e3 * max(t_call),
                 1e3 * np.median(t_dev)))
if os.environ.get("HPF_TIMING") == "1":       # device-synchronised phases, per call, next to the call's triplet count
    from hpfrec_amd import svi
    svi.PF_TIMINGS.clear()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for kind, batches in (("users", user_batches), ("items", item_batches)):
            for b in batches:
                m.partial_fit(b, batch_type=kind)
    ph = svi.PF_TIMINGS
    sizes = [b.shape[0] for b in user_batches + item_batches]
    print("phases per call [ms] (device-synchronised after each; triplets | " + " | ".join(ph) + "):")
    for j, nn in enumerate(sizes):
        print("  %s %9d | " % ("user" if j < len(user_batches) else "item", nn) + " | ".join("%6.2f" % (1e3 * ph[p_][j]) for p_ in ph))
if os.environ.get("PF_PROFILE") == "1":       # where the host spends an item call (cProfile over a second pass of them)
    import cProfile
    import pstats
    pr = cProfile.Profile()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        pr.enable()
        for b in item_batches:
            m.partial_fit(b, batch_type="items")
        torch.cuda.synchronize()
        pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
st = m._state
print("state traffic after the first calls: h2d %.2f GB, d2h %.2f GB" % (st.stats["h2d_bytes"] / 1e9, st.stats["d2h_bytes"] / 1e9))
th = st.rows("Theta", [0, nU - 1])
assert np.isfinite(th).all() and (th > 0).all()
