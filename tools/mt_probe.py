#!/usr/bin/env python3
"""Time of the initial draw's MT19937 stream (hpf_hip_mt19937_words): the jump-ahead path against the single-workgroup
walk, for the word counts of C3 (k=50), C4 (k=100) and C5 (k=200): 2 * (1M + 380k) * k words."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hpfrec_amd.ops_hip import HipOps  # noqa: E402
from hpfrec_amd import cavi, _lib  # noqa: E402

ops = HipOps()
dev = ops.device
for name, k in (("C3", 50), ("C4", 100), ("C5", 200)):
    n = 2 * (1_000_000 + 380_000) * k
    raw = torch.empty(n, dtype=torch.int32, device=dev)
    for label, scratch_on in (("jump-ahead, 512-1024 workgroups", True), ("one workgroup", False)):
        ts = []
        for rep in range(3):
            state = cavi.mt19937_state_words(123).to(dev)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if scratch_on:
                ops.mt19937_words(state, raw)
            else:
                _lib.check(ops.L.hpf_hip_mt19937_words(state.data_ptr(), raw.data_ptr(), n, None, ops._stream()), "mt")
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        print("%s: %d words, %s: %.2f ms (runs: %s), checksum %d" % (
            name, n, label, min(ts), " ".join("%.2f" % t for t in ts), int(raw.to(torch.int64).sum().item())), flush=True)
    del raw
