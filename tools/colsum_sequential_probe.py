#!/usr/bin/env python3
"""hpf_hip_colsum_sequential_f32 (numpy's own float32 row-after-row column sums, HPF_COLSUM_ORDER=reference) timed over table
sizes and checked bit for bit against numpy (profiles/r06_colsum_sequential.txt)."""
import time, numpy as np, torch, sys, os
sys.path.insert(0, os.getcwd())
from hpfrec_amd.ops_hip import HipOps
ops = HipOps(torch.device("cuda", 0))
for n, ld in ((1_000_000, 64), (380_000, 64), (1_000_000, 256), (200_000, 64)):
    tab = torch.rand((n, ld), device="cuda") + 0.1
    out = torch.zeros(ld, device="cuda")
    ops.colsum_sequential(tab, n, ld, out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ops.colsum_sequential(tab, n, ld, out)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    ref = tab[:, :3].cpu().numpy()
    s = np.zeros(3, np.float32)
    want = ref.sum(axis=0)
    print("colsum_sequential %8d x %4d: %.3f ms (%.2f ns per row); equals numpy: %s" % (n, ld, ms, ms * 1e6 / n, np.array_equal(out[:3].cpu().numpy(), want)))
