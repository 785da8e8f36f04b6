#!/usr/bin/env python3
"""A partial_fit batch's upload two ways, 24 times each: copied into a page-locked buffer of our own + asynchronous DMAs, against
plain pageable `tensor.to(device)` -- per call the host copy, the .to() calls and the synchronisation apart
(profiles/r06_partial_fit_upload.txt: why svi.partial_fit_device uploads pageable arrays as they are)."""
import time, numpy as np, torch
dev = torch.device("cuda", 0)
n = 3_100_000
rng = np.random.default_rng(0)
srcs = [(rng.integers(0, 1 << 20, n).astype(np.int64), rng.integers(0, 1 << 18, n).astype(np.int64), rng.random(n).astype(np.float32)) for _ in range(6)]
buf = torch.empty(80 << 20, dtype=torch.uint8, pin_memory=True)
torch.cuda.synchronize()
def run(mode):
    out = []
    for it in range(24):
        a = srcs[it % 6]
        t0 = time.perf_counter()
        off = 0; outs = []; tc = 0.0; tt = 0.0
        for x in a:
            t = torch.from_numpy(x)
            if mode == "pinned":
                st = buf[off: off + x.nbytes].view(t.dtype)
                t1 = time.perf_counter(); st.copy_(t); t2 = time.perf_counter()
                outs.append(st.to(dev, non_blocking=True)); t3 = time.perf_counter()
            else:
                t1 = time.perf_counter(); t2 = t1
                outs.append(t.to(dev)); t3 = time.perf_counter()
            tc += t2 - t1; tt += t3 - t2
            off += -(-x.nbytes // 256) * 256
        ts0 = time.perf_counter()
        torch.cuda.synchronize()
        ts1 = time.perf_counter()
        s = sum(int(o[:10].sum()) for o in outs)   # touch
        torch.cuda.synchronize()
        out.append((1e3 * (ts1 - t0), 1e3 * tc, 1e3 * tt, 1e3 * (ts1 - ts0)))
        del outs
    print(mode, "total | host copy | .to() | sync  [ms]")
    for r in out:
        print("   %7.2f | %6.2f | %6.2f | %6.2f" % r)
run("pinned")
run("pageable")
