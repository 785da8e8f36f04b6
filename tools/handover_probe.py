#!/usr/bin/env python3
"""What does a stream hand-over cost on the compute stream?  Two ~100 us kernels per iteration with, in between:
nothing / an event record / record + wait on a side stream / a (one-rank) RCCL all-reduce issued asynchronously."""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29571")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
a = torch.randn(64 << 20, device=dev)      # 256 MB: a*1.0001 streams it in ~0.1 ms
small = torch.zeros(64, device=dev)
side = torch.cuda.Stream()
ev = torch.cuda.Event(enable_timing=False)


def between(kind):
    if kind == "record":
        ev.record()
    elif kind == "record+wait":
        ev.record()
        side.wait_event(ev)
    elif kind == "side kernel + join":
        ev.record()
        side.wait_event(ev)
        with torch.cuda.stream(side):
            small.add_(1.0)
        torch.cuda.current_stream().wait_stream(side)
    elif kind == "async all_reduce":
        return dist.all_reduce(small, async_op=True)
    elif kind == "sync all_reduce":
        dist.all_reduce(small)
    return None


for kind in ("nothing", "record", "record+wait", "side kernel + join", "async all_reduce", "sync all_reduce"):
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 200
        for _ in range(n):
            a.mul_(1.0001)
            w = between(kind)
            a.mul_(0.9999)
            if w is not None:
                w.wait()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n * 1e6
    print("%-22s %.1f us per iteration" % (kind, dt))
dist.destroy_process_group()
