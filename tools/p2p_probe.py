#!/usr/bin/env python3
"""N processes sharing cuda:0 (or, P2P_PROBE_DEVICE_PER_RANK=1, one GPU each) map one another's exchange regions (hpfrec_amd/p2p.py) and run the primitives of the
direct exchange: flag signal / wait, pulls of a peer's buffer, the k-float all-reduce by granules.  Answers, on a one-GPU
box, what the direct exchange needs from the platform: hipIpc of coarse- and fine-grained memory between processes, kernels
of several processes running at the same time (a waiting kernel must not starve the kernel it waits for), and what a
flag round trip and a vector all-reduce cost.

    python tools/p2p_probe.py [world ...]          (default: 2 3 8)
"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker():
    import torch
    import torch.distributed as dist
    from hpfrec_amd import p2p
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    per_rank = os.environ.get("P2P_PROBE_DEVICE_PER_RANK") == "1"      # one GPU per process: the links carry everything
    idx = rank if per_rank else 0
    torch.cuda.set_device(idx)
    dev = torch.device("cuda", idx)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ld = 64
    n = 1 << 20                     # floats per rank in the data buffer (4 MB)
    reg = p2p.PeerRegion(dev, n * 4, ld, dist=dist, rank=rank, world=world, timeout_ms=15000)
    mine = reg.tensor(0, (n,))
    # 1. visibility of a peer's buffer after its flag
    for rep in range(3):
        e = reg.next_epoch()
        mine.fill_(float(rank + 1) + 0.25 * rep)
        reg.signal(p2p.FLAG_USER, e)
        got = torch.empty((world, n), device=dev)
        for p in range(world):
            reg.pull(got[p], p, 0, kind=p2p.FLAG_USER, epoch=e)
        torch.cuda.synchronize()
        want = torch.arange(1, world + 1, device=dev, dtype=torch.float32) + 0.25 * rep
        assert torch.equal(got, want[:, None].expand(world, n)), (rank, rep, got[:, :2])
        dist.barrier()              # (nobody refills its buffer while a peer is still reading it)
    reg.status()
    # 2. flag round trip between ranks 0 and 1
    rounds = 200
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if rank < 2 and world >= 2:
        for _ in range(rounds):
            e = reg.next_epoch()
            if rank == 0:
                reg.signal(p2p.FLAG_USER, e)
                reg.wait(p2p.FLAG_USER + 1, e, 1 << 1)
            else:
                reg.wait(p2p.FLAG_USER, e, 1 << 0)
                reg.signal(p2p.FLAG_USER + 1, e)
        torch.cuda.synchronize()
    else:
        for _ in range(rounds):
            reg.next_epoch()
    rt = (time.perf_counter() - t0) / rounds * 1e6
    reg.status()
    dist.barrier()
    # 3. the k-float all-reduce
    vec = torch.zeros(ld, device=dev)
    base = torch.arange(ld, device=dev, dtype=torch.float32) * 1e-3
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(rounds):
        e = reg.next_epoch()
        vec.copy_(base + (rank + 1) * (1 + it % 3))
        reg.allreduce_vec(it & 1, e, vec)
    torch.cuda.synchronize()
    ar = (time.perf_counter() - t0) / rounds * 1e6
    it = rounds - 1
    want = sum((base + (r + 1) * (1 + it % 3)) for r in range(world))     # rank order, like the kernel
    assert torch.equal(vec, want), (rank, vec[:4], want[:4])
    allv = [torch.empty_like(vec).cpu() for _ in range(world)]
    dist.all_gather(allv, vec.cpu())
    assert all(torch.equal(a, allv[0]) for a in allv)
    reg.status()
    # 4. a wait that cannot be satisfied comes back with HPF_ETIMEOUT instead of hanging
    if world >= 2:
        reg.L.hpf_hip_p2p_region_set_timeout(reg.handle, 50.0)
        t0 = time.perf_counter()
        reg.wait(p2p.FLAG_USER + 1, 0x7FFFFFF0, 1 << ((rank + 1) % world))
        try:
            reg.status()
            timed_out = False
        except p2p.P2PError:
            timed_out = True
        dt = time.perf_counter() - t0
        assert timed_out and dt < 5.0, (timed_out, dt)
    dist.barrier()
    if rank == 0:
        print("world %d (%s): pulls of every peer's buffer after its flag OK; flag round trip 0<->1 %.1f us; "
              "all-reduce of %d floats by granules %.1f us per call (incl. one copy kernel); time-out path OK"
              % (world, "one GPU per rank" if per_rank else "ranks share cuda:0", rt, ld, ar), flush=True)
    # 5. what the links deliver: per-peer pull GB/s (hpfrec_amd.p2p.link_probe, the block bench.py carries)
    lp = p2p.link_probe(dev, dist, rank, world)
    if rank == 0:
        import json
        print("link_probe " + json.dumps({k_: v for k_, v in lp.items() if k_ != "note"}), flush=True)
    dist.barrier()
    del mine
    reg.close()
    dist.destroy_process_group()


def main():
    worlds = [int(a) for a in sys.argv[1:]] or [2, 3, 8]
    port = 29600
    for w in worlds:
        port += 1
        procs = []
        for r in range(w):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(w), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       P2P_PROBE_WORKER="1")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=env))
        rc = 0
        t0 = time.time()
        for p in procs:
            try:
                rc |= p.wait(timeout=max(1.0, 240 - (time.time() - t0)))
            except subprocess.TimeoutExpired:
                p.kill()
                rc |= 99
        print("world %d: exit %d" % (w, rc), flush=True)
        if rc:
            sys.exit(1)


if __name__ == "__main__":
    if os.environ.get("P2P_PROBE_WORKER") == "1":
        worker()
    else:
        main()
