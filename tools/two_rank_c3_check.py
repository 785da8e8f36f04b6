#!/usr/bin/env python3
"""Full-size check of the N>1 path on ONE GPU: W gloo ranks share cuda:0 (gloo stages the exchange through the
host: slow, but it is the real sharded driver on the real kernels at C3 size -- split popular rows, 64-bit offsets,
pad rows), a few iterations, every rank's tables against the single-process run.

    python tools/two_rank_c3_check.py [world=2] [HPF_SCHEDULE=direct|gather-early|...] [iterations=3] [workload=c3]
    HPF_TEST_NATIVE_GLOO=1 python tools/two_rank_c3_check.py 8 gather-early       (C-issued, gloo behind the callback)
"""
import os
import sys
import time

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def state(model, u0, u1):
    model.flush_items()
    k = model.k
    return {"Theta": model.Theta[:, :k].cpu().numpy(), "Beta": model.Beta[: model.nI, :k].cpu().numpy(),
            "Lambda_shp": model.Lambda_shp[: model.nI, :k].cpu().numpy(), "t_rte": model.t_rte[: model.nI].cpu().numpy(),
            "k_rte": model.k_rte.cpu().numpy(), "llk": model.llk_terms(False), "rows": (u0, u1)}


def fit(rank, world, wl, its):
    import bench
    from hpfrec_amd import cavi
    from hpfrec_amd import cython_loops_float as backend
    from hpfrec_amd.ops_hip import HipOps
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    nU, nI, nnz_t, k, _ = bench.WORKLOADS[wl]
    iu, ii, y = bench.synth_on_device(nU, nI, nnz_t, dev)
    lu, li, ly, (u0, u1) = cavi.shard_users(iu, ii, y, nU, rank, world)
    del iu, ii, y
    hy = cavi.Hyper(k, 0.3, 0.3, 1.0, 0.3, 0.3, 1.0)
    m = cavi.FullBatchCavi(HipOps(dev), dev, lu, li, ly, u1 - u0, nI, hy)
    Theta = np.empty((nU, k), np.float32)
    Beta = np.empty((nI, k), np.float32)
    init = backend.initialize_parameters(Theta, Beta, 123, 0.3, 0.3, 1.0, 0.3, 0.3, 1.0)
    s = slice(u0, u1)
    m.load_state(init[0][s], init[1][s], init[2], init[3], init[4][s], init[5], Theta[s], Beta)
    for _ in range(its):
        m.iterate(True)
    return state(m, u0, u1)


def worker(rank, world, port, wl, its, mode, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HPF_SCHEDULE=mode)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if os.environ.get("HPF_TEST_NATIVE_GLOO") == "1":
        # the C-issued iteration (hpf_hip_shard_iterate) with gloo standing
        # in for RCCL through the collective callback
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from dist_worker import gloo_collective
        dist.native_collective = lambda model: gloo_collective(dist, model)
    st = fit(rank, world, wl, its)
    np.savez(os.path.join(out, "rank%d.npz" % rank), **{k: v for k, v in st.items()})
    dist.destroy_process_group()


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    mode = sys.argv[2] if len(sys.argv) > 2 else "direct"
    its = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    wl = sys.argv[4] if len(sys.argv) > 4 else "c3"
    out = "/tmp/two_rank_check"
    os.makedirs(out, exist_ok=True)
    t0 = time.time()
    ref = fit(0, 1, wl, its)
    print("single process: %.1f s" % (time.time() - t0), flush=True)
    t0 = time.time()
    mp.spawn(worker, args=(world, 29641, wl, its, mode, out), nprocs=world, join=True)
    print("%d ranks sharing the GPU (HPF_SCHEDULE=%s): %.1f s" % (world, mode, time.time() - t0), flush=True)
    worst = {}
    for r in range(world):
        d = np.load(os.path.join(out, "rank%d.npz" % r))
        u0, u1 = [int(v) for v in d["rows"]]
        for n in ("Theta", "k_rte"):
            worst[n] = max(worst.get(n, 0.0), float(np.max(np.abs(d[n] - ref[n][u0:u1]) / np.abs(ref[n][u0:u1]))))
        for n in ("Beta", "Lambda_shp", "t_rte"):
            worst[n] = max(worst.get(n, 0.0), float(np.max(np.abs(d[n] - ref[n]) / np.abs(ref[n]))))
        worst["llk"] = max(worst.get("llk", 0.0), float(np.max(np.abs(d["llk"] / ref["llk"] - 1))))
    print("worst relative deviation from the single-process run after %d iterations: %s"
          % (its, {k: "%.1e" % v for k, v in worst.items()}))
    assert all(v < 2e-5 for v in worst.values()), worst
    print("TWO_RANK_CHECK_OK")
