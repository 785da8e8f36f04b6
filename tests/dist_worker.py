"""Worker for tests/test_dist_gloo.py: one rank of a world_size-N gloo job running the sharded
driver on the numpy stand-in ops -- or, with device="cuda", on the real HIP kernels (all ranks share cuda:0;
gloo moves the CUDA tensors through the host: a correctness run of the N>1 path on a one-GPU box) -- or, with
device="cuda-per-rank" (tests/test_multi_gpu.py, boxes with >= N GPUs), one process per GPU on the nccl backend
(RCCL): rank r on cuda:r, the links between the GPUs carrying the exchange."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.dirname(HERE), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def run(rank, world, port, out_dir, k, its, case, device="cpu"):
    try:
        _run(rank, world, port, out_dir, k, its, case, device)
    except BaseException:   # noqa: BLE001  (leave the traceback where the parent test can show it)
        import traceback
        with open(os.path.join(out_dir, "rank%d.err" % rank), "w") as f:
            f.write(traceback.format_exc())
        raise


def gloo_collective(dist, model):
    """hpf_collective_fn for a cavi.FullBatchCavi: the three collectives of the sharded iteration carried by gloo (which
    stages CUDA tensors through the host), synchronously.  The C side hands over raw device pointers; they are mapped
    back onto the model's exchange buffers."""
    import torch
    bufs = [b for b in (model.acc_i, model.acc_own_all, model.e_own_all, model.eB, model.csT, model.csB,
                        getattr(model, "ag_recv_all", None)) if b is not None]
    calls = model.__dict__.setdefault("_native_collective_calls", [0, 0, 0])

    def view(ptr, count):
        for b in bufs:
            base = b.data_ptr()
            if base <= ptr < base + 4 * b.numel():
                off = (ptr - base) // 4
                assert off + count <= b.numel()
                return b.view(-1)[off: off + count]
        raise KeyError("pointer outside the exchange buffers")

    def coll(ctx, op, send, recv, count, stream):
        try:
            W = dist.get_world_size()
            torch.cuda.synchronize()
            if op == 0:
                dist.all_reduce(view(recv, count))
            elif op == 1:
                dist.reduce_scatter_tensor(view(recv, count), view(send, count * W))
            elif op == 2:
                dist.all_gather_into_tensor(view(recv, count * W), view(send, count))
            else:
                return -1
            torch.cuda.synchronize()
            calls[op] += 1
            return 0
        except BaseException:   # noqa: BLE001  (no exception may cross the C frame)
            import traceback
            traceback.print_exc()
            return -7
    return coll


def _run(rank, world, port, out_dir, k, its, case, device="cpu"):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if device == "cuda-per-rank":
        import torch
        assert torch.cuda.device_count() >= world, "one GPU per rank"
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import cpu_ops
    import datagen
    from hpfrec_amd import cython_loops_float as be
    if device == "cpu":
        be.HipOps = lambda device=None: cpu_ops.CpuOps()   # numpy stand-in for the kernels (host logic test)
    elif device == "cuda-per-rank":
        pass                  # (the library's own defaults: RCCL job with more than one rank -> first iterations checked)
    else:
        import torch
        torch.cuda.set_device(0)
        if os.environ.get("HPF_TEST_NATIVE_GLOO") == "1":
            # the C-issued iteration (hpf_hip_shard_iterate) with gloo standing in for RCCL through its callback hook
            dist.native_collective = lambda model: gloo_collective(dist, model)
    seed = 123
    if case.endswith("-entropy"):       # random_seed <= 0: OS entropy (PXI:127) -- every rank must end up with rank 0's
        case, seed = case[: -len("-entropy")], 0
    if case == "few":         # 3 users x 40 items over more ranks than users: some ranks hold NO user at all
        rs = np.random.RandomState(5)
        nU, nI = 3, 40
        iu = np.repeat(np.arange(3), (30, 3, 12)).astype(np.uint64)
        ii = np.concatenate([rs.choice(40, n, replace=False) for n in (30, 3, 12)]).astype(np.uint64)
        Y = (rs.gamma(1, 1, size=iu.shape[0]) + 1).astype(np.int32).astype(np.float32)
    elif case == "large":     # the large golden's matrix (tests/golden/large_full.npz: the REAL reference at 200k x 50k)
        iu, ii, Y, nU, nI = datagen.large_counts()
    elif case == "c4small":     # BASELINE C4's shape of problem at 2M nonzeros (tests/test_full_size.py)
        nU, nI = 100_000, 30_000
        iu, ii, Y = datagen.synthetic_hpf_shaped(nU, nI, 2_000_000, seed=4)
    else:
        if case == "c1":
            df, nU, nI = datagen.readme_counts()
        else:
            df, nU, nI = datagen.mid_counts(nusers=600, nitems=400, nobs=20000)
        Y, iu, ii = datagen.triplets(df)
    Theta = np.empty((nU, k), np.float32)
    Beta = np.empty((nI, k), np.float32)
    # (HPF_TEST_CHECK_EVERY: llk checks in the middle of the fit -- the sharded schedules are joined between iterations;
    # the model does not depend on it)
    check_every = int(os.environ.get("HPF_TEST_CHECK_EVERY", its))
    i, temp, llk = be.fit_hpf(0.3, 0.3, 1.0, 0.3, 0.3, 1.0, Y, iu, ii, Theta, Beta, its, "maxiter", check_every, 1e-3, 0, 0,
                              None, 0, np.zeros(1, np.uint64), "", seed, 1, 1, 0, 0, np.empty(0, np.float32),
                              np.empty(0, np.uint64), np.empty(0, np.uint64), 0, 1, 0)
    from hpfrec_amd import shard
    native = int(shard.NATIVE_PLANS_CREATED[0])
    if case == "large":       # compact: the golden's sub-sampled rows and float64 column sums
        arrs = dict(zip(("Theta", "Beta", "Gamma_shp", "Gamma_rte", "Lambda_shp", "Lambda_rte", "k_rte", "t_rte"),
                        (Theta, Beta) + tuple(temp)))
        out = {}
        for n, v in arrs.items():
            out[n + "_rows"] = v[::(400 if v.shape[0] == nU else 100)].copy()
            out[n + "_colsum64"] = v.astype(np.float64).sum(axis=0)
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), llk=np.float64(llk), niter=i, native_plans=native,
                 schedule=str(shard.LAST_SCHEDULE[0]), checked_iterations=max(shard._PASSED.values(), default=0),
                 failed_schedules=",".join(sorted(k_[0] for k_ in shard._FAILED)), **out)
        dist.destroy_process_group()
        return
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), Theta=Theta, Beta=Beta, Gamma_shp=temp[0], Gamma_rte=temp[1],
             Lambda_shp=temp[2], Lambda_rte=temp[3], k_rte=temp[4], t_rte=temp[5], llk=np.float64(llk), niter=i,
             native_plans=native, schedule=str(shard.LAST_SCHEDULE[0]),
             checked_iterations=max(shard._PASSED.values(), default=0))
    dist.destroy_process_group()
