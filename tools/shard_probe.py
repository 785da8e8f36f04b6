#!/usr/bin/env python3
"""Compute-side cost of ONE rank of an N-way user-sharded C3 run, measured on a single GPU: shard r of N
(cavi.shard_users) is run through the sharded code path inside a one-rank process group, so the all-reduces
are local no-ops and what is timed is this rank's kernels + host launch work.  3.63 ms / that time bounds the
speed-up an N-GPU run can reach before any communication cost.

    HPF_FORCE_SHARDED=1 python tools/shard_probe.py [N ...]
"""
import os
import sys
import time

import numpy as np
import torch

os.environ["HPF_FORCE_SHARDED"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from hpfrec_amd import cavi  # noqa: E402
from hpfrec_amd import cython_loops_float as backend  # noqa: E402
import torch.distributed as dist  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29561")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)


class _Done:
    def wait(self):
        pass


class _EvWork:
    def __init__(self, e):
        self.e = e

    def wait(self):
        torch.cuda.current_stream().wait_event(self.e)


class EmulatedRank:
    """Stands in for torch.distributed inside cavi: reports (rank, world) of the emulated job; every collective is a
    real call on the one-rank RCCL group (so the stream hand-overs are paid) plus the local data movement that
    leaves this rank's buffers in the right shape.  Timing only: the numbers in the tables are not a real fit."""

    def __init__(self, rank, world):
        self.rank, self.world = rank, world
        self.tiny = torch.zeros(64, device=dev)
        self.side = torch.cuda.Stream()

    def get_world_size(self):
        return self.world

    def get_rank(self):
        return self.rank

    def _call(self, t, async_op):
        mode = os.environ.get("PROBE_NO_COLLECTIVES", "0")
        if mode == "1":     # pure kernel pipeline: no stream hand-overs at all
            return _Done()
        if mode == "sync":  # every collective synchronous (async_op=False)
            dist.all_reduce(t)
            return _Done()
        if mode == "events":   # only what a hand-over needs: record -> side stream waits, records -> we wait
            e0, e1 = torch.cuda.Event(), torch.cuda.Event()
            e0.record()
            self.side.wait_event(e0)
            e1.record(self.side)
            return _EvWork(e1) if async_op else torch.cuda.current_stream().wait_event(e1) or _Done()
        return dist.all_reduce(t, async_op=async_op) or _Done()

    #: HPF_NATIVE_SHARD=1 (the library default): the whole iteration issued by hpf_hip_shard_iterate in its dry-run
    #: form -- this rank alone.  HPF_SCHEDULE=direct (the default): the exchange region is connected to itself, the pulls
    #: read local memory (every rank's slice is read, the own one counted); the RCCL-shaped schedules: every collective =
    #: the local copy of the rank's slice + a one-element call on a one-rank RCCL communicator
    native_dry_run = True
    #: PROBE_BUSBW (GB/s) > 0: every emulated collective also holds its stream for latency + bytes*(N-1)/N / busbw (one
    #: sleeping wavefront): what the iteration would take if the links delivered that, with the schedule's real dependencies
    native_dry_run_busbw = float(os.environ.get("PROBE_BUSBW", "0"))
    native_dry_run_latency_us = float(os.environ.get("PROBE_LAT_US", "20"))
    # > 0: a bulk collective's stand-in is that many 256-thread workgroups with 128 VGPRs and 64 KB of LDS each
    native_dry_run_footprint_blocks = int(os.environ.get("PROBE_FAT_BLOCKS", "0"))

    def direct_comm(self, device, raw=True):
        """The one-rank RCCL communicator the dry run of an RCCL-shaped plan pays its launches on."""
        if not hasattr(EmulatedRank, "_comm"):
            from hpfrec_amd import rccl
            EmulatedRank._comm = rccl.DirectComm(device)
        return EmulatedRank._comm

    def all_reduce(self, t, op=None, async_op=False):
        return self._call(t, async_op)

    def reduce_scatter_tensor(self, out, inp, async_op=False):
        m = out.shape[0]
        out.copy_(inp[self.rank * m: (self.rank + 1) * m])
        return self._call(self.tiny, async_op)

    def all_to_all_single(self, out, inp, async_op=False):
        out.copy_(inp)
        return self._call(self.tiny, async_op)

    def all_gather_into_tensor(self, out, inp, async_op=False):
        m = inp.shape[0]
        out[self.rank * m: (self.rank + 1) * m].copy_(inp)
        return self._call(self.tiny, async_op)


nU, nI, nnz_t, k, _ = bench.WORKLOADS[os.environ.get("PROBE_WORKLOAD", "c3")]
iu, ii, y = bench.synth_on_device(nU, nI, nnz_t, dev)
Theta = np.empty((nU, k), np.float32)
Beta = np.empty((nI, k), np.float32)
init = backend.initialize_parameters(Theta, Beta, 123, 0.3, 0.3, 1.0, 0.3, 0.3, 1.0)
hy = cavi.Hyper(k, 0.3, 0.3, 1.0, 0.3, 0.3, 1.0)
for world in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    for r in ([int(x) for x in os.environ['PROBE_RANKS'].split(',')] if os.environ.get('PROBE_RANKS') else sorted({0, world - 1})):
        lu, li, ly, (u0, u1) = cavi.shard_users(iu, ii, y, nU, r, world)
        emu = EmulatedRank(r, world)
        cavi._dist = lambda: emu
        ops = bench.TimedOps(dev)
        m = cavi.FullBatchCavi(ops, dev, lu, li, ly, u1 - u0, nI, hy)
        s = slice(u0, u1)
        m.load_state(init[0][s], init[1][s], init[2], init[3], init[4][s], init[5], Theta[s], Beta)
        for store in (True, False):
            for _ in range(3):
                m.iterate(store)
            torch.cuda.synchronize()
            ops.events, ops.recording = {}, store and os.environ.get('PROBE_EVENTS', '1') == '1'
            t0 = time.perf_counter()
            steps = 30
            for _ in range(steps):
                m.iterate(store)
            t_issue = (time.perf_counter() - t0) / steps * 1e3
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps * 1e3
            ops.recording = False
            if store:
                ks = {n: round(v["total_ms"] / steps, 3) for n, v in ops.summary().items()}
                if EmulatedRank.native_dry_run_busbw > 0 and m._plan is not None:
                    print("(collectives hold their stream as %.0f GB/s of bus bandwidth + %.0f us would)"
                          % (EmulatedRank.native_dry_run_busbw, EmulatedRank.native_dry_run_latency_us))
                print("world %d rank %d [%s, %s, %d item ranges]: %d users, %d nnz: %.3f ms/iteration (all tables stored; host "
                      "issue time %.3f ms); kernels ms/iter %s"
                      % (world, r, "native C issue" if m._plan is not None else "python issue", m.schedule,
                         len(m.item_chunks), u1 - u0, m.nnz, dt, t_issue, ks), flush=True)
            else:
                print("world %d rank %d: %.3f ms/iteration without the output-table stores (host issue time %.3f ms)"
                      % (world, r, dt, t_issue), flush=True)
        del m, ops, lu, li, ly
        torch.cuda.empty_cache()
dist.destroy_process_group()
