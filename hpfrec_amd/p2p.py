"""Peer-mapped exchange regions (include/hpf_hip.h, "Multi-GPU, direct exchange"; hpfrec_amd/csrc/hpf_p2p.hip).

SURVEY.md section 8(e) wants the item statistics of the user-sharded iteration to cross xGMI directly -- every GPU
talking to its seven peers at once -- instead of through a ring collective.  A `PeerRegion` is this rank's share of
that: one device buffer (the packed item accumulators and the finished [numerators | base rate] rows, layout chosen by
cavi.FullBatchCavi) plus a small control block of flags, both exported with hipIpcGetMemHandle and mapped by every other
rank of the node.  The 128 handle bytes per rank travel through torch.distributed (all_gather_object: the control
plane), exactly like the ncclUniqueId of rccl.DirectComm.  The reference has no counterpart (single-node OpenMP,
/root/reference/hpfrec/cython_loops.pxi:4).
"""
import atexit
import ctypes
import os

import numpy as np
import torch

from . import _lib

HANDLE_BYTES = 64
MAX_RANKS = 16
FLAG_USER = 30


class P2PError(RuntimeError):
    pass


def _check(rc, what):
    if rc != 0:
        raise P2PError("%s failed with code %d%s" % (what, rc, " (a peer's flag never arrived)" if rc == -4 else ""))


class _DevView:
    """A raw device pointer as something torch.as_tensor understands (__cuda_array_interface__)."""

    def __init__(self, ptr, shape, keep):
        self.__cuda_array_interface__ = {"shape": tuple(int(s) for s in shape), "typestr": "<f4", "data": (int(ptr), False),
                                         "version": 2, "strides": None}
        self._keep = keep


class PeerRegion:
    """This rank's exchange region, connected to its peers' (or, `dist` None / `local=True`, standing alone: every peer
    mapped to the local memory, nothing ever waited for -- probes and the bench's compute-only twin)."""

    def __init__(self, device, data_bytes, ld, dist=None, rank=0, world=1, local=False, timeout_ms=None):
        self.L = _lib.lib()
        self.device = torch.device(device)
        self.rank, self.world, self.ld = int(rank), int(world), int(ld)
        if self.world > MAX_RANKS:
            raise P2PError("at most %d ranks" % MAX_RANKS)
        self.data_bytes = int(data_bytes)
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            alone = local or dist is None or self.world == 1
            if os.environ.get("HPF_TEST_P2P_FAIL_CREATE") == str(self.rank) and not alone:    # (tests: one rank's failure)
                rc = -2
            else:
                rc = self.L.hpf_hip_p2p_region_create(self.world, self.rank, self.ld, self.data_bytes, ctypes.byref(h))
            if alone:
                _check(rc, "hpf_hip_p2p_region_create")
            self.handle = h if rc == 0 else None
            if alone:
                _check(self.L.hpf_hip_p2p_region_connect(self.handle, None), "hpf_hip_p2p_region_connect")
                self.local = True
            else:
                # every rank takes part in the exchange whatever happened locally (an allocation that failed on ONE rank
                # must not leave the others waiting in the all-gather), and all connect or none does
                mine = (ctypes.c_uint8 * (2 * HANDLE_BYTES))()
                if rc == 0:
                    rc = self.L.hpf_hip_p2p_region_handles(self.handle, ctypes.addressof(mine))
                got = [None] * self.world
                dist.all_gather_object(got, (rc, bytes(mine)))
                if any(r != 0 for r, _ in got):
                    self.close()
                    raise P2PError("the exchange region could not be created / exported on a rank (codes %s)"
                                   % [r for r, _ in got])
                blob = b"".join(b for _, b in got)
                buf = (ctypes.c_uint8 * len(blob)).from_buffer_copy(blob)
                rc = self.L.hpf_hip_p2p_region_connect(self.handle, ctypes.addressof(buf))
                oks = [None] * self.world
                dist.all_gather_object(oks, rc)
                if any(r != 0 for r in oks):
                    self.close()
                    raise P2PError("hipIpcOpenMemHandle failed on a rank (codes %s)" % oks)
                self.local = False
            if timeout_ms is not None:
                _check(self.L.hpf_hip_p2p_region_set_timeout(self.handle, float(timeout_ms)), "set_timeout")

    # ---- memory -----------------------------------------------------------------------------------------------------
    def data_ptr(self, peer=None):
        p = ctypes.c_void_p()
        _check(self.L.hpf_hip_p2p_region_data(self.handle, self.rank if peer is None else int(peer), ctypes.byref(p)),
               "hpf_hip_p2p_region_data")
        return p.value

    def tensor(self, offset_bytes, shape, peer=None):
        """float32 tensor over [offset, offset + prod(shape)*4) of `peer`'s data buffer as mapped here (default: the
        local buffer).  The region must outlive the tensor."""
        n = int(np.prod(shape)) * 4
        if offset_bytes < 0 or offset_bytes % 16 or offset_bytes + n > self.data_bytes:
            raise P2PError("view outside the region")
        return torch.as_tensor(_DevView(self.data_ptr(peer) + offset_bytes, shape, self), device=self.device)

    # ---- primitive stream operations (tests, probes) ------------------------------------------------------------------
    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def next_epoch(self):
        e = ctypes.c_uint32(0)
        _check(self.L.hpf_hip_p2p_region_next_epoch(self.handle, ctypes.byref(e)), "hpf_hip_p2p_region_next_epoch")
        return e.value

    def signal(self, kind, epoch):
        _check(self.L.hpf_hip_p2p_signal(self.handle, int(kind), int(epoch), self._stream()), "hpf_hip_p2p_signal")

    def wait(self, kind, epoch, src_mask=None):
        mask = (1 << self.world) - 1 if src_mask is None else int(src_mask)
        _check(self.L.hpf_hip_p2p_wait(self.handle, int(kind), int(epoch), mask, self._stream()), "hpf_hip_p2p_wait")

    def allreduce_vec(self, which, epoch, vec):
        assert vec.dtype == torch.float32 and vec.is_contiguous() and vec.numel() == self.ld
        _check(self.L.hpf_hip_p2p_allreduce_vec_f32(self.handle, int(which), int(epoch), vec.data_ptr(), self._stream()),
               "hpf_hip_p2p_allreduce_vec_f32")

    def pull(self, dst, src_rank, src_offset_bytes, kind=-1, epoch=0, grid_blocks=64):
        assert dst.dtype == torch.float32 and dst.is_contiguous()
        _check(self.L.hpf_hip_p2p_pull_f32(self.handle, int(kind), int(epoch), int(src_rank), int(src_offset_bytes),
                                           dst.data_ptr(), dst.numel(), int(grid_blocks), self._stream()),
               "hpf_hip_p2p_pull_f32")

    def status(self):
        """Synchronises the device; raises when a wait of this rank ran into the time-out."""
        err = ctypes.c_uint32(0)
        rc = self.L.hpf_hip_p2p_region_status(self.handle, ctypes.byref(err))
        if rc != 0:
            raise P2PError("direct exchange: a peer's flag never arrived (error word 0x%x, code %d)" % (err.value, rc))

    def close(self):
        """Frees the region.  No collective in here: the caller closes only after a host-side collective every rank
        entered with its device synchronised (the end of a fit: ShardedMixin.release_exchange; the first-iteration vote; the
        barrier of link_probe), so that no peer's pull can still be reading this memory."""
        if getattr(self, "handle", None) is not None and self.handle.value:
            self.L.hpf_hip_p2p_region_destroy(self.handle)
            self.handle = None

    def __del__(self):
        # A region that is merely dropped (an exception mid-fit, a model garbage-collected on ONE rank) is not freed here:
        # a peer may still have a pull of this memory queued, and reading freed memory faults ITS GPU instead of running
        # into the bounded flag time-out.  The handle is parked and freed when the process ends.
        try:
            if getattr(self, "handle", None) is not None and self.handle.value:
                if getattr(self, "local", True):
                    self.close()
                else:
                    _PARKED.append((self.L, self.handle))
                    self.handle = None
        except Exception:   # noqa: BLE001
            pass


_PARKED = []


def _free_parked():
    while _PARKED:
        L, h = _PARKED.pop()
        try:
            L.hpf_hip_p2p_region_destroy(h)
        except Exception:   # noqa: BLE001
            pass


atexit.register(_free_parked)



def link_probe(device, dist, rank, world, mbytes=64, reps=3, timeout_ms=15000.0):
    """What the links between the ranks of a job deliver to the primitives of the direct exchange, measured with the job's
    own ranks (bench.py's `collective.link_probe`; tools/p2p_probe.py): every rank maps every peer's region, then
      * pulls `mbytes` MB of peer (rank + s) % world's buffer for s = 1 .. world-1, all ranks at the same shift at the same
        time (so every link carries one pull per direction), HIP events around `reps` pulls -> GB/s per (rank, peer);
      * ranks 0 and 1 bounce a flag (signal kernel -> wait kernel, host-issued) -> microseconds per round trip;
      * the k-float granule all-reduce -> microseconds per call.
    Collective: every rank of `dist` must call it.  Returns the same dict on every rank; {"error": ...} when the regions
    could not be created / mapped (every rank learns that together: PeerRegion votes)."""
    import time
    device = torch.device(device)
    n = int(mbytes) * (1 << 20) // 4
    ld = 64
    try:
        reg = PeerRegion(device, n * 4, ld, dist=dist, rank=rank, world=world, timeout_ms=timeout_ms)
    except P2PError as exc:
        return {"error": str(exc)[:300]}
    out = {"ranks": world, "bytes_per_pull": n * 4, "shared_device": None}
    try:
        with torch.cuda.device(device):
            mine = reg.tensor(0, (n,))
            mine.fill_(float(rank + 1))
            dst = torch.empty(n, dtype=torch.float32, device=device)
            e = reg.next_epoch()
            reg.signal(FLAG_USER, e)
            pulls, ok = {}, True
            for s in range(1, world):
                peer = (rank + s) % world
                reg.pull(dst, peer, 0, kind=FLAG_USER, epoch=e, grid_blocks=512)        # warm; waits for the peer's flag
                torch.cuda.synchronize(device)
                ok = ok and bool((dst[:: max(1, n // 1024)] == float(peer + 1)).all().item())
                dist.barrier()
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
                ev[0].record()
                for _ in range(reps):
                    reg.pull(dst, peer, 0, grid_blocks=512)
                ev[1].record()
                torch.cuda.synchronize(device)
                pulls[peer] = n * 4 * reps / (ev[0].elapsed_time(ev[1]) * 1e-3) / 1e9
            reg.status()
            dist.barrier()
            # flag round trip 0 <-> 1
            rounds, rt = 100, None
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for _ in range(rounds):
                e = reg.next_epoch()
                if world >= 2 and rank == 0:
                    reg.signal(FLAG_USER, e)
                    reg.wait(FLAG_USER + 1, e, 1 << 1)
                elif world >= 2 and rank == 1:
                    reg.wait(FLAG_USER, e, 1 << 0)
                    reg.signal(FLAG_USER + 1, e)
            torch.cuda.synchronize(device)
            if rank < 2 <= world:
                rt = (time.perf_counter() - t0) / rounds * 1e6
            reg.status()
            dist.barrier()
            vec = torch.ones(ld, dtype=torch.float32, device=device)
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for it in range(rounds):
                e = reg.next_epoch()
                reg.allreduce_vec(it & 1, e, vec)
                vec.fill_(1.0)
            torch.cuda.synchronize(device)
            ar = (time.perf_counter() - t0) / rounds * 1e6
            reg.status()
            try:
                bus = torch.cuda.get_device_properties(device).pci_bus_id
            except Exception:   # noqa: BLE001
                bus = None
            got = [None] * world
            dist.all_gather_object(got, {"rank": rank, "device": str(device), "pci_bus_id": bus, "pull_GBps": pulls,
                                         "values_ok": ok, "flag_round_trip_us": rt, "vec_allreduce_us": ar})
            out["per_rank"] = got
            flat = [v for g in got for v in g["pull_GBps"].values()]
            out["pull_GBps_min"] = min(flat) if flat else None
            out["pull_GBps_median"] = float(np.median(flat)) if flat else None
            out["pull_GBps_max"] = max(flat) if flat else None
            out["flag_round_trip_us_0_1"] = got[0]["flag_round_trip_us"]
            out["vec_allreduce_us_max"] = max(g["vec_allreduce_us"] for g in got)
            out["values_ok"] = all(g["values_ok"] for g in got)
            ids = [(g["device"], g["pci_bus_id"]) for g in got]
            out["shared_device"] = len(set(ids)) < world
            out["note"] = ("pull = p2p_pull_kernel (512 workgroups, 16-byte loads, 8 in flight per lane) reading the peer's "
                           "mapped buffer into local memory, every rank pulling from (rank+s)%world at the same time; flag round "
                           "trip = two host-issued one-wave kernels per hop (launch latency included)")
            del mine, dst
    except P2PError as exc:
        out["error"] = str(exc)[:300]
    finally:
        try:
            dist.barrier()          # (no peer may still be reading this rank's buffer when it is freed)
        except Exception:   # noqa: BLE001
            pass
        reg.close()
    return out
