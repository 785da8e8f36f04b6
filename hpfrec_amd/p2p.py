"""Peer-mapped exchange regions (include/hpf_hip.h, "Multi-GPU, direct exchange"; hpfrec_amd/csrc/hpf_p2p.hip).

SURVEY.md section 8(e) wants the item statistics of the user-sharded iteration to cross xGMI directly -- every GPU
talking to its seven peers at once -- instead of through a ring collective.  A `PeerRegion` is this rank's share of
that: one device buffer (the packed item accumulators and the finished [numerators | base rate] rows, layout chosen by
cavi.FullBatchCavi) plus a small control block of flags, both exported with hipIpcGetMemHandle and mapped by every other
rank of the node.  The 128 handle bytes per rank travel through torch.distributed (all_gather_object: the control
plane), exactly like the ncclUniqueId of rccl.DirectComm.  The reference has no counterpart (single-node OpenMP,
/root/reference/hpfrec/cython_loops.pxi:4).
"""
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
        """Frees the region.  No collective in here (it may run from a destructor): the caller closes only after a host-side
        collective every rank entered with its device synchronised (the end of a fit, the first-iteration vote), so that no
        peer's pull can still be reading this memory."""
        if getattr(self, "handle", None) is not None and self.handle.value:
            self.L.hpf_hip_p2p_region_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001
            pass
