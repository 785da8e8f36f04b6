"""A communicator of our own over RCCL's C API, for the three collectives of the sharded iteration -- reduce-scatter,
all-gather, all-reduce of float32 -- issued straight onto a HIP stream.

Why, when torch.distributed is right there: every torch.distributed call creates a Work object that torch's
ProcessGroupNCCL watchdog thread later polls with hipEventQuery.  That costs ~20 us of host time per call, and it is
not safe under stream capture on this image (a watchdog poll of an event recorded into a capture aborted the process
once in ~90 captures, profiles/r02_hipgraph_watchdog_abort.txt).  Calls on a communicator we own are plain stream
work: nothing polls them, RCCL supports capturing them into a hipGraph, and the sharded iteration can be issued whole
from C (hpf_hip_shard_iterate, include/hpf_hip.h).

RCCL is reached through libhpf_hip.so (hpf_hip_rccl_*): the library resolves RCCL's entry points with dlsym from the
librccl.so PyTorch ships and has already loaded -- same library instance as torch.distributed's, nothing linked, and no
struct passed by value through ctypes (ncclCommInitRank takes its 128-byte id by value; that call is made in C).

torch.distributed stays the control plane: it carries the ncclUniqueId from rank 0 to the others and remains the
path for everything that is not on the per-iteration critical path (llk partials, gathers of output tables).
"""
import atexit
import ctypes
import os

import torch

from . import _lib

UID_BYTES = 128       # sizeof(ncclUniqueId)


class RcclError(RuntimeError):
    pass


def rccl_path():
    return os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")


def open_rccl():
    """Bind RCCL's entry points inside libhpf_hip.so (idempotent) -> the library handle."""
    L = _lib.lib()
    rc = L.hpf_hip_rccl_open(rccl_path().encode())
    if rc != 0:
        rc = L.hpf_hip_rccl_open(None)          # (whatever "librccl.so" the loader finds)
    if rc != 0:
        raise RcclError("hpfrec_amd: RCCL entry points could not be bound (code %d)" % rc)
    return L


def _check(rc, what):
    if rc != 0:
        # HPF_ERCCL_BASE - ncclResult_t (include/hpf_hip.h)
        raise RcclError("%s failed: %s" % (what, "ncclResult_t %d" % (-1000 - rc) if rc <= -1000 else "code %d" % rc))


def new_unique_id():
    """All 128 bytes of a fresh ncclUniqueId (binary: a socket address and a magic number, zeros included)."""
    L = open_rccl()
    buf = (ctypes.c_uint8 * UID_BYTES)()
    _check(L.hpf_hip_rccl_unique_id(ctypes.addressof(buf)), "ncclGetUniqueId")
    return bytes(buf)


_LIVE = []


@atexit.register
def _close_all():
    for c in list(_LIVE):
        try:
            c.close()
        except Exception:   # noqa: BLE001  (interpreter shutdown: the runtime may be gone already)
            pass


class DirectComm:
    """One RCCL communicator spanning the ranks of `dist` (a torch.distributed-like module, or None for a one-rank
    communicator), bound to `device`."""

    def __init__(self, device, dist=None, rank=0, world=1):
        self.L = open_rccl()
        self.device = torch.device(device)
        self.rank, self.world = int(rank), int(world)
        raw = bytes(UID_BYTES)
        with torch.cuda.device(self.device):
            if self.rank == 0:
                raw = new_unique_id()
            if self.world > 1:
                # the id travels through torch.distributed (as a byte tensor on the communicator's device type)
                buf = torch.frombuffer(bytearray(raw), dtype=torch.uint8)      # (zeros on the other ranks)
                buf = buf.to(self.device) if dist.get_backend() == "nccl" else buf
                dist.broadcast(buf, 0)
                raw = buf.cpu().numpy().tobytes()
            assert len(raw) == UID_BYTES
            uid = (ctypes.c_uint8 * UID_BYTES).from_buffer_copy(raw)
            comm = ctypes.c_void_p()
            _check(self.L.hpf_hip_rccl_comm_init(ctypes.byref(comm), self.world, self.rank, ctypes.addressof(uid)),
                   "ncclCommInitRank")
        self.comm = comm
        _LIVE.append(self)

    @property
    def handle(self):
        """ncclComm_t as an integer (for hpf_shard_desc.comm)."""
        return self.comm.value

    def count(self):
        """Ranks of the communicator as RCCL itself reports them (ncclCommCount)."""
        n = ctypes.c_int(0)
        _check(self.L.hpf_hip_rccl_comm_count(self.comm, ctypes.byref(n)), "ncclCommCount")
        return n.value

    def _stream(self):
        # torch's current stream OF THE COMMUNICATOR'S DEVICE (not of whatever device is current)
        return torch.cuda.current_stream(self.device).cuda_stream

    @staticmethod
    def _f32(*tensors):
        for t in tensors:
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise RcclError("DirectComm moves contiguous float32 tensors only")

    # the collectives are enqueued on torch's CURRENT stream of the device, like every kernel launch of this package
    def all_reduce(self, t):
        self._f32(t)
        _check(self.L.hpf_hip_rccl_all_reduce_f32(self.comm, t.data_ptr(), t.numel(), self._stream()), "ncclAllReduce")

    def reduce_scatter(self, out, inp):
        self._f32(out, inp)
        assert inp.numel() == out.numel() * self.world
        _check(self.L.hpf_hip_rccl_reduce_scatter_f32(self.comm, inp.data_ptr(), out.data_ptr(), out.numel(),
                                                      self._stream()), "ncclReduceScatter")

    def all_gather(self, out, inp):
        self._f32(out, inp)
        assert out.numel() == inp.numel() * self.world
        _check(self.L.hpf_hip_rccl_all_gather_f32(self.comm, inp.data_ptr(), out.data_ptr(), inp.numel(),
                                                  self._stream()), "ncclAllGather")

    def self_check(self):
        """One synchronous all-reduce of a single 1.0: True when every rank of the communicator took part."""
        t = torch.ones(1, dtype=torch.float32, device=self.device)
        self.all_reduce(t)
        torch.cuda.synchronize(self.device)
        return float(t.item()) == float(self.world) and self.count() == self.world

    def close(self):
        if getattr(self, "comm", None) is not None and self.comm.value:
            if self in _LIVE:
                _LIVE.remove(self)
            try:
                torch.cuda.synchronize(self.device)
            except Exception:   # noqa: BLE001
                pass
            self.L.hpf_hip_rccl_comm_destroy(self.comm)
            self.comm = None
