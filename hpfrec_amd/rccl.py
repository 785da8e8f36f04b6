"""A communicator of our own over RCCL's C API (the librccl.so PyTorch ships and has already loaded), for the three
collectives of the sharded iteration -- reduce-scatter, all-gather, all-reduce of float32 -- issued straight onto a
HIP stream.

Why, when torch.distributed is right there: every torch.distributed call creates a Work object that torch's
ProcessGroupNCCL watchdog thread later polls with hipEventQuery.  That costs ~20 us of host time per call, and it is
not safe under stream capture on this image (a watchdog poll of an event recorded into a capture aborted the process
once in ~90 captures, profiles/r02_hipgraph_watchdog_abort.txt).  Calls on a communicator we own are plain stream
work: nothing polls them, RCCL supports capturing them into a hipGraph, and the host cost is one ctypes call.

torch.distributed stays the control plane: it carries the ncclUniqueId from rank 0 to the others and remains the
path for everything that is not on the per-iteration critical path (llk partials, gathers of output tables).
Opt-in (HPF_RCCL_DIRECT=1): this build could execute it with one rank only.
"""
import ctypes
import os

import torch

_NCCL_FLOAT32 = 7     # ncclDataType_t (nccl.h): ... ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat64 = 8
_NCCL_SUM = 0         # ncclRedOp_t


class _UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_byte * 128)]     # (not c_char: ctypes cuts a c_char array at its first NUL)


def _uid_bytes(uid):
    """All 128 bytes of an ncclUniqueId (it is binary: a socket address and a magic number, zeros included)."""
    return ctypes.string_at(ctypes.byref(uid), ctypes.sizeof(uid))


def _uid_from(raw):
    assert len(raw) == ctypes.sizeof(_UniqueId)
    uid = _UniqueId()
    ctypes.memmove(ctypes.byref(uid), raw, len(raw))
    return uid


def _lib():
    path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    L = ctypes.CDLL(path)                       # already mapped by torch: same library instance
    vp, ci, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
    L.ncclGetUniqueId.argtypes = [ctypes.POINTER(_UniqueId)]
    L.ncclCommInitRank.argtypes = [ctypes.POINTER(vp), ci, _UniqueId, ci]
    L.ncclCommDestroy.argtypes = [vp]
    L.ncclAllReduce.argtypes = [vp, vp, sz, ci, ci, vp, vp]
    L.ncclReduceScatter.argtypes = [vp, vp, sz, ci, ci, vp, vp]
    L.ncclAllGather.argtypes = [vp, vp, sz, ci, vp, vp]
    L.ncclGetErrorString.argtypes = [ci]
    L.ncclGetErrorString.restype = ctypes.c_char_p
    for f in (L.ncclGetUniqueId, L.ncclCommInitRank, L.ncclCommDestroy, L.ncclAllReduce, L.ncclReduceScatter,
              L.ncclAllGather):
        f.restype = ci
    return L


class RcclError(RuntimeError):
    pass


class DirectComm:
    """One RCCL communicator spanning the ranks of `dist` (a torch.distributed-like module, or None for a one-rank
    communicator), bound to `device`."""

    def __init__(self, device, dist=None, rank=0, world=1):
        self.L = _lib()
        self.device = torch.device(device)
        self.rank, self.world = int(rank), int(world)
        uid = _UniqueId()
        with torch.cuda.device(self.device):
            if self.rank == 0:
                self._check(self.L.ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
            if self.world > 1:
                # the id travels through torch.distributed (as a byte tensor on the communicator's device type)
                buf = torch.frombuffer(bytearray(_uid_bytes(uid)), dtype=torch.uint8)      # (zeros on the other ranks)
                buf = buf.to(self.device) if dist.get_backend() == "nccl" else buf
                dist.broadcast(buf, 0)
                uid = _uid_from(buf.cpu().numpy().tobytes())
            comm = ctypes.c_void_p()
            self._check(self.L.ncclCommInitRank(ctypes.byref(comm), self.world, uid, self.rank), "ncclCommInitRank")
        self.comm = comm

    def _check(self, rc, what):
        if rc != 0:
            raise RcclError("%s failed: %s" % (what, self.L.ncclGetErrorString(rc).decode()))

    @staticmethod
    def _stream():
        return torch.cuda.current_stream().cuda_stream

    @staticmethod
    def _f32(*tensors):
        for t in tensors:
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise RcclError("DirectComm moves contiguous float32 tensors only")

    # the collectives are enqueued on torch's CURRENT stream, like every kernel launch of this package
    def all_reduce(self, t):
        self._f32(t)
        self._check(self.L.ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), _NCCL_FLOAT32, _NCCL_SUM, self.comm,
                                         self._stream()), "ncclAllReduce")

    def reduce_scatter(self, out, inp):
        self._f32(out, inp)
        assert inp.numel() == out.numel() * self.world
        self._check(self.L.ncclReduceScatter(inp.data_ptr(), out.data_ptr(), out.numel(), _NCCL_FLOAT32, _NCCL_SUM,
                                             self.comm, self._stream()), "ncclReduceScatter")

    def all_gather(self, out, inp):
        self._f32(out, inp)
        assert out.numel() == inp.numel() * self.world
        self._check(self.L.ncclAllGather(inp.data_ptr(), out.data_ptr(), inp.numel(), _NCCL_FLOAT32, self.comm,
                                         self._stream()), "ncclAllGather")

    def close(self):
        if getattr(self, "comm", None) is not None and self.comm.value:
            torch.cuda.synchronize(self.device)
            self.L.ncclCommDestroy(self.comm)
            self.comm = None
