"""ctypes binding of the "one rank's whole sharded iteration from C" entries of include/hpf_hip.h
(hpf_hip_shard_plan_create / _iterate / _join / _exchange_only; hpfrec_amd/csrc/hpf_shard.hip).

cavi.FullBatchCavi (hpfrec_amd/shard.py) builds a ShardPlan over its own device tensors whenever the job runs on a GPU:
an iteration is then ONE host call instead of ~30 (kernel launches through ctypes, torch.distributed collectives, stream
and event operations).  The call-by-call Python forms of the schedules (shard.ShardedMixin) stay as the path of gloo /
CPU stand-in runs, as the checker of the first C-issued iteration, and as the fallback when a plan cannot be created on
every rank.
"""
import ctypes

from . import _lib

MAX_ROW_RANGES = 8      # HPF_MAX_ROW_RANGES
COLL_ALL_REDUCE, COLL_REDUCE_SCATTER, COLL_ALL_GATHER = 0, 1, 2

_vp, _i32, _i64, _f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float

#: hpf_collective_fn: int (*)(void *ctx, int op, const float *send, float *recv, int64_t count, void *stream)
COLLECTIVE_FN = ctypes.CFUNCTYPE(ctypes.c_int, _vp, ctypes.c_int, _vp, _vp, _i64, _vp)


class ShardRange(ctypes.Structure):
    """hpf_shard_range"""
    _fields_ = [("lo", _i64), ("hi", _i64), ("seg_lo", _i64), ("nseg", _i64), ("multi_rows", _vp), ("nmulti", _i64),
                ("short_rows", _i32), ("reserved", _i32)]


class ShardDesc(ctypes.Structure):
    """hpf_shard_desc (field for field; tests/test_host_logic.py checks the size against the header's)."""
    _fields_ = [
        ("world", _i32), ("rank", _i32), ("k", _i32), ("ld", _i32),
        ("nU", _i64), ("nI", _i64),
        ("u_segs", _vp), ("u_nseg", _i64), ("u_idx", _vp), ("u_y", _vp),
        ("u_row_seg_ptr", _vp), ("u_multi_rows", _vp), ("u_nmulti", _i64),
        ("i_segs", _vp), ("i_idx", _vp), ("i_y", _vp), ("i_row_seg_ptr", _vp),
        ("nranges", _i32), ("pad0", _i32), ("ranges", ShardRange * MAX_ROW_RANGES),
        ("eB", _vp), ("part_u", _vp), ("part_i", _vp),
        ("Gamma_shp", _vp), ("Theta", _vp), ("k_rte", _vp), ("k_rte_prev", _vp),
        ("Lambda_shp", _vp), ("Beta", _vp), ("t_rte", _vp), ("t_rte_prev", _vp),
        ("csT", _vp), ("csB", _vp), ("csB_used", _vp),
        ("csT_part", _vp), ("csT_part_rows", _i32), ("user_sweep_grid", _i32), ("user_multi_grid", _i32), ("pad1", _i32),
        ("csB_part", _vp), ("csB_part_rows", _i32), ("pad2", _i32),
        ("acc_i", _vp), ("acc_own", _vp), ("e_own", _vp),
        ("e_own_ld", _i32), ("item_sweep_grid", _i32),
        ("ag_recv", _vp),
        ("a", _f32), ("k_shp", _f32), ("add_k_rte", _f32), ("c", _f32), ("t_shp", _f32), ("add_t_rte", _f32),
        ("comm", _vp), ("coll", COLLECTIVE_FN), ("coll_ctx", _vp),
        ("xstream", _vp),
        ("dry_run", _i32), ("schedule", _i32),
        ("shp_own", _vp),
        ("dry_run_busbw_GBps", _f32), ("dry_run_latency_us", _f32),
        ("dry_run_footprint_blocks", _i32), ("direct_prefetch", _i32),
        ("direct_pull_grid", _i32), ("direct_gather_gx", _i32),
        ("p2p_region", _vp), ("p2p_acc_offset", _i64), ("p2p_send_offset", _i64),
    ]


class TraceRec(ctypes.Structure):
    """hpf_shard_trace_rec: one issued operation of a traced plan (desc.dry_run = 2)."""
    _fields_ = [("kind", _i32), ("id", _i32), ("stream", _i64), ("arg", _i64)]


TRACE_KERNEL, TRACE_COLLECTIVE, TRACE_RECORD, TRACE_WAIT, TRACE_COPY = 1, 2, 3, 4, 5
TRACE_KERNELS = {1: "sweep", 2: "segsum", 3: "sweep_finalize", 4: "row_finalize", 5: "row_finalize_ranges", 6: "colsum_reduce",
                 7: "item_shape", 8: "item_apply", 10: "pull_reduce", 11: "gather_pull", 12: "colsum_allreduce",
                 13: "signal", 14: "wait"}
TRACE_EVENT_BASE = 0x1000


class ShardPlan:
    """Owns one hpf_shard plan.  `keep`: whatever must outlive the plan on the Python side (tensors whose pointers the
    descriptor holds, the callback object of a stand-in collective)."""

    def __init__(self, desc, keep=()):
        self.L = _lib.lib()
        self.desc = desc
        self.keep = list(keep)
        h = ctypes.c_void_p()
        rc = self.L.hpf_hip_shard_plan_create(ctypes.byref(desc), ctypes.byref(h))
        if rc != 0:
            raise _lib.HpfHipError("hpf_hip_shard_plan_create failed with code %d" % rc)
        self.handle = h

    def iterate(self, eT, eT_next, store, stream):
        _lib.check(self.L.hpf_hip_shard_iterate(self.handle, eT.data_ptr(), eT_next.data_ptr(), int(bool(store)), stream),
                   "hpf_hip_shard_iterate")

    def join(self, stream):
        _lib.check(self.L.hpf_hip_shard_join(self.handle, stream), "hpf_hip_shard_join")

    def status(self):
        """Direct schedule: synchronises the device and raises when a wait for a peer's flag ran out (no-op otherwise)."""
        rc = self.L.hpf_hip_shard_status(self.handle)
        if rc != 0:
            raise _lib.HpfHipError("hpfrec_amd: the direct exchange timed out waiting for a peer (code %d)" % rc)

    def exchange_only(self, op, rng, stream):
        _lib.check(self.L.hpf_hip_shard_exchange_only(self.handle, int(op), int(rng), stream),
                   "hpf_hip_shard_exchange_only")

    def iterate_raw(self, eT, eT_next, store, stream):
        """iterate() on raw pointer values (a traced plan: nothing is dereferenced)."""
        _lib.check(self.L.hpf_hip_shard_iterate(self.handle, eT, eT_next, int(bool(store)), stream), "hpf_hip_shard_iterate")

    def trace(self):
        """The operations a traced plan (desc.dry_run = 2) has issued since the last call, in order: a list of
        (kind, id, stream, arg) tuples (include/hpf_hip.h, HPF_TRACE_*)."""
        n = ctypes.c_int64(0)
        _lib.check(self.L.hpf_hip_shard_trace(self.handle, None, 0, ctypes.byref(n)), "hpf_hip_shard_trace")
        buf = (TraceRec * max(1, n.value))()
        _lib.check(self.L.hpf_hip_shard_trace(self.handle, ctypes.addressof(buf), n.value, ctypes.byref(n)),
                   "hpf_hip_shard_trace")
        return [(r.kind, r.id, r.stream, r.arg) for r in buf[: n.value]]

    def close(self):
        if getattr(self, "handle", None) is not None and self.handle.value:
            self.L.hpf_hip_shard_plan_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001
            pass
