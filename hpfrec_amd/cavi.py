"""Full-batch CAVI driver: the device-side counterpart of the loop body of fit_hpf
(/root/reference/hpfrec/cython_loops.pxi:227-259, "PXI").

Statement order of the reference iteration, which this driver preserves:
  (1) phi from the *old* shapes/rates                       PXI:232   -> both sweeps read eT/eB of the previous iteration
  (2) Gamma_rte = k_shp/k_rte + colsum(Beta_old)            PXI:236   -> user row_finalize (cs_other = csB)
  (3) Gamma_shp = a + sum phi ; Lambda_shp = c + sum phi    PXI:239-249
  (4) Theta = Gamma_shp/Gamma_rte                           PXI:251
  (5) Lambda_rte = t_shp/t_rte + colsum(Theta_new)          PXI:255   -> item row_finalize (cs_other = csT) runs after the user side
  (6) Beta = Lambda_shp/Lambda_rte                          PXI:256
  (7) k_rte = a'/b' + rowsum(Theta); t_rte = c'/d' + rowsum(Beta)   PXI:258-259

Multi-GPU (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm): users are
sharded in contiguous nnz-balanced ranges, the item E table is replicated, and the exchange per iteration is
the sum of the item accumulators (nI*k floats, overlapped with the user side) plus k-float all-reduces of the
column sums -- either as a reduce-scatter / sharded item finalizer / all-gather of the new E rows ("scatter",
_iterate_scatter) or as an all-reduce with a replicated finalizer ("allreduce", _iterate_sharded).

The kernels are reached through an `ops` object (hpfrec_amd.ops_hip.HipOps).  There is no CPU
implementation in this package.
"""
import contextlib
import os

import numpy as np
import torch

from . import _lib, _streams, layout


class Hyper:
    """Hyper-parameters rounded to float32 as the reference's `real_t` arguments are
    (PXI:147-148) and the derived constants of PXI:173-174 and PXI:209-210."""

    def __init__(self, k, a, a_prime, b_prime, c, c_prime, d_prime):
        f = np.float32
        self.k = int(k)
        self.a, self.a_prime, self.b_prime = f(a), f(a_prime), f(b_prime)
        self.c, self.c_prime, self.d_prime = f(c), f(c_prime), f(d_prime)
        self.k_shp = f(self.a_prime + f(self.k) * self.a)
        self.t_shp = f(self.c_prime + f(self.k) * self.c)
        self.add_k_rte = f(self.a_prime / self.b_prime)
        self.add_t_rte = f(self.c_prime / self.d_prime)


def _dist():
    """torch.distributed when this process is one rank of a multi-rank job, else None.
    HPF_FORCE_SHARDED=1 takes the sharded code path even with a single rank (used to exercise the
    RCCL/packed-exchange path on a one-GPU box)."""
    import os
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and (
            dist.get_world_size() > 1 or os.environ.get("HPF_FORCE_SHARDED") == "1"):
        return dist
    return None


_DIRECT_COMMS = {}
_side_stream = _streams.side_stream      # (process-wide side streams: hpfrec_amd/_streams.py)
NATIVE_PLANS_CREATED = [0]      # how many models of this process run their sharded iteration from C (tests, bench)


def _direct_comm(dist, device, rank, world):
    """The process-wide DirectComm for (device, world) when HPF_RCCL_DIRECT=1 and the job runs on RCCL; else None
    (torch.distributed carries the collectives).  A stand-in for torch.distributed may bring its own (`direct_comm`)."""
    if dist is None or os.environ.get("HPF_RCCL_DIRECT", "0") != "1" or torch.device(device).type != "cuda":
        return None
    if hasattr(dist, "direct_comm"):
        return dist.direct_comm(device)
    try:
        if dist.get_backend() != "nccl":
            return None
    except Exception:   # noqa: BLE001
        return None
    key = (str(device), world, rank)
    if key not in _DIRECT_COMMS:
        from . import rccl
        _DIRECT_COMMS[key] = rccl.DirectComm(device, dist, rank, world)
    return _DIRECT_COMMS[key]


def shard_users(ix_u, ix_i, y, nU, rank, world):
    """Keep the nonzeros of this rank's contiguous, nnz-balanced user range.
    Returns (local ix_u, ix_i, y, (u0, u1))."""
    if world <= 1:
        return ix_u, ix_i, y, (0, int(nU))
    counts = torch.bincount(ix_u.to(torch.int64), minlength=int(nU))
    indptr = torch.zeros(int(nU) + 1, dtype=torch.int64, device=ix_u.device)
    torch.cumsum(counts, 0, out=indptr[1:])
    u0, u1 = layout.nnz_balanced_ranges(indptr, world)[rank]
    keep = (ix_u >= u0) & (ix_u < u1)
    return ix_u[keep] - u0, ix_i[keep], y[keep], (u0, u1)


class _SideView:
    """The segments [seg_lo, seg_hi) of a SparseSide, as the sweep launcher sees a side."""

    def __init__(self, side, seg_lo, seg_hi, nnz=None):
        self.seg_lo = seg_lo
        self.segs = side.segs[seg_lo:seg_hi]
        self.nseg = seg_hi - seg_lo
        self.idx, self.y = side.idx, side.y
        # launch hint (hpf_hip_sweep_f32): a shard of a many-rank run leaves ~16 nonzeros per item row
        self.short_rows = 1 if (nnz is not None and self.nseg > 0 and
                                                   nnz / self.nseg < layout.SHORT_ROW_NNZ) else 0


def mt19937_state_words(random_seed):
    """numpy's MT19937 state for this seed (seed <= 0 / None: OS entropy, PXI:127) as int32[625]: key + pos."""
    bg = np.random.MT19937(seed=random_seed if (random_seed is not None and random_seed > 0) else None)
    st = bg.state["state"]
    words = np.concatenate([st["key"].astype(np.uint32), np.array([st["pos"]], dtype=np.uint32)])
    return torch.from_numpy(words.view(np.int32).copy())


def draw_init_words(ops, mt_state, nU, nI, k):
    """The 2*(nU + nI)*k stream words initialize_parameters consumes (PXI:127-138), on the current stream; `mt_state`
    (device int32[625]) is advanced.  Long draws are walked by 512-1024 workgroups at once after a polynomial jump-ahead
    (hpf_mt19937.hip: 2.9 ms for C3's 138M words, 4.2 ms for C5's 552M; one workgroup: 61 / 242 ms); callers still start
    it on a side stream before the CSR/CSC build."""
    raw = torch.empty(2 * (int(nU) + int(nI)) * int(k), dtype=torch.int32, device=mt_state.device)
    ops.mt19937_words(mt_state, raw)
    return raw


class FullBatchCavi:
    """Device-resident state + one-iteration step for (a shard of) the HPF model."""

    def __init__(self, ops, device, ix_u, ix_i, y, nU, nI, hyper, seg_cap=layout.SEG_CAP):
        """ix_u/ix_i/y: COO triplets of THIS rank (user ids local to the shard), torch tensors."""
        self.ops = ops
        self.device = torch.device(device)
        self.hy = hyper
        self.k = hyper.k
        self.ld = _lib.ld_for_k(self.k)
        self.nU, self.nI = int(nU), int(nI)
        dev = self.device
        self.users, self.items, self.u_sorted = layout.build_sides(ix_u.to(dev), ix_i.to(dev), y.to(dev),
                                                                   self.nU, self.nI, seg_cap)
        self.nnz = self.users.nnz
        self.dist = _dist()
        # sweep grid of THIS model (the op set is shared): sharded launches cover short item ranges and want fewer,
        # fatter blocks (16 per CU costs 3 % at N=8, gains 1 % at N=1).  THREE workgroups per CU, not the four that fill
        # every wave slot the fused sweep's 104 VGPRs allow: the user sweep is the kernel the exchange hides under, and
        # a collective's kernel (or the exchange stream's shape kernel) must become RESIDENT beside it.  With four, a
        # stand-in of a collective kernel's footprint (32 workgroups x 256 threads, 128 VGPRs, 64 KB of LDS) took slots
        # from the sweep's persistent grid and stretched it from 240 to 280-295 us; with three the sweep itself is 6 %
        # slower and the 8-rank iteration at an emulated 300 GB/s 4-12 % faster in every schedule
        # (profiles/r03_shard_probe_collective_footprint.txt)
        self.sweep_blocks = ops.sweep_blocks
        if self.dist and "HPF_SWEEP_BPC" not in os.environ and hasattr(ops, "cu_count"):
            self.sweep_blocks = max(1, ops.cu_count) * int(os.environ.get("HPF_SHARD_SWEEP_BPC", "3"))
        # the sharded item pass is many short rows: more, smaller blocks even out its tail (tools/sweep_micro.py:
        # 203 us at 32 blocks per CU vs 220 at 8 for rank 0 of 8 at C3)
        self.item_sweep_blocks = max(1, getattr(ops, "cu_count", 1)) * int(os.environ.get("HPF_ITEM_SWEEP_BPC", "32")) \
            if hasattr(ops, "cu_count") else self.sweep_blocks
        ld = self.ld
        f32 = dict(dtype=torch.float32, device=dev)
        z = lambda n: torch.zeros((n, ld), **f32)
        # sharded exchange: "scatter" = reduce-scatter of the item statistics, each rank finalizes 1/N of the items,
        # all-gather of the new E rows; "allreduce" = all-reduce + replicated (deferred) finalizer
        self.world = self.dist.get_world_size() if self.dist else 1
        self.rank = self.dist.get_rank() if self.dist else 0
        # default: "scatter" at every rank count -- it is the mode the C-issued iteration exists for, and per rank it costs
        # what the all-reduce form costs at 2 ranks and less from 4 on (C3, tools/shard_probe.py, round 3: N=2 1.91-1.93
        # vs 1.91-1.93 ms, N=4 1.01 vs 1.11-1.15, N=8 0.54-0.55 vs 0.80; profiles/r03_shard_probe_n2_n4.txt)
        self.shard_mode = os.environ.get("HPF_SHARD_MODE", "scatter") if self.dist else None
        assert self.shard_mode in (None, "scatter", "allreduce"), self.shard_mode
        # item ranges per iteration.  scatter mode: the all-gather of range j+1 hides under the sweep of range j and
        # the first range's is exposed, so more ranges expose less -- but each extra range costs 0.06 ms of launches
        # and stream dependencies per iteration at 8 ranks (tools/shard_probe.py): two ranges (18 % / 82 % of the rows)
        # (what the order of the item ranges depends on; the full comment is where the switches are read again below)
        self._early_order = self.shard_mode == "scatter" and os.environ.get("HPF_GATHER_EARLY", "1") in ("1", "2") and \
            os.environ.get("HPF_RS_ALLTOALL", "0") != "1" and os.environ.get("HPF_ITEM_STREAM", "0") != "1"
        default_chunks = "2" if self.shard_mode == "scatter" else "3"
        nchunks = int(os.environ.get("HPF_AR_CHUNKS", default_chunks))
        self.item_bounds = self._item_bounds(nchunks) if self.dist else None
        # scatter mode: item tables carry a few pad rows so that every range splits into N equal slices
        nIa = self.nI_alloc = self.item_bounds[-1][1] if self.shard_mode == "scatter" else self.nI
        self.Gamma_shp, self.Gamma_rte, self.Theta = z(self.nU), z(self.nU), z(self.nU)
        self.Lambda_shp, self.Lambda_rte, self.Beta = z(nIa), z(nIa), z(nIa)
        self.k_rte = torch.zeros(self.nU, **f32)
        self.t_rte = torch.ones(nIa, **f32)
        # Gamma_rte = k_shp/k_rte_old + colsum(Beta_old) and Lambda_rte = t_shp/t_rte_old + colsum(Theta) are rank-1
        # (row scalar + column vector): the iteration keeps only these factors (n + k floats per side) and the
        # [n,k] rate tables are expanded on output (fetch), bit-identically to what the kernels would store
        self.k_rte_prev = torch.zeros(self.nU, **f32)
        self.t_rte_prev = torch.ones(nIa, **f32)
        self.csB_used = torch.zeros(ld, **f32)
        self.rte_factored = False
        self.eT, self.eT_next, self.eB = z(self.nU), z(self.nU), z(nIa)
        self.part_u = torch.empty((max(1, self.users.nseg), ld), **f32)
        self.part_i = torch.empty((max(1, self.items.nseg), ld), **f32)
        # column-sum partials: [fused-sweep blocks | finalize blocks of the rows the sweep cannot finish]
        self.fused = True
        self.gsu = ops.sweep_grid(self.users.nseg, self.sweep_blocks)
        self.gsi = ops.sweep_grid(self.items.nseg, self.sweep_blocks)
        self.gu, self.gi = ops.finalize_grid(self.nU), ops.finalize_grid(self.nI)
        self.csT_part = torch.zeros((self.gsu + self.gu, ld), **f32)
        self.csB_part = torch.zeros((self.gsi + self.gi, ld), **f32)   # re-sized below for the sharded path
        self.cs_scratch = torch.zeros((max(self.gu, self.gi), ld), **f32)  # for whole-table column sums
        self.csB = torch.zeros(ld, **f32)
        # multi-GPU exchange buffer: the item accumulators packed to k columns (pads are zero: not sent),
        # cut into nnz-balanced item ranges so that the all-reduce of one range overlaps the sweep of the next
        self.acc_i = torch.zeros((nIa, self.k), **f32) if self.dist else None
        self.item_chunks = self._item_chunks() if self.dist else None
        self._tables_split = False
        if self.dist:   # one block range of column-sum partials per item range
            rows = self.gsi + sum(ops.finalize_grid(hi - lo) for lo, hi, _, _ in self.item_chunks)
            self.csB_part = torch.zeros((max(rows, self.gsi + self.gi), ld), **f32)
        self.csT = torch.zeros(ld, **f32)
        self.niter_done = 0
        self._chunk_views = None
        self.rs_alltoall = os.environ.get("HPF_RS_ALLTOALL", "0") == "1"   # scatter mode: all-to-all + local sum
        # scatter mode, opt-in: the per-iteration collectives on an RCCL communicator of our own (hpfrec_amd/rccl.py)
        # instead of torch.distributed -- no Work objects, no watchdog polls, capturable into hipGraphs
        self.comm = _direct_comm(self.dist, self.device, self.rank, self.world) \
            if (self.shard_mode == "scatter" and not self.rs_alltoall) else None
        self.item_stream = os.environ.get("HPF_ITEM_STREAM", "0") == "1"   # scatter mode: item pass on its own stream
        # scatter mode, HPF_AG_PACKED=1: the new E rows are all-gathered k-PACKED (the pad columns -- 22 % of an ld = 64
        # row at k = 50 -- stay off the links) into the exchange buffer and a streaming kernel restores the padded layout
        # the sweeps gather from.  Off by default: the unpack launch costs 30-40 us per C3 iteration at 8 ranks
        # (profiles/r03_shard_probe_native_c3_c4.txt) against ~50 us of link time it would save at 300 GB/s -- only a
        # run on real links can decide, so it is one of bench.py's autotune candidates
        self.ag_packed = self.shard_mode == "scatter" and not self.rs_alltoall and \
            os.environ.get("HPF_AG_PACKED", "0") == "1"
        # scatter mode, HPF_GATHER_EARLY=1: the "gather-early" schedule -- the item finalizer split in two
        # (hpf_hip_item_shape_rows_f32 right after the reduce-scatters, hpf_hip_item_apply_rows_f32 on every rank once
        # colsum(Theta) is known), so that the all-gather of the new item expectations runs UNDER THE USER SWEEP instead
        # of after it (include/hpf_hip.h, HPF_SCHEDULE_GATHER_EARLY).  One more float32 rounding in
        # the E rows than the one-part finalizer; everything else is the same arithmetic.  ON by default: on one GPU
        # (collectives emulated by local copies) it costs 0-40 us more compute per iteration than finalize-then-gather
        # (profiles/r03_shard_probe_gather_early.txt), but it is the only schedule in which the all-gather -- 85 MB per
        # rank at C3, 0.15-0.4 ms on xGMI depending on the rank count -- has something to hide under; bench.py's autotune
        # measures both on whatever links it runs on
        self.gather_early = self.shard_mode == "scatter" and not self.rs_alltoall and not self.item_stream and \
            os.environ.get("HPF_GATHER_EARLY", "1") in ("1", "2")
        # HPF_GATHER_EARLY=2, "gather-carried" (C-issued iteration only; opt-in until it has run on real links): the
        # exchange of iteration t runs on into iteration t+1 -- range j's apply half is carried to just ahead of that
        # range's next item sweep, so its all-gather has a whole iteration to hide under; needs a second communicator
        # for the k-float all-reduces (include/hpf_hip.h, HPF_SCHEDULE_GATHER_CARRIED)
        self.gather_carried = self.gather_early and os.environ.get("HPF_GATHER_EARLY", "1") == "2"
        if self.gather_early:
            self.ag_packed = False
        # scatter mode on RCCL: the whole iteration issued by ONE C call (hpf_hip_shard_iterate) on a communicator of
        # our own; HPF_NATIVE_SHARD=0 keeps the call-by-call Python form (also the path of gloo / stand-in runs)
        self._plan = None
        self.native = True            # (False: issue call by call even when a plan exists -- bench's per-kernel event pass)
        self._last_native = False
        self.native_error = None
        self.lazy_items = os.environ.get("HPF_LAZY_ITEMS", "1") == "1"
        self.item_pending = False   # sharded path: acc_i holds reduced statistics not yet applied to the item tables

    # ------------------------------------------------------------------------------------
    def _pad(self, host_arr, out):
        t = torch.from_numpy(np.ascontiguousarray(host_arr, dtype=np.float32)).to(self.device)
        out.zero_()
        out[: t.shape[0], : self.k] = t.view(-1, self.k)   # item tables may carry pad rows (scatter mode)

    def load_state(self, Gamma_shp, Gamma_rte, Lambda_shp, Lambda_rte, k_rte, t_rte, Theta, Beta):
        """Upload host arrays ([n,k] / [n,1], this rank's user rows) and derive eT, eB and colsum(Beta)."""
        self._pad(Gamma_shp, self.Gamma_shp)
        self._pad(Gamma_rte, self.Gamma_rte)
        self._pad(Lambda_shp, self.Lambda_shp)
        self._pad(Lambda_rte, self.Lambda_rte)
        self._pad(Theta, self.Theta)
        self._pad(Beta, self.Beta)
        self.k_rte.copy_(torch.from_numpy(np.ascontiguousarray(k_rte, dtype=np.float32).reshape(-1)))
        self.t_rte[: self.nI].copy_(torch.from_numpy(np.ascontiguousarray(t_rte, dtype=np.float32).reshape(-1)))
        self.item_pending = False
        self._tables_split = False
        if self.shard_mode == "scatter":
            self._sync_scatter()      # only waits for exchanges still in flight
        self.rte_factored = False
        self.refresh_expectations()

    def init_state(self, raw, u0=0, nU_global=None):
        """initialize_parameters (PXI:117-143) without the host: the four tables are prior + 0.01*U drawn from ONE
        MT19937 stream in the reference's order -- user rates, item rates, user shapes, item shapes, all with a'/c'
        -- bit-identical to numpy's Generator.random(dtype=float32); the means are the ratios, k_rte = b', t_rte = d'.
        `raw`: the stream's first 2*(nU_global + nI)*k state words (`draw_init_words`).  A rank of a sharded fit
        keeps rows u0 <= r < u0 + nU of the user tables of nU_global rows (every rank walks the whole stream: it is
        sequential)."""
        ops, hy, k, ld = self.ops, self.hy, self.k, self.ld
        nUg = self.nU if nU_global is None else int(nU_global)
        for t in (self.Gamma_shp, self.Gamma_rte, self.Lambda_shp, self.Lambda_rte, self.Theta, self.Beta):
            t.zero_()
        nI, nU = self.nI, self.nU
        assert raw.numel() == 2 * (nUg + nI) * k
        mine = slice(u0 * k, (u0 + nU) * k)          # this rank's rows of a user table's words
        draws = (raw[: nUg * k][mine], raw[nUg * k: (nUg + nI) * k],
                 raw[(nUg + nI) * k: (2 * nUg + nI) * k][mine], raw[(2 * nUg + nI) * k:])
        ops.uniform_rows(draws[0], self.Gamma_rte, nU, k, ld, hy.a_prime, 0.01)
        ops.uniform_rows(draws[1], self.Lambda_rte, nI, k, ld, hy.c_prime, 0.01)
        ops.uniform_rows(draws[2], self.Gamma_shp, nU, k, ld, hy.a_prime, 0.01, den=self.Gamma_rte, ratio=self.Theta)
        ops.uniform_rows(draws[3], self.Lambda_shp, nI, k, ld, hy.c_prime, 0.01, den=self.Lambda_rte, ratio=self.Beta)
        self.k_rte.fill_(float(hy.b_prime))
        self.t_rte[: self.nI].fill_(float(hy.d_prime))
        self.item_pending = False
        self._tables_split = False
        if self.shard_mode == "scatter":
            self._sync_scatter()
        self.rte_factored = False
        self.refresh_expectations()

    def refresh_expectations(self):
        ops, k, ld = self.ops, self.k, self.ld
        ops.expect(self.Gamma_shp, self.Gamma_rte, self.eT, self.nU, k, ld)
        ops.expect(self.Lambda_shp, self.Lambda_rte, self.eB, self.nI, k, ld)
        ops.colsum(self.Beta, self.nI, ld, self.cs_scratch)
        ops.colsum_reduce(self.cs_scratch, self.csB, ld)

    def _item_bounds(self, nchunks):
        """Contiguous item ranges [(lo, hi)] with ~equal GLOBAL nonzeros (identical on every rank).  Scatter mode:
        every range is a multiple of the world size long; the last one runs past nI into pad rows."""
        it = self.items
        deg = (it.indptr[1:] - it.indptr[:-1]).clone()
        self.dist.all_reduce(deg)
        # HPF_RANGE_ROW_WEIGHT = w: a row counts as its nonzeros + w x the mean row's (0: equal nonzeros = equal sweep
        # time, 18 % / 82 % of the rows at C3; large: equal rows = equal exchange bytes).  Scatter mode: 2 (31 % / 69 %
        # of the nonzeros) -- the range swept first is the one with most rows, and with equal nonzeros its exchange (82 %
        # of the bytes) ended after the iteration did: at an emulated 300 GB/s the 8-rank iteration went 0.87 -> 0.83 ms
        # (finalize-then-gather), 0.80 -> 0.78 (gather-early), 0.72 -> 0.63 (gather-carried), at no cost without link
        # time (profiles/r03_shard_probe_gather_carried.txt, "range split")
        w = float(os.environ.get("HPF_RANGE_ROW_WEIGHT", "2" if self.shard_mode == "scatter" else "0"))
        if w > 0 and self.nI > 0:
            deg = deg + int(round(w * float(deg.sum().item()) / self.nI))
        gptr = torch.zeros(self.nI + 1, dtype=torch.int64, device=deg.device)
        torch.cumsum(deg, 0, out=gptr[1:])
        cuts = [lo for lo, _ in layout.nnz_balanced_ranges(gptr, max(1, nchunks))] + [self.nI]
        if self.shard_mode == "scatter":
            W, fixed = self.world, [0]
            for c in cuts[1:-1]:
                c = fixed[-1] + ((c - fixed[-1] + W - 1) // W) * W
                if fixed[-1] < c < self.nI:
                    fixed.append(c)
            fixed.append(fixed[-1] + ((self.nI - fixed[-1] + W - 1) // W) * W)
            cuts = fixed
        return [(lo, hi) for lo, hi in zip(cuts[:-1], cuts[1:]) if hi > lo]

    def _item_chunks(self):
        """[(row_lo, row_hi, SideView over the range's segments, its split/empty rows)] in issue order."""
        it = self.items
        rsp = it.row_seg_ptr.cpu()
        ptr = it.indptr.cpu()
        out = []
        for lo, hi in self.item_bounds:
            top = min(hi, self.nI)
            multi = it.multi_rows[(it.multi_rows >= lo) & (it.multi_rows < top)].contiguous()
            out.append((lo, hi, _SideView(it, int(rsp[lo]), int(rsp[top]), nnz=int(ptr[top] - ptr[lo])), multi))
        if self.shard_mode == "scatter" and self._early_order:
            # gather-early: MOST rows first.  The ranges hold equal nonzeros (equal sweep time) but very different row
            # counts (= exchange bytes: 82 % / 18 % at C3 with two ranges); the whole exchange -- reduce-scatters, then the
            # all-gather -- runs under what is left of the iteration, so the big reduce-scatter must start after the FIRST
            # sweep, not the last (profiles/r03_timeline_links_*.txt: -0.1 ms of exposed exchange at 300 GB/s emulated)
            out.sort(key=lambda c: c[0] - c[1])
        elif self.shard_mode == "scatter":
            # finalize-then-gather: fewest rows first -- the all-gather of the big range (the tail items) then overlaps
            # the sweep of the small one in the next iteration, and its reduce-scatter overlaps the user side in this one
            out.sort(key=lambda c: c[1] - c[0])
        else:
            # issue order: most rows (= largest all-reduce payload) first.  With nnz-balanced ranges every range
            # costs the same sweep time, so the bulk of the exchange starts after 1/nchunks of the item sweep.
            out.sort(key=lambda c: c[0] - c[1])
        return out

    def set_fused(self, flag):
        """Choose between the fused sweep+finalize launches and separate launches.  Each mode writes
        a fixed subset of the column-sum partial rows, so they are cleared on a switch."""
        self.fused = bool(flag)
        self.csT_part.zero_()
        self.csB_part.zero_()

    # ------------------------------------------------------------------------------------
    def _side_update(self, side, nrows, e_self, e_other, e_new, part, shp, rs_prev, fac, rs, cs_other, cs_part, gs, gf,
                     prior, top, add, store):
        """sweep one side and apply its closed-form updates.  Fused mode: the wavefront that swept a
        single-segment row finishes it (fp64 work overlaps other waves' gathers); split and empty rows
        are finished by a small follow-up launch over side.multi_rows."""
        ops, k, ld = self.ops, self.k, self.ld
        shp, fac = (shp, fac) if store else (None, None)
        if self.fused and side.nseg > 0:
            ops.sweep_finalize(side, e_self, e_other, part, e_new, shp, None, fac, rs, cs_other, cs_part[:gs],
                               prior, top, add, k, ld, rs_prev=rs_prev)
            nm = side.nmulti
            gm = max(1, min(gf, (nm + 3) // 4))
            ops.row_finalize(part, side.row_seg_ptr, nm, e_self, e_new, shp, None, fac, rs, cs_other,
                             cs_part[gs: gs + gm], prior, top, add, k, ld, row_list=side.multi_rows, rs_prev=rs_prev)
        else:
            ops.sweep(side, e_self, e_other, part, k, ld, grid_blocks=self.sweep_blocks)
            ops.row_finalize(part, side.row_seg_ptr, nrows, e_self, e_new, shp, None, fac, rs, cs_other,
                             cs_part[gs:], prior, top, add, k, ld, rs_prev=rs_prev)

    def iterate(self, store=True):
        """One CAVI iteration.  store=False skips writing the Gamma/Lambda shape and rate tables AND the mean
        tables Theta/Beta: all six are outputs (and llk inputs) only -- the iteration itself runs on the E
        tables, the scalar rates and the column sums, which are always kept current.  Callers pass store=True
        on the iterations whose state they read (checks, the last one)."""
        if self.dist:
            return self._iterate_scatter(store) if self.shard_mode == "scatter" else self._iterate_sharded(store)
        ops, hy, ld = self.ops, self.hy, self.ld
        # user side: phi-weighted gather over CSR rows, then the closed-form user updates
        self._side_update(self.users, self.nU, self.eT, self.eB, self.eT_next, self.part_u, self.Gamma_shp,
                          self.k_rte_prev, self.Theta, self.k_rte, self.csB, self.csT_part, self.gsu, self.gu,
                          hy.a, hy.k_shp, hy.add_k_rte, store)
        ops.colsum_reduce(self.csT_part, self.csT, ld)
        # item side: same kernel over CSC rows; still reads the OLD eT (double-buffered)
        self._keep_csB(store)
        self._side_update(self.items, self.nI, self.eB, self.eT, self.eB, self.part_i, self.Lambda_shp,
                          self.t_rte_prev, self.Beta, self.t_rte, self.csT, self.csB_part, self.gsi, self.gi,
                          hy.c, hy.t_shp, hy.add_t_rte, store)
        ops.colsum_reduce(self.csB_part, self.csB, ld)
        self.eT, self.eT_next = self.eT_next, self.eT
        self.niter_done += 1

    def iterate_many(self, n, store=True):
        """n iterations with the same `store` flag.  Scatter mode on RCCL with HPF_GRAPH=1: pairs of iterations (the E
        tables are double-buffered, so a pair restores every pointer) are replayed from a captured hipGraph -- one
        host call per pair instead of ~25 launches / collective calls / stream operations per iteration; anything
        that cannot be captured falls back to plain calls for good."""
        n = int(n)
        if n >= 4 and self._graph_capable():
            while self.niter_done < 2:      # warm up eagerly first (lazy allocations, RCCL channel setup)
                self.iterate(store)
                n -= 1
            g = self._pair_graph(bool(store))
            if g is not None:
                for _ in range(n // 2):
                    g.replay()
                self.niter_done += 2 * (n // 2)
                n -= 2 * (n // 2)
        for _ in range(n):
            self.iterate(store)

    def _graph_capable(self):
        if not (self.dist and self.shard_mode == "scatter" and self.device.type == "cuda"):
            return False
        if os.environ.get("HPF_GRAPH", "0") != "1" or getattr(self, "_graph_failed", False) or self.item_stream:
            return False        # (three-stream captures crash the ROCm 7.0 runtime: graphs only with HPF_ITEM_STREAM=0)
        if self.comm is not None:
            return True         # our own communicator: plain stream work, nothing of torch's polls it
        try:
            return self.dist.get_backend() == "nccl"     # gloo collectives run on the host: nothing to capture
        except Exception:   # noqa: BLE001  (stand-ins for torch.distributed in probes: assume capturable)
            return True

    def _pair_graph(self, store):
        graphs = self.__dict__.setdefault("_graphs", {})
        if store in graphs:
            return graphs[store]
        g = None
        done0 = self.niter_done
        try:
            self._sync_scatter_streams()
            torch.cuda.synchronize(self.device)
            # let torch's RCCL watchdog thread retire the eager collectives issued so far before events start being
            # recorded into a capture (it polls every 100 ms)
            import time
            time.sleep(float(os.environ.get("HPF_GRAPH_DRAIN_S", "0.35")))
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                self.iterate(store)
                self.iterate(store)
                self._sync_scatter_streams()    # side streams join the capturing stream
            torch.cuda.synchronize(self.device)
        except Exception as exc:   # noqa: BLE001
            g = None
            self._graph_failed = True
            self._graph_error = "%s: %s" % (type(exc).__name__, str(exc)[:200])
            torch.cuda.synchronize(self.device)
            self._sc_fresh = True
        self.niter_done = done0         # capture records the launches, it does not run them
        # all ranks replay, or none does: a rank that fell back would issue its collectives call by call while its peers
        # replay theirs -- same count, but nothing guarantees the same order against the control plane's
        if self.world > 1 and hasattr(self.dist, "ReduceOp"):
            ok = torch.tensor([1.0 if g is not None else 0.0], device=self.device)
            self.dist.all_reduce(ok, op=self.dist.ReduceOp.MIN)
            if float(ok.item()) < 1.0 and g is not None:
                g = None
                self._graph_failed = True
                self._graph_error = "capture failed on another rank"
        graphs[store] = g
        return g

    def _sharded_views(self):
        """Per item range: tensor views and column-sum partial slots, built once (the sharded loop is
        host-overhead sensitive).  Two partial layouts: `csp` for the standalone finalizer (flush), and
        `csp_lazy` for the prologue-fused sweep; each layout is written completely whenever it is used."""
        if self._chunk_views is not None:
            return self._chunk_views
        ops, ld = self.ops, self.ld
        f32 = dict(dtype=torch.float32, device=self.device)
        gm = max(1, min(self.gi, (self.items.nmulti + 3) // 4))
        lazy_rows = sum(ops.sweep_grid(v.nseg, self.sweep_blocks) for _, _, v, _ in self.item_chunks) + gm
        self.csB_part_lazy = torch.zeros((lazy_rows, ld), **f32)
        self._csp_multi = self.csB_part_lazy[lazy_rows - gm:]
        views, g0, l0 = [], self.gsi, 0
        for lo, hi, view, multi in self.item_chunks:
            g1 = g0 + ops.finalize_grid(hi - lo)
            l1 = l0 + ops.sweep_grid(view.nseg, self.sweep_blocks)
            views.append(dict(
                n=hi - lo, view=view, multi=multi, nmulti=int(multi.shape[0]), part=self.part_i[view.seg_lo:],
                acc=self.acc_i[lo:hi], eB=self.eB[lo:hi], shp=self.Lambda_shp[lo:hi], rsp=self.t_rte_prev[lo:hi],
                fac=self.Beta[lo:hi], rs=self.t_rte[lo:hi], csp=self.csB_part[g0:g1],
                csp_lazy=self.csB_part_lazy[l0:l1]))
            g0, l0 = g1, l1
        self._chunk_views = views
        return views

    def _iterate_sharded(self, store):
        """Users sharded over ranks.  Both sweeps read only last iteration's eT/eB, so the ITEM sweep goes
        first, in nnz-balanced item ranges: the all-reduce of one range (item accumulators, packed [rows,k])
        runs on the communication stream while the next range is swept and then while this rank does its
        whole user side; a k-float all-reduce of colsum(Theta) ends the iteration.

        The replicated item finalizer is DEFERRED: the reduced accumulators stay in acc_i and the next
        iteration's item sweep finishes each row in its prologue (hpf_hip_sweep_prefinalize_f32; split and
        empty rows by a small launch before it), on identical inputs on every rank, so replicas stay
        bit-identical.  flush_items() materialises the item tables when somebody needs them (llk, outputs)."""
        ops, hy, k, ld, dist = self.ops, self.hy, self.k, self.ld, self.dist
        views = self._sharded_views()
        lazy = self.item_pending
        if lazy:
            it = self.items
            ops.row_finalize(self.acc_i, None, it.nmulti, self.eB, self.eB, self.Lambda_shp if store else None,
                             None, self.Beta if store else None, self.t_rte, self.csT,
                             self._csp_multi,
                             hy.c, hy.t_shp, hy.add_t_rte, k, ld, row_list=it.multi_rows, part_ld=k,
                             rs_prev=self.t_rte_prev)
        xs = self._xstream()
        for c in views:
            # whole-row segments leave their accumulator straight in the packed buffer; only split rows
            # (and rows without local nonzeros: zeros) go through part[] + segsum
            if lazy and c["view"].nseg > 0:   # (a range without local nonzeros: its rows are all in multi_rows)
                ops.sweep_prefinalize(c["view"], self.eB, self.eT, c["part"], self.acc_i, k,
                                      self.Lambda_shp if store else None, None,
                                      self.Beta if store else None, self.t_rte, self.csT, c["csp_lazy"], hy.c, hy.t_shp,
                                      hy.add_t_rte, k, ld, rs_prev=self.t_rte_prev)
            elif not lazy and c["view"].nseg > 0:
                ops.sweep(c["view"], self.eB, self.eT, c["part"], k, ld, acc_rows=self.acc_i, acc_ld=k,
                          grid_blocks=self.sweep_blocks)
            if c["nmulti"] > 0:
                ops.segsum(self.part_i, self.items.row_seg_ptr, c["nmulti"], self.acc_i, ld, row_list=c["multi"],
                           acc_ld=k, acc_by_row=True)
            with self._exchange(xs):       # stream-ordered on the exchange stream (see _iterate_scatter)
                dist.all_reduce(c["acc"])
        if lazy:
            ops.colsum_reduce(self.csB_part_lazy, self.csB, ld)   # colsum(Beta) of the rows just finished
        self._keep_csB(store)
        self._side_update(self.users, self.nU, self.eT, self.eB, self.eT_next, self.part_u, self.Gamma_shp,
                          self.k_rte_prev, self.Theta, self.k_rte, self.csB, self.csT_part, self.gsu, self.gu,
                          hy.a, hy.k_shp, hy.add_k_rte, store)
        ops.colsum_reduce(self.csT_part, self.csT, ld)
        dist.all_reduce(self.csT)
        self._wait(self._mark(xs))
        self.item_pending = True
        self.eT, self.eT_next = self.eT_next, self.eT
        self.niter_done += 1
        if not self.lazy_items:
            self.flush_items(store)

    # ------------------------------------------------------------------------------------
    def _scatter_views(self):
        """Scatter mode, per item range: the range, this rank's slice of it and the exchange buffers.  The slices
        this rank owns of all ranges are concatenated (reduce-scatter outputs `acc_own_all`, new E rows `e_own_all`)
        so that ONE finalize launch covers them (hpf_hip_row_finalize_ranges_f32)."""
        if self._chunk_views is not None:
            return self._chunk_views
        ops, ld, k, W, r = self.ops, self.ld, self.k, self.world, self.rank
        f32 = dict(dtype=torch.float32, device=self.device)
        cuda = self.device.type == "cuda"
        total = sum((hi - lo) // W for lo, hi, _, _ in self.item_chunks)
        self.acc_own_all = torch.zeros((total, k), **f32)
        e_ld = k if self.ag_packed else ld            # row stride of the all-gather send buffer
        if self.gather_early:                         # [k numerators | base rate] rows, gathered in one collective
            e_ld = ops.gather_payload_ld(k)
            self.ag_recv_all = torch.ones((W * total, e_ld), **f32)
            self.shp_own_all = torch.zeros((total, ld), **f32)      # the shapes, between the finalizer's two halves
        self.e_own_all = torch.zeros((total, e_ld), **f32)
        views, t0, ranges = [], 0, []
        for lo, hi, view, multi in self.item_chunks:
            m = (hi - lo) // W
            o0 = lo + r * m
            n_real = max(0, min(m, self.nI - o0))
            if n_real > 0:
                ranges.append((n_real, t0, o0))
            views.append(dict(
                lo=lo, hi=hi, m=m, o0=o0, o1=o0 + m, n_real=n_real, view=view, multi=multi, nmulti=int(multi.shape[0]),
                part=self.part_i[view.seg_lo:], acc=self.acc_i[lo:hi], acc_own=self.acc_own_all[t0:t0 + m],
                e_own=self.e_own_all[t0:t0 + m],
                # views used every iteration (slicing costs host time in a loop that is ~40 % host-bound at 8 ranks)
                a2a_recv=torch.zeros((W * m, k), **f32) if self.rs_alltoall else None,
                eB_range=self.eB[lo:hi],
                # packed all-gather: received into the exchange buffer's rows of the range (free by then: the
                # reduce-scatter that read them precedes the all-gather on the exchange stream, and the next sweep of
                # the range -- their next writer -- waits for the all-gather)
                ag_recv=self.acc_i[lo:hi] if self.ag_packed else None,
                # dedicated events (re-recorded every iteration, waited for before the next record)
                sw_done=torch.cuda.Event() if cuda else None, ag_done=torch.cuda.Event() if cuda else None))
            t0 += m
        self._fin_ranges = ranges
        self._range_rows = [(c["lo"], c["hi"]) for c in views]       # in issue order (= slice order inside a rank's block)
        fin_rows = max(1, sum(n for n, _, _ in ranges))
        # (gather-early: the apply kernel streams ALL item rows -- twice the finalize grid keeps 32 waves per CU in flight)
        grid = ops.finalize_grid(fin_rows)
        if self.gather_early:      # (gx blocks for each rank's block of the gathered buffer: a multiple of the world size)
            grid = W * max(len(self.item_chunks), -(-2 * ops.finalize_grid(self.nI) // W))
        self.csB_part_sc = torch.zeros((grid, ld), **f32)
        self._csT_ready = torch.cuda.Event() if cuda else None
        self._sc_fresh = True
        self._chunk_views = views
        self._plan = self._make_plan(views)
        return views

    def _make_plan(self, views):
        """The native form of _iterate_scatter (hpf_hip_shard_iterate) over this model's tensors, or None: it needs a
        GPU, the fused user side, the two-stream schedule, and a way to run the collectives from C -- RCCL (backend
        "nccl": a communicator of our own, created here), or what a stand-in for torch.distributed brings
        (`native_collective`: a callback, tests with gloo ranks; `native_dry_run`: this rank alone, probes).  Every rank
        must end up with a plan, or none does (one collective of torch.distributed decides)."""
        dist = self.dist
        if (self.device.type != "cuda" or os.environ.get("HPF_NATIVE_SHARD", "1") != "1" or not self.fused
                or self.item_stream or self.rs_alltoall or self.users.nseg == 0 or len(views) > 8):
            return None
        plan, err = None, None
        try:
            from . import rccl, shard_native as sn
            coll = comm = comm_small = None
            dry = 0
            keep = []
            if hasattr(dist, "native_collective"):
                coll = sn.COLLECTIVE_FN(dist.native_collective(self))
                keep.append(coll)
            elif getattr(dist, "native_dry_run", False):
                dry = 1
                comm = dist.direct_comm(self.device, raw=True) if hasattr(dist, "direct_comm") else None
            elif dist.get_backend() == "nccl":
                comm = self.comm
                if comm is None:
                    key = (str(self.device), self.world, self.rank)
                    if key not in _DIRECT_COMMS:
                        _DIRECT_COMMS[key] = rccl.DirectComm(self.device, dist, self.rank, self.world)
                    comm = _DIRECT_COMMS[key]
                if not comm.self_check():
                    raise RuntimeError("the communicator's self-check failed")
                # a second communicator: the k-float all-reduces overtake the bulk collectives (HPF_CARRIED_ONE_COMM=1:
                # all on one -- correct either way, slower if RCCL orders a communicator's operations across streams)
                if self.gather_carried and os.environ.get("HPF_CARRIED_ONE_COMM", "0") != "1":
                    key = (str(self.device), self.world, self.rank, "small")
                    if key not in _DIRECT_COMMS:
                        _DIRECT_COMMS[key] = rccl.DirectComm(self.device, dist, self.rank, self.world)
                    comm_small = _DIRECT_COMMS[key]
                    if not comm_small.self_check():
                        raise RuntimeError("the second communicator's self-check failed")
            else:
                return None
            d = sn.ShardDesc()
            hy, u, it = self.hy, self.users, self.items
            d.world, d.rank, d.k, d.ld, d.nU, d.nI = self.world, self.rank, self.k, self.ld, self.nU, self.nI
            d.u_segs, d.u_nseg, d.u_idx, d.u_y = u.segs.data_ptr(), u.nseg, u.idx.data_ptr(), u.y.data_ptr()
            d.u_row_seg_ptr, d.u_nmulti = u.row_seg_ptr.data_ptr(), u.nmulti
            d.u_multi_rows = u.multi_rows.data_ptr() if u.nmulti else None
            d.i_segs, d.i_idx, d.i_y, d.i_row_seg_ptr = (it.segs.data_ptr(), it.idx.data_ptr(), it.y.data_ptr(),
                                                         it.row_seg_ptr.data_ptr())
            d.nranges = len(views)
            for j, c in enumerate(views):
                r = d.ranges[j]
                r.lo, r.hi, r.seg_lo, r.nseg = c["lo"], c["hi"], c["view"].seg_lo, c["view"].nseg
                r.nmulti, r.short_rows = c["nmulti"], int(c["view"].short_rows)
                r.multi_rows = c["multi"].data_ptr() if c["nmulti"] else None
            for n in ("eB", "part_u", "part_i", "Gamma_shp", "Theta", "k_rte", "k_rte_prev", "Lambda_shp", "Beta", "t_rte",
                      "t_rte_prev", "csT", "csB", "csB_used", "csT_part", "acc_i"):
                setattr(d, n, getattr(self, n).data_ptr())
            d.csT_part_rows, d.user_sweep_grid = int(self.csT_part.shape[0]), self.gsu
            d.user_multi_grid = max(1, min(self.gu, (u.nmulti + 3) // 4))
            d.csB_part, d.csB_part_rows = self.csB_part_sc.data_ptr(), int(self.csB_part_sc.shape[0])
            d.acc_own, d.e_own = self.acc_own_all.data_ptr(), self.e_own_all.data_ptr()
            d.e_own_ld, d.item_sweep_grid = int(self.e_own_all.shape[1]), int(self.item_sweep_blocks)
            d.ag_recv = self.acc_i.data_ptr() if self.ag_packed else None
            if self.gather_early:
                d.schedule, d.ag_recv, d.shp_own = 1, self.ag_recv_all.data_ptr(), self.shp_own_all.data_ptr()
            if self.gather_carried:
                d.schedule = 2
                d.comm_small = comm_small.handle if comm_small is not None else None
                if getattr(self, "_ss", None) is None:     # colsum(Beta): reduced + summed under the last item sweep
                    self._ss = _side_stream(self.device, "small", -1)
                d.sstream = self._ss.cuda_stream
            d.a, d.k_shp, d.add_k_rte = float(hy.a), float(hy.k_shp), float(hy.add_k_rte)
            d.c, d.t_shp, d.add_t_rte = float(hy.c), float(hy.t_shp), float(hy.add_t_rte)
            d.comm = comm.handle if comm is not None else None
            if coll is not None:
                d.coll = coll
            d.xstream = self._xstream().cuda_stream
            d.dry_run = dry
            if dry:      # (probes: hold the streams for the time real links would take, at an assumed bus bandwidth)
                d.dry_run_busbw_GBps = float(getattr(dist, "native_dry_run_busbw", 0.0))
                d.dry_run_latency_us = float(getattr(dist, "native_dry_run_latency_us", 0.0))
                d.dry_run_footprint_blocks = int(getattr(dist, "native_dry_run_footprint_blocks", 0))
            plan = sn.ShardPlan(d, keep=keep + [comm, comm_small, views])
        except Exception as exc:   # noqa: BLE001
            plan, err = None, "%s: %s" % (type(exc).__name__, str(exc)[:200])
        # all or none (a rank that issued its collectives through another communicator than its peers would hang them)
        if self.world > 1 and hasattr(dist, "all_reduce") and not getattr(dist, "native_dry_run", False):
            ok = torch.tensor([1.0 if plan is not None else 0.0], device=self.device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if float(ok.item()) < 1.0 and plan is not None:
                plan.close()
                plan, err = None, "another rank could not create its plan"
        self.native_error = err
        if plan is not None:
            NATIVE_PLANS_CREATED[0] += 1
        return plan

    def _iterate_scatter(self, store):
        """Users sharded over ranks, item FINALIZER sharded too.  Per item range (fewest rows first):
        sweep the local CSC slice into the packed exchange buffer, then a REDUCE-SCATTER on the exchange stream
        leaves each rank with the global statistics of its 1/N slice of the range; the user side runs under the
        exchange.  Everything after it is ONE in-order chain on the exchange stream: the k-float all-reduce of
        colsum(Theta), the finalizer of this rank's slices (ONE dense launch over the slices of all ranges: 1/N of
        the fp64 work and of the table stores), the ALL-GATHERS of the new E rows -- straight into the replicated E
        table, each waited for only by the next iteration's sweep of that range -- and, behind them (the next user
        side is its only reader), the k-float all-reduce of this rank's partial colsum(Beta).  Same bytes on the
        wire as the all-reduce form.  Lambda_shp / Beta / t_rte are current on the owning rank only; flush_items()
        gathers them.

        Streams: compute (sweeps, user finalizer), exchange (collectives + item finalizer; collectives are issued
        with async_op=False inside the exchange stream's context, i.e. ordered on THAT stream) and, with
        HPF_ITEM_STREAM=1, a third one for the item sweeps.  A cross-stream dependency costs ~15-20 us on the
        waiting stream (tools/handover_probe.py), so the critical cycle user side -> finalizer -> all-gather ->
        next sweep crosses streams exactly twice; all other waits are for work that finished long before."""
        ops, hy, k, ld, dist = self.ops, self.hy, self.k, self.ld, self.dist
        views = self._scatter_views()
        if self._plan is not None and self.native and self.fused:       # the same schedule, issued by one C call
            if not self._last_native:
                self._sync_scatter_streams()      # (switching forms: the other one's exchanges first)
            self._last_native = True
            self._plan.iterate(self.eT, self.eT_next, store, torch.cuda.current_stream(self.device).cuda_stream)
            self.rte_factored = True         # (_keep_csB: the C call copies colsum(Beta) on storing iterations)
            self._sc_fresh = False
            self._tables_split = True
            self.eT, self.eT_next = self.eT_next, self.eT
            self.niter_done += 1
            return
        if self._last_native:
            self._sync_scatter_streams()
        self._last_native = False
        if self.gather_early:
            return self._iterate_gather_early(store)
        xs = self._xstream()
        cuda = xs is not None
        cs = torch.cuda.current_stream(self.device) if cuda else None
        ist = (self._istream() if self.item_stream else cs) if cuda else None
        fresh = self._sc_fresh
        if cuda and fresh:
            for st in (ist, xs):            # first iteration after load_state: order after whatever the caller queued
                if st is not cs:
                    st.wait_event(self._mark(cs))
        on = (lambda st: torch.cuda.stream(st)) if cuda else (lambda st: contextlib.nullcontext())
        for c in views:
            if cuda and not fresh:
                ist.wait_event(c["ag_done"])   # this range's E rows from the previous iteration's finalizers
            with on(ist):
                if c["view"].nseg > 0:
                    ops.sweep(c["view"], self.eB, self.eT, c["part"], k, ld, acc_rows=self.acc_i, acc_ld=k,
                              grid_blocks=self.item_sweep_blocks)
                if c["nmulti"] > 0:
                    ops.segsum(self.part_i, self.items.row_seg_ptr, c["nmulti"], self.acc_i, ld, row_list=c["multi"],
                               acc_ld=k, acc_by_row=True)
            if cuda:
                c["sw_done"].record(ist)
                xs.wait_event(c["sw_done"])
            with on(xs):
                if self.rs_alltoall:
                    # direct form: slice j of the range goes straight to rank j (one hop over every xGMI link at
                    # once), which then adds up the N slices it received, in rank order
                    dist.all_to_all_single(c["a2a_recv"], c["acc"])
                    torch.sum(c["a2a_recv"].view(self.world, c["m"], k), dim=0, out=c["acc_own"])
                elif self.comm is not None:
                    self.comm.reduce_scatter(c["acc_own"], c["acc"])
                else:
                    dist.reduce_scatter_tensor(c["acc_own"], c["acc"])
        if cuda and not fresh:
            if ist is not cs:                  # (same stream: the sweeps above waited already; the last range's
                for c in views:                # all-gather also orders the colsum(Beta) all-reduce issued ahead of it)
                    cs.wait_event(c["ag_done"])
        self._keep_csB(store)
        self._side_update(self.users, self.nU, self.eT, self.eB, self.eT_next, self.part_u, self.Gamma_shp,
                          self.k_rte_prev, self.Theta, self.k_rte, self.csB, self.csT_part, self.gsu, self.gu,
                          hy.a, hy.k_shp, hy.add_k_rte, store)
        ops.colsum_reduce(self.csT_part, self.csT, ld)
        if cuda:
            self._csT_ready.record(cs)
            xs.wait_event(self._csT_ready)
        ar = self.comm.all_reduce if self.comm is not None else dist.all_reduce
        with on(xs):
            ar(self.csT)
            if self._fin_ranges:
                ops.row_finalize_ranges(self.acc_own_all, self._fin_ranges, self.eB, self.e_own_all,
                                        self.Lambda_shp if store else None, None, self.Beta if store else None,
                                        self.t_rte, self.csT, self.csB_part_sc, hy.c, hy.t_shp, hy.add_t_rte, k, ld, k,
                                        rs_prev=self.t_rte_prev, e_new_ld=int(self.e_own_all.shape[1]))
            for j, c in enumerate(views):
                if j == len(views) - 1:
                    # colsum(Beta) -- read by the next USER side only -- goes ahead of the last all-gather (which the
                    # user side waits for anyway) and behind the first one (which the next item sweep is waiting for)
                    ops.colsum_reduce(self.csB_part_sc, self.csB, ld)      # this rank's partial colsum(Beta) ...
                    ar(self.csB)                                           # ... summed over ranks
                ag_out = c["ag_recv"] if self.ag_packed else c["eB_range"]
                if self.comm is not None:
                    self.comm.all_gather(ag_out, c["e_own"])
                else:
                    dist.all_gather_into_tensor(ag_out, c["e_own"])
                if self.ag_packed:
                    ops.unpack_rows(ag_out, c["eB_range"], c["hi"] - c["lo"], k, ld)
                if cuda:
                    c["ag_done"].record(xs)
        self._sc_fresh = False
        self._tables_split = True
        self.eT, self.eT_next = self.eT_next, self.eT
        self.niter_done += 1

    def _iterate_gather_early(self, store):
        """The gather-early schedule issued call by call, IN ORDER on the current stream (gloo / stand-in runs and the
        fallback; the overlapped form is hpf_hip_shard_iterate with HPF_SCHEDULE_GATHER_EARLY): item sweeps +
        reduce-scatter per range; the shape half of the finalizer for this rank's slices; one all-gather of the
        [numerators | base rate] rows; the user side; colsum(Theta) summed over ranks; the rates applied to all items
        locally; colsum(Beta) summed over ranks."""
        ops, hy, k, ld, dist = self.ops, self.hy, self.k, self.ld, self.dist
        views = self._chunk_views
        self._sync_scatter_streams()
        for c in views:
            if c["view"].nseg > 0:
                ops.sweep(c["view"], self.eB, self.eT, c["part"], k, ld, acc_rows=self.acc_i, acc_ld=k,
                          grid_blocks=self.item_sweep_blocks)
            if c["nmulti"] > 0:
                ops.segsum(self.part_i, self.items.row_seg_ptr, c["nmulti"], self.acc_i, ld, row_list=c["multi"],
                           acc_ld=k, acc_by_row=True)
            dist.reduce_scatter_tensor(c["acc_own"], c["acc"])
        if self._fin_ranges:
            ops.item_shape_rows(self.acc_own_all, self._fin_ranges, self.eB, self.shp_own_all, self.e_own_all, self.t_rte,
                                hy.c, hy.t_shp, k, ld, rs_prev=self.t_rte_prev)
        dist.all_gather_into_tensor(self.ag_recv_all.view(-1), self.e_own_all.view(-1))
        self._keep_csB(store)
        self._side_update(self.users, self.nU, self.eT, self.eB, self.eT_next, self.part_u, self.Gamma_shp,
                          self.k_rte_prev, self.Theta, self.k_rte, self.csB, self.csT_part, self.gsu, self.gu,
                          hy.a, hy.k_shp, hy.add_k_rte, store)
        ops.colsum_reduce(self.csT_part, self.csT, ld)
        dist.all_reduce(self.csT)
        ops.item_apply_rows(self.ag_recv_all, self.shp_own_all, self.eB, self.Lambda_shp if store else None,
                            self.Beta if store else None, self.t_rte, self.csT, self.csB_part_sc, hy.add_t_rte, k, ld,
                            self.rank, self.world, self.nI, self._range_rows)
        ops.colsum_reduce(self.csB_part_sc, self.csB, ld)
        dist.all_reduce(self.csB)
        self._sc_fresh = True          # (in order on one stream: nothing stays in flight)
        self._tables_split = True
        self.eT, self.eT_next = self.eT_next, self.eT
        self.niter_done += 1

    # exchange-stream plumbing (CPU tensors / no GPU: everything degenerates to plain in-order calls)
    def _xstream(self):
        if self.device.type != "cuda":
            return None
        if getattr(self, "_xs", None) is None:
            # HIGH priority: the exchange chain is latency-critical, and a priority stream is served by another hardware
            # queue than the normal-priority compute stream.  ROCm multiplexes streams of one priority over 4 hardware
            # queues in creation order; when the exchange stream landed on the compute stream's queue, a collective
            # that waits for the links held back the sweeps queued behind it (seen with tools/shard_probe.py
            # PROBE_BUSBW=...: the second model of a process ran 0.3 ms slower per iteration than the first)
            self._xs = _side_stream(self.device, "exchange", -1)
        return self._xs

    def _istream(self):
        if self.device.type != "cuda":
            return None
        if getattr(self, "_is", None) is None:
            self._is = _side_stream(self.device, "item")
        return self._is

    def _event(self):
        """Events are re-used round-robin (a re-recorded event is only ever waited for after its latest record)."""
        pool = getattr(self, "_ev_pool", None)
        if pool is None:
            pool = self._ev_pool = [torch.cuda.Event() for _ in range(32)]
            self._ev_next = -1
        self._ev_next = (self._ev_next + 1) % 32
        return pool[self._ev_next]

    def _exchange(self, xs):
        """Context: what is issued inside runs on the exchange stream, after everything issued so far on the
        compute stream."""
        if xs is None:
            return contextlib.nullcontext()
        ev = self._event()
        ev.record()
        xs.wait_event(ev)
        return torch.cuda.stream(xs)

    def _mark(self, xs):
        """Event at the current end of the exchange stream (None without one)."""
        if xs is None:
            return None
        ev = self._event()
        ev.record(xs)
        return ev

    def _wait(self, ev):
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)

    def _sync_scatter_streams(self):
        """Scatter mode: the current stream waits for the exchanges still in flight on the side streams."""
        views = self._chunk_views or []
        if views and not getattr(self, "_sc_fresh", True):
            if self._last_native:
                self._plan.join(torch.cuda.current_stream(self.device).cuda_stream)
            else:
                for c in views:
                    self._wait(c["ag_done"])
            self._sc_fresh = True        # the next iteration re-synchronises its side streams with this one

    def _sync_scatter(self):
        """Scatter mode: wait for the outstanding exchanges and gather the per-owner item tables."""
        self._sync_scatter_streams()
        if self._tables_split:
            for c in self._scatter_views():
                for tab in (self.Lambda_shp, self.Beta, self.t_rte, self.t_rte_prev):
                    self.dist.all_gather_into_tensor(tab[c["lo"]: c["hi"]], tab[c["o0"]: c["o1"]].clone())
            self._tables_split = False

    def flush_items(self, store=True):
        """Sharded path: make the item tables current on this rank.  All-reduce mode: apply the deferred item
        finalizer (Lambda_shp, Lambda_rte, Beta, t_rte, eB, colsum Beta); scatter mode: wait + gather."""
        if self.dist and self.shard_mode == "scatter":
            return self._sync_scatter()
        if not (self.dist and self.item_pending):
            return
        ops, hy, k, ld = self.ops, self.hy, self.k, self.ld
        for c in self._sharded_views():
            ops.row_finalize(c["acc"], None, c["n"], c["eB"], c["eB"], c["shp"] if store else None, None,
                             c["fac"] if store else None, c["rs"], self.csT,
                             c["csp"], hy.c, hy.t_shp, hy.add_t_rte, k, ld, part_ld=k, rs_prev=c["rsp"])
        ops.colsum_reduce(self.csB_part, self.csB, ld)
        self.item_pending = False

    # ------------------------------------------------------------------------------------
    def llk_terms(self, full_llk=False):
        """Global (all-reduced) float64 [sum y*log(yhat)(-lgamma), sum sq.err, sum yhat, nnz] over the training nonzeros."""
        self.flush_items()
        t = self.ops.llk_sweep(self.users, self.Theta, self.Beta, self.k, self.ld, full_llk)
        out = torch.cat([t.to(torch.float64), torch.tensor([float(self.nnz)], dtype=torch.float64,
                                                           device=t.device)])
        if self.dist:
            self.dist.all_reduce(out)
        return out.cpu().numpy()

    def pair_llk_terms(self, ix_u, ix_i, y, full_llk=False):
        """Same terms over caller-listed pairs (validation set); ix_u must be local to this shard."""
        self.flush_items()
        t = self.ops.pair_llk(self.Theta, self.Beta, ix_u, ix_i, y, self.k, self.ld, full_llk)
        out = torch.cat([t.to(torch.float64), torch.tensor([float(ix_u.shape[0])], dtype=torch.float64,
                                                           device=t.device)])
        if self.dist:
            self.dist.all_reduce(out)
        return out.cpu().numpy()

    def colsum_dot(self):
        """(sum_u Theta) . (sum_i Beta) in float32, the subtrahend of the train llk (PXI:78)."""
        self.flush_items()
        if self.niter_done == 0:
            self.ops.colsum(self.Theta, self.nU, self.ld, self.cs_scratch)
            self.ops.colsum_reduce(self.cs_scratch, self.csT, self.ld)
            if self.dist:
                self.dist.all_reduce(self.csT)
        a = self.csT[: self.k].cpu().numpy()
        b = self.csB[: self.k].cpu().numpy()
        return np.dot(a, b)

    # ------------------------------------------------------------------------------------
    def _keep_csB(self, store):
        """Remember the colsum(Beta) the user side of this iteration used (Gamma_rte's rank-1 term) before
        the item side replaces it."""
        if store:
            self.csB_used.copy_(self.csB)
        self.rte_factored = True

    def materialize_rates(self):
        """Expand the rank-1 rate tables into Gamma_rte / Lambda_rte (PXI:236, PXI:255): the kernels keep
        only the old scalar rate per row and the column sums; same fp32 operations as the table form."""
        self.flush_items()
        if not self.rte_factored:
            return
        k = self.k
        self.Gamma_rte[:, :k] = (float(self.hy.k_shp) / self.k_rte_prev)[:, None] + self.csB_used[None, :k]
        nI = self.nI
        self.Lambda_rte[:nI, :k] = (float(self.hy.t_shp) / self.t_rte_prev[:nI])[:, None] + self.csT[None, :k]
        self.rte_factored = False

    def fetch(self, name, out=None):
        """Unpadded host copy of one state array (this rank's rows); into `out` -- a C-contiguous float32 host array
        of that shape -- when given (one device-to-host copy, no second pass on the host)."""
        self.flush_items()
        if name in ("Gamma_rte", "Lambda_rte"):
            self.materialize_rates()
        t = getattr(self, name)
        if name in ("Lambda_shp", "Lambda_rte", "Beta", "t_rte", "eB"):
            t = t[: self.nI]          # scatter mode keeps pad rows at the end of the item tables
        t = t.reshape(-1, 1) if t.dim() == 1 else t[:, : self.k]
        if out is not None and out.flags.c_contiguous and out.flags.writeable and out.dtype == np.float32 \
                and tuple(out.shape) == tuple(t.shape):
            torch.from_numpy(out).copy_(t)
            return out
        host = t.contiguous().cpu().numpy()
        if out is not None:
            out[...] = host
            return out
        return host
