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

Multi-GPU (one process per GPU, torch.distributed as the control plane): users are sharded in contiguous nnz-balanced
ranges, the item E table is replicated, and the exchange per iteration is the sum of the item accumulators (nI*k floats,
overlapped with the user side) plus two k-float column sums -- hpfrec_amd/shard.py (ShardedMixin), mixed in below.

The kernels are reached through an `ops` object (hpfrec_amd.ops_hip.HipOps).  There is no CPU
implementation in this package.
"""
import os

import numpy as np
import torch

from . import _lib, layout
from .shard import NATIVE_PLANS_CREATED, ShardedMixin, _DIRECT_COMMS  # noqa: F401  (re-exported: tests, bench, tools)


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


class FullBatchCavi(ShardedMixin):
    """Device-resident state + one-iteration step for (a shard of) the HPF model."""

    def __init__(self, ops, device, ix_u, ix_i, y, nU, nI, hyper, seg_cap=None, sides=None):
        """ix_u/ix_i/y: COO triplets of THIS rank (user ids local to the shard), torch tensors.  sides: the (users, items,
        u_sorted) layouts of another model over the SAME triplets (`model.sides()`), taken over instead of being rebuilt --
        they are read-only (bench.py builds up to seven models of one matrix in a row; the triplets may then be None)."""
        self.ops = ops
        self.device = torch.device(device)
        self.hy = hyper
        self.k = hyper.k
        self.ld = _lib.ld_for_k(self.k)
        self.nU, self.nI = int(nU), int(nI)
        dev = self.device
        if sides is not None:
            self.users, self.items, self.u_sorted = sides
            assert self.users.nrows == self.nU and self.items.nrows == self.nI
        else:
            self.users, self.items, self.u_sorted = layout.build_sides(ix_u.to(dev), ix_i.to(dev), y.to(dev),
                                                                       self.nU, self.nI, seg_cap)
        self.nnz = self.users.nnz
        self.dist = _dist()
        # sweep grid of THIS model (the op set is shared): sharded launches cover short item ranges and want fewer,
        # fatter blocks (16 per CU costs 3 % at N=8, gains 1 % at N=1).  THREE workgroups per CU for the user sweep, not
        # the four that fill every wave slot its 104 VGPRs allow: the user sweep is the kernel the exchange hides under,
        # and the exchange stream's kernels (or a collective's) must become RESIDENT beside it
        # (profiles/r03_shard_probe_collective_footprint.txt)
        self.sweep_blocks = ops.sweep_blocks
        if self.dist and "HPF_SWEEP_BPC" not in os.environ and hasattr(ops, "cu_count"):
            # (the direct exchange's kernels are light -- no collective library's footprint to leave room for: 4 measured
            #  0.588 vs 0.605 ms per 8-rank iteration, profiles/r04_shard_probe_direct.txt; the RCCL schedules keep 3)
            from .shard import requested_schedule
            direct = requested_schedule() in ("auto", "direct") and self.device.type == "cuda" and \
                os.environ.get("HPF_NATIVE_SHARD", "1") == "1"
            self.sweep_blocks = max(1, ops.cu_count) * int(os.environ.get("HPF_SHARD_SWEEP_BPC", "4" if direct else "3"))
        # the sharded item pass is many short rows: more, smaller blocks even out its tail (tools/sweep_micro.py:
        # 203 us at 32 blocks per CU vs 220 at 8 for rank 0 of 8 at C3)
        self.item_sweep_blocks = max(1, getattr(ops, "cu_count", 1)) * int(os.environ.get("HPF_ITEM_SWEEP_BPC", "32")) \
            if hasattr(ops, "cu_count") else self.sweep_blocks
        ld = self.ld
        f32 = dict(dtype=torch.float32, device=dev)
        z = lambda n: torch.zeros((n, ld), **f32)
        self._init_sharded(ops)         # world / rank / schedule / item ranges (shard.py); nI_alloc: items + pad rows
        nIa = self.nI_alloc
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
        self.csB_part = torch.zeros((self.gsi + self.gi, ld), **f32)
        self.cs_scratch = torch.zeros((max(self.gu, self.gi), ld), **f32)  # for whole-table column sums
        self.csB = torch.zeros(ld, **f32)
        self.csT = torch.zeros(ld, **f32)
        self.niter_done = 0
        # HPF_COLSUM_ORDER=reference (single GPU): Theta.sum(axis=0) / Beta.sum(axis=0) in numpy's own order --
        # float32, row after row (PXI:236,255) -- instead of the sweeps' per-block partials summed in double: the reference's
        # sums bit for bit, at the price of a chain of nrows dependent adds per iteration and side
        self.ref_sums = os.environ.get("HPF_COLSUM_ORDER", "tree") == "reference"
        if self.ref_sums and self.dist:
            raise ValueError("HPF_COLSUM_ORDER=reference is a single-GPU mode")

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
        self._tables_split = False
        if self.dist:
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
        if nU > 0:
            ops.uniform_rows(draws[0], self.Gamma_rte, nU, k, ld, hy.a_prime, 0.01)
            ops.uniform_rows(draws[2], self.Gamma_shp, nU, k, ld, hy.a_prime, 0.01, den=self.Gamma_rte, ratio=self.Theta)
        ops.uniform_rows(draws[1], self.Lambda_rte, nI, k, ld, hy.c_prime, 0.01)
        ops.uniform_rows(draws[3], self.Lambda_shp, nI, k, ld, hy.c_prime, 0.01, den=self.Lambda_rte, ratio=self.Beta)
        self.k_rte.fill_(float(hy.b_prime))
        self.t_rte[: self.nI].fill_(float(hy.d_prime))
        self._tables_split = False
        if self.dist:
            self._sync_scatter()
        self.rte_factored = False
        self.refresh_expectations()

    def sides(self):
        """The sparse layouts of this model's triplets, for another model of the same matrix (constructor's `sides`)."""
        return self.users, self.items, self.u_sorted

    def refresh_expectations(self):
        ops, k, ld = self.ops, self.k, self.ld
        if self.nU > 0:
            ops.expect(self.Gamma_shp, self.Gamma_rte, self.eT, self.nU, k, ld)
        ops.expect(self.Lambda_shp, self.Lambda_rte, self.eB, self.nI, k, ld)
        if self.ref_sums:
            ops.colsum_sequential(self.Beta, self.nI, ld, self.csB)
            return
        ops.colsum(self.Beta, self.nI, ld, self.cs_scratch)
        ops.colsum_reduce(self.cs_scratch, self.csB, ld)

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
        if nrows == 0:          # (a rank without users: its column-sum partial rows stay zero)
            return
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
            return self._iterate_scatter(store)          # (shard.py)
        ops, hy, ld = self.ops, self.hy, self.ld
        store = store or self.ref_sums       # (the reference-order sums walk the stored mean tables)
        # user side: phi-weighted gather over CSR rows, then the closed-form user updates
        self._side_update(self.users, self.nU, self.eT, self.eB, self.eT_next, self.part_u, self.Gamma_shp,
                          self.k_rte_prev, self.Theta, self.k_rte, self.csB, self.csT_part, self.gsu, self.gu,
                          hy.a, hy.k_shp, hy.add_k_rte, store)
        if self.ref_sums:
            ops.colsum_sequential(self.Theta, self.nU, ld, self.csT)
        else:
            ops.colsum_reduce(self.csT_part, self.csT, ld)
        # item side: same kernel over CSC rows; still reads the OLD eT (double-buffered)
        self._keep_csB(store)
        self._side_update(self.items, self.nI, self.eB, self.eT, self.eB, self.part_i, self.Lambda_shp,
                          self.t_rte_prev, self.Beta, self.t_rte, self.csT, self.csB_part, self.gsi, self.gi,
                          hy.c, hy.t_shp, hy.add_t_rte, store)
        if self.ref_sums:
            ops.colsum_sequential(self.Beta, self.nI, ld, self.csB)
        else:
            ops.colsum_reduce(self.csB_part, self.csB, ld)
        self.eT, self.eT_next = self.eT_next, self.eT
        self.niter_done += 1

    def iterate_many(self, n, store=True):
        """n iterations with the same `store` flag."""
        for _ in range(int(n)):
            self.iterate(store)

    # ------------------------------------------------------------------------------------
    def llk_terms(self, full_llk=False):
        """Global (all-reduced) float64 [sum y*log(yhat)(-lgamma), sum sq.err, sum yhat, nnz] over the training nonzeros."""
        self.flush_items()
        if self.nU > 0:
            t = self.ops.llk_sweep(self.users, self.Theta, self.Beta, self.k, self.ld, full_llk)
        else:                   # (a rank without users contributes nothing)
            t = torch.zeros(3, dtype=torch.float64, device=self.device)
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
            if self.nU > 0 and self.ref_sums:
                self.ops.colsum_sequential(self.Theta, self.nU, self.ld, self.csT)
            elif self.nU > 0:
                self.ops.colsum(self.Theta, self.nU, self.ld, self.cs_scratch)
                self.ops.colsum_reduce(self.cs_scratch, self.csT, self.ld)
            else:
                self.csT.zero_()
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
