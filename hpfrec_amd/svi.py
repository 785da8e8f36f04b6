"""Stochastic (mini-batch) paths on the device: the SVI epochs of fit_hpf
(/root/reference/hpfrec/cython_loops.pxi:262-377, "PXI"), the Cython partial_fit (PXI:423-473) and the
single-user fold-in calc_user_factors (PXI:476-520) -- the "next" rows f1/f4 of SURVEY.md section 8.

Division of labour in this round:
  * O(batch_nnz * k): phi and both shape accumulations -> the same `sweep_kernel` as the full-batch
    path, run over the batch's rows from both sides (two passes, no atomics, deterministic), plus the
    row-list forms of `expect_kernel` / `segsum_kernel` (update_phi_csr PXI:666-692 always
    max-subtracts; our E rows are rescaled per row in every mode);
  * O((nU+nI) * k) per batch: the reference recomputes whole tables with numpy statements every batch
    (PXI:300,318,322 ...); here those statements are three HIP row kernels (svi_shape_rows,
    svi_refresh, svi_rate_rows; include/hpf_hip.h) issued in the reference's order.
torch is used for index plumbing only (grouping a batch by row, aligning accumulators with row lists).
No numerics run on the host; shuffles and seeds use numpy's generators so batches are the reference's.
"""
import os

import numpy as np
import torch

from . import _lib, layout

_NAMES = ("Gamma_shp", "Gamma_rte", "Lambda_shp", "Lambda_rte", "Theta", "Beta")
SVI_TIMINGS = {}      # HPF_TIMING=1: {"epochs": n, "seconds": wall time of the last fit's epoch loop}


class BatchSide:
    """Segments over the rows touched by a batch (same fields the sweep launcher reads from SparseSide)."""

    def __init__(self, rows, cols, y, seg_cap=layout.SEG_CAP, grouped=False):
        """rows/cols: int64 device tensors of a COO batch.  grouped=True: the triplets already come grouped by
        ascending `rows` (what gather_rows returns for a sorted row list) -- no sort needed; otherwise they are
        grouped here with a stable sort."""
        if grouped:
            r_s, c_s, y_s = rows, cols, y
        else:
            order = torch.argsort(rows, stable=True)
            r_s, c_s, y_s = rows[order], cols[order], y[order]
        self.rows, counts = torch.unique_consecutive(r_s, return_counts=True)   # rows present, ascending
        self.idx = c_s.to(torch.int32).contiguous()
        self.y = y_s.to(torch.float32).contiguous()
        indptr = torch.zeros(self.rows.shape[0] + 1, dtype=torch.int64, device=rows.device)
        torch.cumsum(counts, 0, out=indptr[1:])
        segs, self.row_seg_ptr = layout.build_segments(indptr, seg_cap)
        if segs.shape[0] > 0:
            local = segs[:, 1] >> 32
            meta = segs[:, 1] & 0xFFFFFFFF      # length | whole-row flag
            segs = torch.stack([segs[:, 0], meta | (self.rows[local] << 32)], dim=1).contiguous()
        self.segs = segs
        self.nseg = int(segs.shape[0])
        self.nrows = int(self.rows.shape[0])
        self.short_rows = 1 if (self.nseg > 0 and
                                                   int(self.y.shape[0]) / self.nseg < layout.SHORT_ROW_NNZ) else 0
        nseg_row = self.row_seg_ptr[1:] - self.row_seg_ptr[:-1]
        self.multi_local = torch.nonzero(nseg_row > 1).reshape(-1)              # rows cut into several segments
        self.nmulti = int(self.multi_local.shape[0])

    @classmethod
    def from_parts(cls, rows, idx, y, row_seg_ptr, segs, multi_local, nseg, nmulti):
        """The same object from ready-made pieces (no device work, no host synchronisation)."""
        self = cls.__new__(cls)
        self.rows, self.idx, self.y, self.row_seg_ptr, self.segs, self.multi_local = rows, idx, y, row_seg_ptr, segs, multi_local
        self.nseg, self.nmulti, self.nrows = int(nseg), int(nmulti), int(rows.shape[0])
        self.short_rows = 1 if (self.nseg > 0 and
                                                   int(y.shape[0]) / self.nseg < layout.SHORT_ROW_NNZ) else 0
        return self

    def tensors(self):
        return [self.rows, self.idx, self.y, self.row_seg_ptr, self.segs, self.multi_local]


class PinnedStaging:
    """Page-locked host memory for the small int64 arrays a batch sends to the device (row lists, offsets, segment
    descriptors), allocated ONCE: a copy from pageable memory makes the host wait for everything queued on the stream
    before it, and pinning per copy costs milliseconds on this platform.  All arrays of a batch go up in ONE
    asynchronous copy.  A few slots are used round-robin; a slot is re-used only after its copy has completed."""

    def __init__(self, device, words=1 << 20, slots=4):
        self.device = device
        self.bufs = [torch.empty(words, dtype=torch.int64, pin_memory=True) for _ in range(slots)]
        self.events = [None] * slots
        self.turn = -1

    def upload(self, arrays):
        """int64 numpy arrays -> device tensors of the same shapes (one transfer)."""
        arrays = [np.ascontiguousarray(a, dtype=np.int64) for a in arrays]
        total = sum(int(a.size) for a in arrays)
        self.turn = (self.turn + 1) % len(self.bufs)
        slot, buf = self.turn, self.bufs[self.turn]
        if total > buf.shape[0]:                                  # (does not fit: plain synchronous copies)
            return [torch.from_numpy(a).to(self.device) for a in arrays]
        if self.events[slot] is not None:
            self.events[slot].synchronize()
        host, at = buf.numpy(), 0
        for a in arrays:
            host[at: at + a.size] = a.reshape(-1)
            at += a.size
        dev_all = buf[:total].to(self.device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.events[slot] = ev
        out, at = [], 0
        for a in arrays:
            out.append(dev_all[at: at + a.size].reshape(a.shape))
            at += a.size
        return out


class _BatchInFlight:
    """A batch whose device work has been launched (batch_sides_start) but whose data-dependent sizes -- how many rows
    of the other side it touches, into how many segments they are cut -- have not been read yet."""
    pass


def batch_sides_start(ops, side, indptr_host, ids, n_other, seg_cap=layout.SEG_CAP, staging=None):
    """First half of `batch_sides`: everything that needs no data-dependent size on the host.  The batch's own side is
    complete after it (numpy on the host copy of the row pointers -- whose sizes the host therefore knows -- and ONE
    gather launch); the other side is sorted (int32 keys) and its three sizes (rows, segments, split rows) are on
    their way to pinned host memory.  No host synchronisation."""
    dev = ops.device
    b = _BatchInFlight()
    b.ops, b.cap = ops, int(seg_cap)
    rows_h = np.sort(np.ascontiguousarray(ids).astype(np.int64, copy=False))
    st = indptr_host[rows_h]
    deg = indptr_host[rows_h + 1] - st
    keep = deg > 0
    rk, stk, dk = (rows_h, st, deg) if keep.all() else (rows_h[keep], st[keep], deg[keep])
    nr = int(rk.shape[0])
    dst = np.zeros(nr + 1, dtype=np.int64)
    np.cumsum(dk, out=dst[1:])
    total = b.total = int(dst[-1])
    nsr = (dk + (seg_cap - 1)) // seg_cap
    rsp = np.zeros(nr + 1, dtype=np.int64)
    np.cumsum(nsr, out=rsp[1:])
    nseg = int(rsp[-1])
    if nseg == nr:                       # no row of the batch is longer than a segment
        begin, length = dst[:-1], dk | layout.SEG_WHOLE_ROW
        row_of = rk
    else:
        local = np.repeat(np.arange(nr, dtype=np.int64), nsr)
        within = np.arange(nseg, dtype=np.int64) - rsp[local]
        begin = dst[local] + within * seg_cap
        length = np.minimum(dk[local] - within * seg_cap, seg_cap) | np.where(nsr[local] == 1, layout.SEG_WHOLE_ROW, 0)
        row_of = rk[local]
    segs_h = np.stack([begin, length | (row_of << 32)], axis=1)
    multi_h = np.nonzero(nsr > 1)[0]

    host_arrays = [rows_h, rk, stk, dst, rsp, segs_h, multi_h]
    if staging is not None:
        d_rows, d_rk, d_stk, d_dst, d_rsp, d_segs, d_multi = staging.upload(host_arrays)
    else:
        d_rows, d_rk, d_stk, d_dst, d_rsp, d_segs, d_multi = (torch.from_numpy(np.ascontiguousarray(a)).to(dev)
                                                              for a in host_arrays)
    b.rows_all = d_rows
    rows_own = d_rows if nr == rows_h.shape[0] else d_rk
    out_idx = torch.empty(total, dtype=torch.int32, device=dev)
    out_y = torch.empty(total, dtype=torch.float32, device=dev)
    out_row = torch.empty(total, dtype=torch.int32, device=dev)
    if total:
        ops.gather_rows(d_stk, d_dst, rows_own, side.idx, side.y, out_idx, out_y, out_row)
    b.own = BatchSide.from_parts(rows_own, out_idx, out_y, d_rsp, d_segs.reshape(-1, 2), d_multi, nseg,
                                 multi_h.shape[0])
    # the other side: a stable sort of the gathered ids groups the batch by them (ties keep the order of the own side)
    order = torch.sort(out_idx, stable=True).indices
    b.o_idx, b.o_y = out_row[order], out_y[order]
    b.sizes_host = b.sizes_ready = None
    if total:
        # nonzeros per row of the other side (a histogram over ALL its rows: integer atomics, exact) and, from it, the
        # three sizes the second half needs: rows present, segments, rows cut into several segments
        b.per_row = torch.zeros(int(n_other), dtype=torch.int32, device=dev)
        b.per_row.index_add_(0, out_idx, torch.ones(total, dtype=torch.int32, device=dev))
        sizes = torch.stack([(b.per_row > 0).sum(), ((b.per_row + (seg_cap - 1)) // seg_cap).sum(),
                             (b.per_row > seg_cap).sum()])
        if dev.type == "cuda":
            b.sizes_host = torch.empty(3, dtype=torch.int64, pin_memory=True)
            b.sizes_host.copy_(sizes, non_blocking=True)
            b.sizes_ready = torch.cuda.Event()
            b.sizes_ready.record()
        else:
            b.sizes_host = sizes
    return b


def batch_sides_finish(b):
    """Second half: reads the three sizes (long since on the host when the first half ran a batch earlier) and lays out
    the other side's row list and segment descriptors.  -> (rows of the list, ascending, on the device; own BatchSide;
    other BatchSide)."""
    ops, cap, dev = b.ops, b.cap, b.ops.device
    if not b.total:
        e = torch.empty(0, dtype=torch.int64, device=dev)
        other = BatchSide.from_parts(e, b.o_idx, b.o_y, torch.zeros(1, dtype=torch.int64, device=dev),
                                     torch.empty((0, 2), dtype=torch.int64, device=dev), e, 0, 0)
        return b.rows_all, b.own, other
    if b.sizes_ready is not None:
        b.sizes_ready.synchronize()
    R, nseg, nmulti = (int(v) for v in b.sizes_host.tolist())
    o_rows = torch.nonzero_static(b.per_row > 0, size=R).reshape(-1)            # ascending = the sorted keys' runs
    counts = b.per_row[o_rows].to(torch.int64)
    starts = torch.cumsum(counts, 0) - counts
    o_nsr = (counts + (cap - 1)) // cap
    o_rsp = torch.zeros(R + 1, dtype=torch.int64, device=dev)
    torch.cumsum(o_nsr, 0, out=o_rsp[1:])
    o_segs = torch.empty((nseg, 2), dtype=torch.int64, device=dev)
    ops.fill_segments(starts, counts, o_rsp, o_rows, cap, o_segs)
    o_multi = torch.nonzero_static(o_nsr > 1, size=nmulti).reshape(-1)
    other = BatchSide.from_parts(o_rows, b.o_idx, b.o_y, o_rsp, o_segs, o_multi, nseg, nmulti)
    return b.rows_all, b.own, other


def batch_sides(ops, side, indptr_host, ids, n_other, seg_cap=layout.SEG_CAP):
    """The two BatchSides of the batch made of the listed rows of `side` (the users' CSR for a user batch, the items'
    CSC for an item batch) -> (rows of the list, ascending, on the device; the batch grouped by those rows; the batch
    grouped by the other side's rows).  Same structures as BatchSide(gather_rows(...)) builds with ~150 tensor-library
    launches and four host synchronisations per batch -- a batch was bound by the host, not by its kernels.  In two
    halves, so that a driver can run the first one a batch ahead and never waits for a size."""
    return batch_sides_finish(batch_sides_start(ops, side, indptr_host, ids, n_other, seg_cap))


class DeviceModel:
    """The variational state as padded device tables; `v(name)` is the [:, :k] view used by the dense algebra."""

    def __init__(self, ops, k, nU, nI, tables=None):
        """`tables`: padded device tensors ([n, ld] tables, [n] scalar rates; zero pad columns) to take over instead of
        allocating -- the state a full-batch fit leaves on the device (the arrays it names must all be given)."""
        self.ops, self.k, self.ld = ops, int(k), _lib.ld_for_k(int(k))
        self.nU, self.nI = int(nU), int(nI)
        dev = ops.device
        f32 = dict(dtype=torch.float32, device=dev)
        tables = tables or {}
        for names, rows in ((("Gamma_shp", "Gamma_rte", "Theta", "eT"), self.nU),
                            (("Lambda_shp", "Lambda_rte", "Beta", "eB"), self.nI)):
            for n in names:
                t = tables.get(n)
                if t is None:
                    t = torch.zeros((rows, self.ld), **f32)
                assert tuple(t.shape) == (rows, self.ld) and t.dtype == torch.float32 and t.is_contiguous(), n
                setattr(self, n, t)
        self.k_rte = tables["k_rte"] if "k_rte" in tables else torch.zeros(self.nU, **f32)
        self.t_rte = tables["t_rte"] if "t_rte" in tables else torch.zeros(self.nI, **f32)
        assert tuple(self.k_rte.shape) == (self.nU,) and tuple(self.t_rte.shape) == (self.nI,)
        self._cs_part = torch.zeros((ops.refresh_grid(max(self.nU, self.nI)), self.ld), **f32)
        # full-height accumulator tables: the batch sweeps write a row's phi-sums straight to acc[row]
        self.acc_u = torch.zeros((self.nU, self.ld), **f32)
        self.acc_i = torch.zeros((self.nI, self.ld), **f32)
        self.flag_u = torch.zeros(self.nU, dtype=torch.uint8, device=dev)   # rows of the current step, per side
        self.flag_i = torch.zeros(self.nI, dtype=torch.uint8, device=dev)
        self.csT = torch.zeros(self.ld, **f32)      # Theta.sum(axis=0) / Beta.sum(axis=0): set by put(), kept
        self.csB = torch.zeros(self.ld, **f32)      # current by every step
        if "Theta" in tables:
            self.csT = self.colsum("Theta")
        if "Beta" in tables:
            self.csB = self.colsum("Beta")

    def v(self, name):
        return getattr(self, name)[:, : self.k]

    def put(self, name, host):
        """Upload one state array ([n,k] table or [n,1] scalar-rate vector) from the host."""
        dev = self.ops.device
        a = torch.from_numpy(np.ascontiguousarray(host, dtype=np.float32))
        if name in ("k_rte", "t_rte"):
            getattr(self, name).copy_(a.reshape(-1))
        else:
            self.v(name).copy_(a.to(dev))
            if name == "Theta":           # Theta.sum(axis=0) / Beta.sum(axis=0) are kept current by every step
                self.csT = self.colsum("Theta")
            elif name == "Beta":
                self.csB = self.colsum("Beta")

    def get(self, name, out=None):
        """Download one state array; into `out` (in place) when given."""
        t = getattr(self, name).reshape(-1, 1) if name in ("k_rte", "t_rte") else self.v(name)
        if out is not None and isinstance(out, np.ndarray) and out.flags.c_contiguous and out.flags.writeable \
                and out.dtype == np.float32 and tuple(out.shape) == tuple(t.shape):
            torch.from_numpy(out).copy_(t)       # one device-to-host copy, no second pass on the host
            return out
        a = t.contiguous().cpu().numpy()
        if out is None:
            return a
        out[...] = a
        return out

    def load(self, Gamma_shp, Gamma_rte, Lambda_shp, Lambda_rte, k_rte, t_rte, Theta, Beta):
        for n, a in zip(_NAMES + ("k_rte", "t_rte"),
                        (Gamma_shp, Gamma_rte, Lambda_shp, Lambda_rte, Theta, Beta, k_rte, t_rte)):
            self.put(n, a)

    def init_state(self, raw, hy):
        """initialize_parameters (PXI:127-141) from the MT19937 stream's first 2*(nU + nI)*k words (cavi.draw_init_words):
        the four tables in the reference's draw order, bit-identical to its numpy draws; see cavi.init_state."""
        ops, k, ld, nU, nI = self.ops, self.k, self.ld, self.nU, self.nI
        assert raw.numel() == 2 * (nU + nI) * k
        for n in _NAMES:
            getattr(self, n).zero_()
        ops.uniform_rows(raw[: nU * k], self.Gamma_rte, nU, k, ld, hy.a_prime, 0.01)
        ops.uniform_rows(raw[nU * k: (nU + nI) * k], self.Lambda_rte, nI, k, ld, hy.c_prime, 0.01)
        ops.uniform_rows(raw[(nU + nI) * k: (2 * nU + nI) * k], self.Gamma_shp, nU, k, ld, hy.a_prime, 0.01,
                         den=self.Gamma_rte, ratio=self.Theta)
        ops.uniform_rows(raw[(2 * nU + nI) * k:], self.Lambda_shp, nI, k, ld, hy.c_prime, 0.01, den=self.Lambda_rte,
                         ratio=self.Beta)
        self.k_rte.fill_(float(hy.b_prime))
        self.t_rte.fill_(float(hy.d_prime))
        self.csT = self.colsum("Theta")
        self.csB = self.colsum("Beta")

    def store(self, Gamma_shp, Gamma_rte, Lambda_shp, Lambda_rte, k_rte, t_rte, Theta, Beta):
        for n, a in zip(_NAMES + ("k_rte", "t_rte"),
                        (Gamma_shp, Gamma_rte, Lambda_shp, Lambda_rte, Theta, Beta, k_rte, t_rte)):
            self.get(n, out=a)

    def colsum(self, name):
        """tab.sum(axis=0) -> [ld] (HIP colsum kernels; PXI:300,320,352,372)."""
        tab = getattr(self, name)
        part = torch.zeros((self.ops.finalize_grid(tab.shape[0]), self.ld), dtype=torch.float32, device=self.ops.device)
        self.ops.colsum(tab, tab.shape[0], self.ld, part)
        out = torch.zeros(self.ld, dtype=torch.float32, device=self.ops.device)
        self.ops.colsum_reduce(part, out, self.ld)
        return out

    # ------------------------------------------------------------------------------------------
    def batch_phi_sums(self, su, si):
        """Per touched row, sum_n w_n * (other side's E row) over the batch's nonzeros (update_phi[_csr] +
        update_G_n_L_sh[_csr] restricted to the batch; sum phi = E_row (*) this), from the CURRENT shapes/rates,
        left in acc_u[user] / acc_i[item] for the rows present in the batch (su / si: its two BatchSides)."""
        ops, k, ld = self.ops, self.k, self.ld
        ops.expect(self.Gamma_shp, self.Gamma_rte, self.eT, su.nrows, k, ld, row_list=su.rows)
        ops.expect(self.Lambda_shp, self.Lambda_rte, self.eB, si.nrows, k, ld, row_list=si.rows)
        for side, e_self, e_other, acc in ((su, self.eT, self.eB, self.acc_u), (si, self.eB, self.eT, self.acc_i)):
            if side.nseg == 0:
                continue
            part = torch.empty((side.nseg, ld), dtype=torch.float32, device=ops.device)
            # a row that is one segment long writes its sums straight to acc[row]; split rows go through part[]
            ops.sweep(side, e_self, e_other, part, k, ld, acc_rows=acc, acc_ld=ld)
            if side.nmulti > 0:
                tmp = torch.zeros((side.nmulti, ld), dtype=torch.float32, device=ops.device)
                ops.segsum(part, side.row_seg_ptr, side.nmulti, tmp, ld, row_list=side.multi_local)
                acc.index_copy_(0, side.rows[side.multi_local], tmp)


def _svi_step(m, hy, su, si, users_tb, items_tb, step, mult, user_batch, all_scalar_rows):
    """One stochastic update in the reference's statement order (user batch: PXI:292-325 / 438-473;
    item batch: PXI:344-377).  su / si: the batch grouped by user / by item; users_tb / items_tb: the row lists
    the updates run over (supersets of the rows present in the batch: listed rows without a nonzero get a zero
    phi-sum).  `hy` carries a, c, k_shp, t_shp, add_k_rte, add_t_rte as python floats."""
    ops, k, ld = m.ops, m.k, m.ld
    step_prev = float(np.float32(1) - np.float32(step))
    step = float(np.float32(step))
    w_other = float(np.float32(step * float(np.float32(mult))))   # step*multiplier as one float32 scalar (PXI:316)
    for tb, side, acc in ((users_tb, su, m.acc_u), (items_tb, si, m.acc_i)):
        if tb.shape[0] != side.nrows:        # listed rows without any nonzero in the batch
            acc.index_fill_(0, tb, 0.0)
    m.batch_phi_sums(su, si)                                      # phi from the OLD parameters
    for flag, tb in ((m.flag_u, users_tb), (m.flag_i, items_tb)):
        flag.zero_()
        flag.index_fill_(0, tb, 1)
    U = dict(n=m.nU, flag=m.flag_u, shp=m.Gamma_shp, rte=m.Gamma_rte, fac=m.Theta, rs=m.k_rte, e=m.eT, acc=m.acc_u,
             prior=hy["a"], top=hy["k_shp"], add=hy["add_k_rte"], cs="csT")
    I = dict(n=m.nI, flag=m.flag_i, shp=m.Lambda_shp, rte=m.Lambda_rte, fac=m.Beta, rs=m.t_rte, e=m.eB, acc=m.acc_i,
             prior=hy["c"], top=hy["t_shp"], add=hy["add_t_rte"], cs="csB")
    B, O = (U, I) if user_batch else (I, U)     # batch side, other side
    # SVI epochs blend the scalar rates of the step's rows only (PXI:324-325, 376-377), partial_fit of all (PXI:472-473)
    rs_mode = 2 if all_scalar_rows else 1
    # One pass per side (hpf_hip_svi_side_f32).  Batch side: shapes of its rows reset to prior + phi, the rate of
    # EVERY row from the other side's current column sums, means, scalar rates, column sums ...
    ops.svi_side(B["n"], B["flag"], B["acc"], B["e"], B["shp"], B["rte"], B["fac"], B["rs"], getattr(m, O["cs"]),
                 m._cs_part, B["prior"], 1.0, 0.0, B["top"], B["add"], step, step_prev, 0, rs_mode, k, ld)
    cs_batch = torch.zeros(ld, dtype=torch.float32, device=ops.device)
    ops.colsum_reduce(m._cs_part, cs_batch, ld)
    setattr(m, B["cs"], cs_batch)
    # ... other side: shapes and rates of the touched rows blended towards the step's estimate (the rates with the
    # batch side's NEW column sums), means of every row, scalar rates, column sums
    ops.svi_side(O["n"], O["flag"], O["acc"], O["e"], O["shp"], O["rte"], O["fac"], O["rs"], cs_batch, m._cs_part,
                 O["prior"], w_other, step_prev, O["top"], O["add"], step, step_prev, 1, rs_mode, k, ld)
    cs_o = torch.zeros(ld, dtype=torch.float32, device=ops.device)
    ops.colsum_reduce(m._cs_part, cs_o, ld)
    setattr(m, O["cs"], cs_o)


def gather_rows(side, rows):
    """COO triplets (row, col, y) of the listed rows of a SparseSide (ascending `rows`: the triplets come grouped)."""
    st = side.indptr[rows]
    deg = side.indptr[rows + 1] - st
    total = int(deg.sum().item())
    offs = torch.cumsum(deg, 0) - deg
    pos = torch.repeat_interleave(st - offs, deg, output_size=total) + torch.arange(total, device=rows.device)
    return (torch.repeat_interleave(rows, deg, output_size=total), side.idx[pos].to(torch.int64), side.y[pos])


def _dev_ids(a, dev):
    return layout.ids_to_device(a, dev)


# -- PXI:423-473 ------------------------------------------------------------------------------------
def partial_fit_device(m, Y_batch, ix_u_batch, ix_i_batch, add_k_rte, add_t_rte, a, c, k_shp, t_shp,
                       users_this_batch, items_this_batch, step_size_batch, multiplier_batch, user_batch):
    """One partial_fit step on a DeviceModel that already holds the current state: only the batch's triplets and
    row lists cross PCIe.  Mutates all eight state tables of `m` (as the reference mutates its arrays, PXI:443-473)."""
    ops = m.ops
    if ix_u_batch.size and (int(ix_u_batch.max()) >= m.nU or int(ix_i_batch.max()) >= m.nI):
        raise ValueError("partial_fit: user/item id out of range")
    dev = ops.device
    hy = {"a": float(np.float32(a)), "c": float(np.float32(c)), "k_shp": float(np.float32(k_shp)),
          "t_shp": float(np.float32(t_shp)), "add_k_rte": float(np.float32(add_k_rte)),
          "add_t_rte": float(np.float32(add_t_rte))}
    bu, bi = _dev_ids(ix_u_batch, dev), _dev_ids(ix_i_batch, dev)
    by = torch.from_numpy(np.ascontiguousarray(Y_batch, dtype=np.float32)).to(dev)
    su, si = BatchSide(bu, bi, by), BatchSide(bi, bu, by)
    users_tb, items_tb = _dev_ids(users_this_batch, dev), _dev_ids(items_this_batch, dev)
    if not (bool(torch.isin(su.rows, users_tb).all()) and bool(torch.isin(si.rows, items_tb).all())):
        raise ValueError("the batch contains users/items that are not in users_in_batch/items_in_batch")
    _svi_step(m, hy, su, si, users_tb, items_tb, step_size_batch, multiplier_batch, user_batch, all_scalar_rows=True)


def partial_fit_step(ops, Y_batch, ix_u_batch, ix_i_batch, Theta, Beta, Gamma_shp, Gamma_rte, Lambda_shp,
                     Lambda_rte, k_rte, t_rte, add_k_rte, add_t_rte, a, c, k_shp, t_shp, k, users_this_batch,
                     items_this_batch, step_size_batch, multiplier_batch, user_batch):
    """The module-level form (drop-in for the extension function): the caller's host arrays are the state, so all
    eight go up and come back, in place.  hpfrec_amd.HPF keeps the state on the device between calls instead
    (hpfrec_amd.resident)."""
    m = DeviceModel(ops, k, Theta.shape[0], Beta.shape[0])
    m.load(Gamma_shp, Gamma_rte, Lambda_shp, Lambda_rte, k_rte, t_rte, Theta, Beta)
    partial_fit_device(m, Y_batch, ix_u_batch, ix_i_batch, add_k_rte, add_t_rte, a, c, k_shp, t_shp,
                       users_this_batch, items_this_batch, step_size_batch, multiplier_batch, user_batch)
    m.store(Gamma_shp, Gamma_rte, Lambda_shp, Lambda_rte, k_rte, t_rte, Theta, Beta)


# -- PXI:262-377 (+ the shared convergence tail PXI:380-418) ---------------------------------------------
def fit_hpf_svi(hy, Y, ix_u, ix_i, Theta, Beta, Gamma_shp, Gamma_rte, Lambda_shp, Lambda_rte, k_rte, t_rte, maxiter,
                stop_crit, check_every, stop_thr, users_per_batch, items_per_batch, step_size, save_folder,
                random_seed, verbose, has_valset, Yval, ix_u_val, ix_i_val, full_llk, keep_all_objs, make_ops,
                device_triplets=None, init_draw=None, resident=None, tick=None):
    """`init_draw`: the running device draw of the initial state (cython_loops_float.start_init_draw); the eight host
    arrays are then outputs only.  Without it they hold the initial state (initialize_parameters).
    `resident`: see cython_loops_float.fit_hpf."""
    from . import cython_loops_float as be   # printing helpers and save_parameters
    import time
    tick = tick or (lambda phase: None)
    ops = make_ops()
    dev = ops.device
    nU, k = Theta.shape
    nI = Beta.shape[0]
    m = DeviceModel(ops, k, nU, nI)
    tick("state tables allocated (the phase clock also waits for the MT19937 recurrence on its side stream here)")
    if init_draw is None:
        m.load(Gamma_shp, Gamma_rte, Lambda_shp, Lambda_rte, k_rte, t_rte, Theta, Beta)
    hyd = {"a": float(hy.a), "c": float(hy.c), "k_shp": float(hy.k_shp), "t_shp": float(hy.t_shp),
           "add_k_rte": float(hy.add_k_rte), "add_t_rte": float(hy.add_t_rte)}

    if device_triplets is not None:
        tu, ti, ty = (t.to(dev) for t in device_triplets)
    else:
        tu = _dev_ids(ix_u, dev)
        ti = _dev_ids(ix_i, dev)
        ty = torch.from_numpy(np.ascontiguousarray(Y, dtype=np.float32)).to(dev)
    users, items, u_sorted = layout.build_sides(tu, ti, ty, nU, nI)
    tick("triplets to the device, CSR/CSC layout")
    if init_draw is not None:        # (the MT19937 recurrence ran on its own stream under the uploads and sorts above)
        m.init_state(be.finish_init_draw(dev, init_draw), hy)
        tick("initial tables (incl. waiting for the MT19937 recurrence)")

    val = None
    if has_valset and Yval is not None and Yval.shape[0] > 0:
        val = (_dev_ids(ix_u_val, dev).to(torch.int32), _dev_ids(ix_i_val, dev).to(torch.int32),
               torch.from_numpy(np.ascontiguousarray(Yval, dtype=np.float32)).to(dev))

    users_numeration = np.arange(nU, dtype=np.uint64) if users_per_batch != 0 else None
    items_numeration = np.arange(nI, dtype=np.uint64) if items_per_batch > 0 else None
    nbatches_u = int(np.ceil(float(nU) / float(users_per_batch))) if users_per_batch != 0 else 0
    nbatches_i = int(np.ceil(float(nI) / float(items_per_batch))) if items_per_batch > 0 else 0
    rng = np.random.default_rng(seed=random_seed if random_seed > 0 else None)   # PXI:207

    # HPF_SVI_PREP=torch: the batch structures built with tensor-library calls only (the round-1 path; A/B and tests)
    fast_prep = os.environ.get("HPF_SVI_PREP", "fast") != "torch"
    indptr_host = (users.indptr.cpu().numpy(), items.indptr.cpu().numpy()) if fast_prep else None
    # (HPF_SVI_PREP_STREAM=0: the preparation on the compute stream, between the batches -- measured slower, 4.7 vs 4.2 ms
    # per C5 batch)
    own_stream = os.environ.get("HPF_SVI_PREP_STREAM", "1") == "1"
    prep_stream = torch.cuda.Stream(device=dev) if (dev.type == "cuda" and own_stream) else None
    # (a batch sends ~7 words per listed row; a batch that needs more falls back to plain copies)
    staging = PinnedStaging(dev, words=8 * max(int(users_per_batch), int(items_per_batch), 1) + 4096, slots=3) \
        if (fast_prep and dev.type == "cuda") else None
    if prep_stream is not None:
        prep_stream.wait_stream(torch.cuda.current_stream(dev))      # the CSR / CSC built above

    import contextlib

    def on_prep_stream():
        return torch.cuda.stream(prep_stream) if prep_stream is not None else contextlib.nullcontext()

    def prepare_start(ids, user_epoch):
        """Launches the index work of a batch (the listed rows gathered from the CSR / CSC, the other side sorted) on the
        preparation stream; nothing here waits for the device (fast path)."""
        with on_prep_stream():
            if fast_prep:
                return batch_sides_start(ops, users if user_epoch else items, indptr_host[0 if user_epoch else 1], ids,
                                         nI if user_epoch else nU, staging=staging)
            rows = torch.sort(_dev_ids(ids, dev)).values             # ascending: the gathered triplets come grouped
            if user_epoch:
                bu, bi, by = gather_rows(users, rows)
                return BatchSide(bu, bi, by, grouped=True), BatchSide(bi, bu, by), rows
            bi, bu, by = gather_rows(items, rows)
            return BatchSide(bu, bi, by), BatchSide(bi, bu, by, grouped=True), rows

    def prepare_finish(started, user_epoch):
        """-> ((su, si, users_tb, items_tb), ready event): the batch grouped by user and by item."""
        with on_prep_stream():
            if fast_prep:
                rows, own, other = batch_sides_finish(started)
                su, si = (own, other) if user_epoch else (other, own)
            else:
                su, si, rows = started
            out = (su, si, rows, si.rows) if user_epoch else (su, si, su.rows, rows)
            ready = None
            if prep_stream is not None:
                ready = torch.cuda.Event()
                ready.record(prep_stream)
        return out, ready

    errs = np.zeros(2, dtype=np.longdouble)
    last_crit = -np.inf
    Theta_prev = m.Theta.clone() if stop_crit == "diff-norm" else None

    def evaluate(final=False):
        if val is not None:
            t = ops.pair_llk(m.Theta, m.Beta, val[0], val[1], val[2], k, m.ld, full_llk).cpu().numpy()
            if final:
                sub = float(np.dot(m.v("Theta")[val[0].long()].sum(dim=0).cpu().numpy(),
                                   m.v("Beta")[val[1].long()].sum(dim=0).cpu().numpy()))
            else:
                sub = t[2]
            errs[0] = np.longdouble(t[0]) - np.longdouble(sub)
            errs[1] = np.sqrt(np.longdouble(t[1]) / val[0].shape[0])
        else:
            t = ops.pair_llk(m.Theta, m.Beta, u_sorted, users.idx, users.y, k, m.ld, full_llk).cpu().numpy()
            sub = np.dot(m.csT[:k].cpu().numpy(), m.csB[:k].cpu().numpy())
            errs[0] = np.longdouble(t[0]) - np.longdouble(sub)
            errs[1] = np.sqrt(np.longdouble(t[1]) / users.nnz)

    if verbose > 0:
        print("Initializing optimization procedure...")
    st_time = time.time()
    timing = os.environ.get("HPF_TIMING") == "1"      # SVI_TIMINGS: device-synchronised wall time of the epoch loop
    if timing and dev.type == "cuda":
        torch.cuda.synchronize(dev)
    t_loop = time.perf_counter()
    host_s = [0.0, 0.0]       # host time spent preparing batches / issuing their kernels
    i = -1
    for i in range(maxiter):
        step = float(np.float32(step_size(i)))
        if users_per_batch > 0 and items_per_batch > 0:
            user_epoch = ((i + 1) % 2) == 0        # PXI:265-269: epoch 0 is an item epoch
        else:
            user_epoch = users_per_batch > 0
        if user_epoch:
            rng.shuffle(users_numeration)
            chunks = [users_numeration[bt * users_per_batch: min(nU, (bt + 1) * users_per_batch)].copy()
                      for bt in range(nbatches_u)]
        else:
            rng.shuffle(items_numeration)
            chunks = [items_numeration[bt * items_per_batch: min(nI, (bt + 1) * items_per_batch)].copy()
                      for bt in range(nbatches_i)]
        n_side = nU if user_epoch else nI
        # a batch's index structures (its rows gathered from the CSR / CSC, grouped by both sides, cut into segments)
        # are data-only and are prepared on a second stream in two halves (batch_sides_start / _finish): before batch j
        # is issued, batch j+1 -- started one batch ago, so the sizes it needs from the device are on the host by now
        # -- is finished and batch j+2 is started, so a batch's structures are ready long before its turn.  What is
        # left of the preparation in a C5 batch is ~0.7 ms of 4.0 (3.3 ms with every batch re-using the first one's
        # structures): its ~50 small launches run between / beside the batch's long whole-GPU kernels
        # (profiles/r02_svi_c5_timeline.txt)
        pending = prepare_finish(prepare_start(chunks[0], user_epoch), user_epoch)
        started = prepare_start(chunks[1], user_epoch) if len(chunks) > 1 else None
        for j in range(len(chunks)):
            t_h = time.perf_counter()
            # batch j+1 was started one batch ago: its sizes are on the host by now; batch j+2 is started
            nxt = prepare_finish(started, user_epoch) if started is not None else None
            started = prepare_start(chunks[j + 2], user_epoch) if j + 2 < len(chunks) else None
            host_s[0] += time.perf_counter() - t_h
            t_h = time.perf_counter()
            (su, si, utb, itb), ready = pending
            if ready is not None:
                torch.cuda.current_stream(dev).wait_event(ready)
                for t in su.tensors() + si.tensors() + [utb, itb]:
                    t.record_stream(torch.cuda.current_stream(dev))
            _svi_step(m, hyd, su, si, utb, itb, step, float(n_side) / float(chunks[j].shape[0]), user_epoch,
                      all_scalar_rows=False)
            pending = nxt
            host_s[1] += time.perf_counter() - t_h

        if check_every > 0 and ((i + 1) % check_every) == 0:
            if stop_crit == "diff-norm":
                d = (m.Theta - Theta_prev).double()
                last_crit = float(torch.sqrt((d * d).sum()).item())
                if verbose:
                    be._print_norm_diff(i + 1, check_every, last_crit)
                if last_crit < stop_thr:
                    break
                Theta_prev.copy_(m.Theta)
            else:
                evaluate()
                if verbose:
                    be._print_llk_iter(i + 1, errs[0], float(errs[1]), has_valset)
                if stop_crit != "maxiter":
                    if (i + 1) == check_every:
                        last_crit = errs[0]
                    else:
                        if (1.0 - errs[0] / last_crit) <= stop_thr:
                            break
                        last_crit = errs[0]

    tick("epochs and checks")
    if timing:
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        SVI_TIMINGS.clear()
        SVI_TIMINGS.update(epochs=i + 1, seconds=time.perf_counter() - t_loop, host_prepare_s=host_s[0],
                           host_issue_s=host_s[1])
    last_llk = None
    if stop_crit in ("diff-norm", "maxiter") and verbose > 0:
        evaluate(final=True)
        last_llk = errs[0]
    if verbose:
        be._print_final_msg(i + 1, errs[0], float(errs[1]), (time.time() - st_time) / 60.0)

    if resident is not None and keep_all_objs and save_folder == "":
        resident.adopt(m)              # the state stays on the device; host copies are made when somebody reads them
        return i, None, last_llk
    m.store(Gamma_shp, Gamma_rte, Lambda_shp, Lambda_rte, k_rte, t_rte, Theta, Beta)
    tick("outputs to the host")
    temp = (Gamma_shp, Gamma_rte, Lambda_shp, Lambda_rte, k_rte, t_rte)
    if save_folder != "":
        be.save_parameters(verbose, save_folder,
                           ["Theta", "Beta", "Gamma_shp", "Gamma_rte", "Lambda_shp", "Lambda_rte", "kappa_rte", "tau_rte"],
                           [Theta, Beta] + list(temp))
    return i, (temp if keep_all_objs else None), last_llk


# -- PXI:476-520 ------------------------------------------------------------------------------------
def calc_user_factors(ops, a, a_prime, b_prime, c, c_prime, d_prime, Y, ix_i, Theta, Beta, Lambda_shp, Lambda_rte,
                      nY, k, maxiter, random_seed, stop_thr, return_all, resident=None):
    """Local CAVI for ONE user with the item parameters fixed.  Fills `Theta` (k,) in place; returns
    (Gamma_shp, Gamma_rte, phi/Y) when return_all else None.

    resident: a DeviceModel holding the current Beta / Lambda_shp / Lambda_rte (then the host arguments of those
    names are not read at all: the user's item rows are gathered on the device and Beta.sum(axis=0) is the
    model's resident column sum)."""
    f = np.float32
    dev = ops.device
    ld = _lib.ld_for_k(k)
    a, a_prime, b_prime = f(a), f(a_prime), f(b_prime)
    k_shp = f(a_prime + f(k) * a)
    add_k_rte = f(a_prime / b_prime)
    # initialisation: numpy default_rng stream, in the reference's draw order (PXI:490-497)
    rng = np.random.default_rng(seed=random_seed if random_seed > 0 else None)
    Theta[:] = rng.gamma(a, 1 / b_prime, size=k).astype(np.float32)
    k_rte = f(b_prime + Theta.sum())
    if resident is not None:
        cs = resident.csB
    else:
        Beta_dev = torch.zeros((Beta.shape[0], ld), dtype=torch.float32, device=dev)
        Beta_dev[:, :k] = torch.from_numpy(np.ascontiguousarray(Beta, dtype=np.float32)).to(dev)
        csp = torch.zeros((ops.finalize_grid(Beta.shape[0]), ld), dtype=torch.float32, device=dev)
        cs = torch.zeros(ld, dtype=torch.float32, device=dev)
        ops.colsum(Beta_dev, Beta.shape[0], ld, csp)
        ops.colsum_reduce(csp, cs, ld)
    g1 = rng.gamma(a_prime, b_prime / a_prime, size=1).astype(np.float32)
    unif = rng.uniform(low=.85, high=1.15, size=k).astype(np.float32)
    ix = np.ascontiguousarray(ix_i).astype(np.int64)
    n = int(nY)
    # Gamma_rte = g + Beta.sum(axis=0); Gamma_shp = Gamma_rte * Theta * U(0.85, 1.15) (PXI:493-497), on the device
    init = np.zeros((3, ld), dtype=np.float32)
    init[0, :k], init[1, :k], init[2, 0] = Theta, unif, g1[0]
    init_d = torch.from_numpy(init).to(dev)
    Gr = torch.zeros(ld, dtype=torch.float32, device=dev)
    Gs = torch.zeros(ld, dtype=torch.float32, device=dev)
    Gr[:k] = init_d[2, 0] + cs[:k]
    Gs[:k] = Gr[:k] * init_d[0, :k] * init_d[1, :k]
    torch.nan_to_num_(Gs)
    torch.nan_to_num_(Gr)
    th = init_d[0].clone()
    # E rows of the user's items: in place in the resident model's scratch table, or of the uploaded rows
    y_d = torch.from_numpy(np.ascontiguousarray(Y, dtype=np.float32)).to(dev)
    if resident is not None:
        ixd = torch.from_numpy(ix).to(dev)
        ops.expect(resident.Lambda_shp, resident.Lambda_rte, resident.eB, n, k, ld, row_list=ixd)
        e_items, idx32 = resident.eB, ixd.to(torch.int32)
    else:
        Ls = torch.zeros((n, ld), dtype=torch.float32, device=dev)
        Lr = torch.zeros((n, ld), dtype=torch.float32, device=dev)
        Ls[:, :k] = torch.from_numpy(np.ascontiguousarray(Lambda_shp[ix], dtype=np.float32)).to(dev)
        Lr[:, :k] = torch.from_numpy(np.ascontiguousarray(Lambda_rte[ix], dtype=np.float32)).to(dev)
        e_items = torch.zeros((n, ld), dtype=torch.float32, device=dev)
        ops.expect(Ls, Lr, e_items, n, k, ld)
        idx32 = torch.arange(n, dtype=torch.int32, device=dev)
    e_last = torch.zeros(ld, dtype=torch.float32, device=dev)
    rounds = torch.zeros(1, dtype=torch.int32, device=dev)
    # the whole local coordinate ascent (PXI:505-517) is one launch
    ops.fold_in(idx32, y_d, e_items, cs, Gs, Gr, th, e_last, rounds, a, k_shp, add_k_rte, k_rte, stop_thr, maxiter, k, ld)
    Theta[:] = th[:k].cpu().numpy()
    if not return_all:
        return None
    # phi / Y: the multinomial probabilities of the LAST phi (computed from the Gamma before its final update)
    prob = e_last[None, :] * e_items[idx32.long()]
    prob = (prob / prob.sum(dim=1, keepdim=True))[:, :k]
    return Gs[:k].cpu().numpy(), Gr[:k].cpu().numpy(), prob.contiguous().cpu().numpy()
