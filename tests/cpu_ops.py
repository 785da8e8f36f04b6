"""CpuOps: numpy stand-in for hpfrec_amd.ops_hip.HipOps -- TEST INFRASTRUCTURE ONLY.

Same method names and argument meaning as HipOps, float64 arithmetic on CPU torch tensors.
Two uses:
  * `-m "not gpu"` tests exercise the host logic (layout, driver loop, sharding over gloo,
    the HPF class) on machines without a GPU by monkeypatching the name `HipOps` in
    hpfrec_amd.cython_loops_float (tests/conftest.py fixtures);
  * `-m gpu` tests use it as the op-by-op reference of each HIP kernel (alongside the
    end-to-end comparison with the oracle / golden vectors).
It lives under tests/ so that nothing in the package can route through it.
"""
import numpy as np
import scipy.special as sp
import torch


_LN2 = float(np.log(2.0))


def _np(t):
    return t.numpy()


def _live_nseg(side):
    """Segments of a side that count: all of them, or -- a batch built on the device -- the count kept there."""
    nd = getattr(side, "nseg_dev", None)
    return int(side.nseg) if nd is None else min(int(side.nseg), int(_np(nd)[0]))


def _decode_segs(side):
    s = _np(side.segs)[: _live_nseg(side)]
    begin = s[:, 0].copy()
    meta = s[:, 1]
    length = (meta & 0x00FFFFFF).astype(np.int64)
    row = (meta >> 32).astype(np.int64)
    return begin, length, row


class CpuOps:
    name = "cpu-standin"

    def __init__(self):
        self.device = torch.device("cpu")
        self.sweep_blocks = 8
        self.finalize_blocks = 4

    def finalize_grid(self, nrows):
        return int(max(1, min(self.finalize_blocks, (nrows + 3) // 4)))

    def sweep_grid(self, nseg, blocks=None):
        return int(max(1, min(blocks or self.sweep_blocks, (nseg + 3) // 4)))

    def sweep_finalize(self, side, tab_self, tab_other, part, e_new, shp, rte, fac, rs, cs_other, cs_partial,
                       prior_shp, top_shp, add_rte, k, ld, rs_prev=None):
        """sweep, then finish exactly the rows that consist of one segment (HPF_SEG_WHOLE_ROW)."""
        e_old = tab_self.clone()
        self.sweep(side, tab_self, tab_other, part, k, ld)
        rsp = _np(side.row_seg_ptr)
        single = torch.from_numpy(np.nonzero((rsp[1:] - rsp[:-1]) == 1)[0].astype(np.int64))
        self.row_finalize(part, side.row_seg_ptr, int(single.shape[0]), e_old, e_new, shp, rte, fac, rs, cs_other,
                          cs_partial, prior_shp, top_shp, add_rte, k, ld, row_list=single, rs_prev=rs_prev)

    def sweep_prefinalize(self, side, tab_self, tab_other, part, acc_rows, acc_ld, shp, rte, fac, rs, cs_other,
                          cs_partial, prior_shp, top_shp, add_rte, k, ld, rs_prev=None):
        begin, length, row = _decode_segs(side)
        whole = (_np(side.segs)[:, 1] & 0x40000000) != 0
        rows = torch.from_numpy(row[whole].astype(np.int64))
        self.row_finalize(acc_rows, None, int(rows.shape[0]), tab_self, tab_self, shp, rte, fac, rs, cs_other,
                          cs_partial, prior_shp, top_shp, add_rte, k, ld, row_list=rows, part_ld=acc_ld,
                          rs_prev=rs_prev)
        self.sweep(side, tab_self, tab_other, part, k, ld, acc_rows=acc_rows, acc_ld=acc_ld)

    def sweep(self, side, tab_self, tab_other, part, k, ld, acc_rows=None, acc_ld=0, grid_blocks=None):
        nseg = _live_nseg(side)
        if nseg == 0:
            return
        begin, length, row = _decode_segs(side)
        seg_of_nnz = np.repeat(np.arange(nseg), length)
        offs = np.concatenate([[0], np.cumsum(length)[:-1]])
        # positions of every segment's nonzeros (segments may be any subset of the rows)
        pos = np.repeat(begin, length) + (np.arange(length.sum()) - np.repeat(offs, length))
        idx = _np(side.idx).astype(np.int64)[pos]
        y = _np(side.y).astype(np.float64)[pos]
        # (columns < k only: the pad columns are zero in both tables; float64 products summed row by row without the
        #  nnz x ld temporary)
        S = _np(tab_self)[:, :k].astype(np.float64)[row[seg_of_nnz]]
        O = _np(tab_other)[:, :k].astype(np.float64)[idx]
        s = np.einsum("ij,ij->i", S, O)
        w = np.where(y > 0, y / s, 0.0)
        O *= w[:, None]
        out = np.zeros((nseg, ld))
        out[:, :k] = np.add.reduceat(O, offs, axis=0)
        if acc_rows is not None:
            whole = (_np(side.segs)[:nseg, 1] & 0x40000000) != 0
            _np(acc_rows)[row[whole], :acc_ld] = out[whole][:, :acc_ld].astype(np.float32)
            _np(part)[:nseg][~whole] = out[~whole].astype(np.float32)
        else:
            _np(part)[:nseg] = out.astype(np.float32)

    def row_finalize(self, part, row_seg_ptr, nrows, e_old, e_new, shp, rte, fac, rs, cs_other, cs_partial,
                     prior_shp, top_shp, add_rte, k, ld, row_list=None, part_ld=None, rs_prev=None):
        P = _np(part).astype(np.float64)
        if part_ld is not None and part_ld != ld:   # packed accumulator rows (stride part_ld <= ld)
            Pp = np.zeros((P.shape[0], ld))
            Pp[:, :part_ld] = P
            P = Pp
        rows = np.arange(nrows) if row_list is None else _np(row_list)[:nrows].astype(np.int64)
        cp = _np(cs_partial)
        cp[:] = 0
        if rows.shape[0] == 0:
            return
        if row_seg_ptr is None:
            acc = P[rows]
        else:
            rsp = _np(row_seg_ptr)
            acc = np.zeros((rows.shape[0], ld))
            for t, r in enumerate(rows):  # rows may be any subset; plain loop keeps this obviously right
                if rsp[r + 1] > rsp[r]:
                    acc[t] = P[rsp[r]: rsp[r + 1]].sum(axis=0)
        f = np.float32
        sh = (f(prior_shp) + (_np(e_old).astype(np.float64)[rows] * acc)).astype(np.float32)
        rt = (f(top_shp) / _np(rs)[rows][:, None] + _np(cs_other)[None, :]).astype(np.float32)
        valid = np.arange(ld) < k
        fc = np.where(valid[None, :], sh / rt, 0).astype(np.float32)
        with np.errstate(divide="ignore", invalid="ignore"):
            E = sp.psi(sh.astype(np.float64)) - np.log(rt.astype(np.float64))
        E = np.where(valid[None, :], E, -np.inf)
        E = E - _LN2 * np.floor(E.max(axis=1, keepdims=True) / _LN2)   # power-of-two row scale: max in [1,2)
        _np(e_new)[rows] = np.exp(E).astype(np.float32)
        if shp is not None:
            _np(shp)[rows] = np.where(valid[None, :], sh, 0)
        if rte is not None:
            _np(rte)[rows] = np.where(valid[None, :], rt, 0)
        if fac is not None:
            _np(fac)[rows] = fc
        if rs_prev is not None:
            _np(rs_prev)[rows] = _np(rs)[rows]
        _np(rs)[rows] = (f(add_rte) + fc.astype(np.float64).sum(axis=1)).astype(np.float32)
        cp[0] = fc.astype(np.float64).sum(axis=0).astype(np.float32)

    def row_finalize_ranges(self, acc, ranges, e_old, e_new, shp, rte, fac, rs, cs_other, cs_partial, prior_shp,
                            top_shp, add_rte, k, ld, acc_ld, rs_prev=None, e_new_ld=None):
        """ranges = [(rows, first acc row, first table row)]: acc / e_new are indexed by accumulator row, the tables
        by table row (hpf_hip_row_finalize_ranges_f32); e_new_ld: row stride of e_new (ld, or k when packed)."""
        e_ld = ld if e_new_ld is None else int(e_new_ld)
        total = np.zeros(ld, np.float64)
        for n, t0, r0 in ranges:
            if n <= 0:
                continue
            rows = torch.arange(r0, r0 + n, dtype=torch.int64)
            tmp_e = torch.zeros((int(e_old.shape[0]), ld), dtype=torch.float32)
            big = torch.zeros((int(e_old.shape[0]), acc_ld), dtype=torch.float32)
            big[r0:r0 + n] = acc[t0:t0 + n, :acc_ld]
            cp = torch.zeros_like(cs_partial)
            self.row_finalize(big, None, n, e_old, tmp_e, shp, rte, fac, rs, cs_other, cp, prior_shp, top_shp, add_rte,
                              k, ld, row_list=rows, part_ld=acc_ld, rs_prev=rs_prev)
            _np(e_new)[t0:t0 + n, :e_ld] = _np(tmp_e)[r0:r0 + n, :e_ld]
            total += _np(cp).astype(np.float64).sum(axis=0)
        cpo = _np(cs_partial)
        cpo[:] = 0
        cpo[0] = total.astype(np.float32)

    @staticmethod
    def gather_payload_ld(k):
        return ((int(k) + 1 + 3) // 4) * 4

    def item_shape_rows(self, acc, ranges, e_old, shp_out, send, rs, prior_shp, top_shp, k, ld, rs_prev=None):
        f = np.float32
        A, S, E, R, SH = _np(acc), _np(send), _np(e_old), _np(rs), _np(shp_out)
        for n, t0, r0 in ranges:
            if n <= 0:
                continue
            sh = (f(prior_shp) + E[r0:r0 + n, :k].astype(np.float64) * A[t0:t0 + n, :k]).astype(np.float32)
            P = sp.psi(sh.astype(np.float64))
            P = P - _LN2 * np.floor(P.max(axis=1, keepdims=True) / _LN2)       # power-of-two row scale: max in [1,2)
            S[t0:t0 + n] = 0
            S[t0:t0 + n, :k] = np.exp(P).astype(np.float32)
            S[t0:t0 + n, k] = f(top_shp) / R[r0:r0 + n]
            SH[t0:t0 + n] = 0
            SH[t0:t0 + n, :k] = sh
            if rs_prev is not None:
                _np(rs_prev)[r0:r0 + n] = R[r0:r0 + n]

    def item_apply_rows(self, recv, shp_own, e_tab, shp, fac, rs, cs_other, cs_partial, add_rte, k, ld, rank, world, nrows,
                        range_rows):
        f = np.float32
        Rv, So, Et, cs = _np(recv), _np(shp_own), _np(e_tab), _np(cs_other)[:k]
        total = sum((hi - lo) // world for lo, hi in range_rows)
        cp = _np(cs_partial)
        cp[:] = 0
        csum = np.zeros(ld, np.float64)
        t0 = 0
        for lo, hi in range_rows:
            m = (hi - lo) // world
            for q in range(world):
                r0, r1 = lo + q * m, min(lo + (q + 1) * m, nrows)
                if r1 <= r0:
                    continue
                n = r1 - r0
                src = Rv[q * total + t0: q * total + t0 + n]
                rt = (src[:, k:k + 1] + cs[None, :]).astype(np.float32)
                en = (src[:, :k] / rt).astype(np.float32)
                ex = np.floor(np.log2(en.max(axis=1, keepdims=True).astype(np.float64)))
                Et[r0:r1, :k] = (en * np.exp2(-ex).astype(np.float32)).astype(np.float32)
                Et[r0:r1, k:] = 0
                if q == rank:
                    sh = So[t0:t0 + n, :k]
                    fc = (sh / rt).astype(np.float32)
                    _np(rs)[r0:r1] = (f(add_rte) + fc.astype(np.float64).sum(axis=1)).astype(np.float32)
                    if shp is not None:
                        _np(shp)[r0:r1, :k] = sh
                    if fac is not None:
                        _np(fac)[r0:r1, :k] = fc
                    csum[:k] += fc.astype(np.float64).sum(axis=0)
            t0 += m
        cp[0] = csum.astype(np.float32)

    def unpack_rows(self, src, dst, nrows, k, ld):
        _np(dst)[:nrows, :k] = _np(src).reshape(-1)[: nrows * k].reshape(nrows, k)

    def colsum_reduce(self, cs_partial, cs_out, ld):
        _np(cs_out)[:] = _np(cs_partial).astype(np.float64).sum(axis=0).astype(np.float32)

    def colsum_sequential(self, tab, nrows, ld, cs_out):
        _np(cs_out)[:] = _np(tab)[:nrows].sum(axis=0)        # (numpy's own float32 row-after-row order: the reference's)

    def colsum(self, tab, nrows, ld, cs_partial):
        cp = _np(cs_partial)
        cp[:] = 0
        cp[0] = _np(tab)[:nrows].astype(np.float64).sum(axis=0).astype(np.float32)

    def expect(self, shp, rte, e, nrows, k, ld, row_list=None, flag=None, factored=None, rte_out=None):
        rows = np.arange(nrows) if row_list is None else _np(row_list)[:nrows].astype(np.int64)
        if flag is not None:
            rows = rows[_np(flag)[rows] != 0]
        if rows.shape[0] == 0:
            return
        valid = np.arange(ld) < k
        if factored is not None:        # rte = top / rs[r] + cs[c] in float32, as the kernels form it
            rs, cs, top = factored
            R = (np.float32(top) / _np(rs)[rows][:, None] + _np(cs)[None, :]).astype(np.float32)
            R[:, k:] = 1.0
            if rte_out is not None:
                _np(rte_out)[rows, :k] = R[:, :k]
                _np(rte_out)[rows, k:] = 0
        else:
            R = _np(rte)[rows]
        with np.errstate(divide="ignore", invalid="ignore"):
            E = sp.psi(_np(shp).astype(np.float64)[rows]) - np.log(R.astype(np.float64))
        E = np.where(valid[None, :], E, -np.inf)
        E = E - _LN2 * np.floor(E.max(axis=1, keepdims=True) / _LN2)   # power-of-two row scale: max in [1,2)
        _np(e)[rows] = np.exp(E).astype(np.float32)

    def segsum(self, part, row_seg_ptr, nrows, acc, ld, row_list=None, acc_ld=None, acc_by_row=False):
        acc_ld = ld if acc_ld is None else acc_ld
        rows = np.arange(nrows) if row_list is None else _np(row_list)[:nrows].astype(np.int64)
        rsp = _np(row_seg_ptr)
        P = _np(part).astype(np.float64)
        out = np.zeros((rows.shape[0], ld))
        for t, r in enumerate(rows):
            if rsp[r + 1] > rsp[r]:
                out[t] = P[rsp[r]: rsp[r + 1]].sum(axis=0)
        if acc_by_row:
            _np(acc)[rows] = out[:, :acc_ld].astype(np.float32)
        else:
            _np(acc)[: rows.shape[0]] = out[:, :acc_ld].astype(np.float32)

    def pair_llk(self, T, B, ix_u, ix_i, y, k, ld, full_llk):
        Tn = _np(T).astype(np.float64)[_np(ix_u).astype(np.int64)]
        Bn = _np(B).astype(np.float64)[_np(ix_i).astype(np.int64)]
        yy = _np(y).astype(np.float64)
        yhat = (Tn * Bn).sum(axis=1)
        a0 = (yy * np.log(yhat)).sum()
        if full_llk:
            a0 -= sp.gammaln(yy + 1.0).sum()
        return torch.tensor([a0, ((yy - yhat) ** 2).sum(), yhat.sum()], dtype=torch.float64)

    def llk_sweep(self, side, T, B, k, ld, full_llk):
        rows = torch.repeat_interleave(torch.arange(side.nrows), side.indptr[1:] - side.indptr[:-1]).to(torch.int32)
        return self.pair_llk(T, B, rows, side.idx, side.y, k, ld, full_llk)

    def pair_dot(self, T, B, ix_u, ix_i, out, k, ld):
        Tn = _np(T).astype(np.float64)[_np(ix_u).astype(np.int64)]
        Bn = _np(B).astype(np.float64)[_np(ix_i).astype(np.int64)]
        _np(out)[:] = (Tn * Bn).sum(axis=1).astype(np.float32)

    def score_rows(self, vec, tab, out, k, ld):
        _np(out)[:] = (_np(tab).astype(np.float64) @ _np(vec).astype(np.float64)).astype(np.float32)

    def fold_in(self, idx, y, e_items, cs_other, shp, rte, fac, e_last, rounds, prior, top, add, rs, stop_thr, maxiter,
                k, ld):
        """hpf_hip_fold_in_f32: the rounds of PXI:505-513 in float32 numpy (phi-sums accumulated in float64)."""
        f = np.float32
        ids, yv = _np(idx).astype(np.int64), _np(y).astype(np.float64)
        EB = _np(e_items)[ids].astype(np.float64)[:, :k]
        gs, gr, th_prev = _np(shp), _np(rte), _np(fac)[:k].copy()
        cs = _np(cs_other)[:k]
        rs, it = f(rs), 0
        e_row = torch.zeros((1, ld), dtype=torch.float32)
        th = th_prev.copy()
        while it < maxiter:
            self.expect(shp.reshape(1, -1), rte.reshape(1, -1), e_row, 1, k, ld)
            et = _np(e_row)[0, :k].astype(np.float64)
            s = EB @ et
            w = np.where(yv > 0, yv / s, 0.0)
            acc = (w[:, None] * EB).sum(axis=0).astype(np.float32)
            gr[:k] = f(top) / rs + cs
            gs[:k] = f(prior) + _np(e_row)[0, :k] * acc
            th = gs[:k] / gr[:k]
            rs = f(add) + th.sum(dtype=np.float32)
            it += 1
            if float(np.sqrt(((th - th_prev).astype(np.float32) ** 2).sum(dtype=np.float32))) < stop_thr:
                break
            th_prev = th.copy()
        _np(fac)[:k] = th
        _np(e_last)[:] = _np(e_row)[0]
        _np(rounds)[0] = it

    def svi_prep_scratch_words(self):
        return 8

    @staticmethod
    def _batch_structures(own, oth, cap, flag):
        """The structures of the batch made of the rows with flag != 0 (numpy): own side's kept descriptors + split-row
        list, other side's filtered nonzeros, present rows, segments cut at cap, split-row list."""
        segs = _np(own.segs)
        row = (segs[:, 1] >> 32).astype(np.int64)
        keep = flag[row] != 0
        kept = segs[keep]
        krow = row[keep]
        rsp = _np(own.row_seg_ptr)
        opens = np.nonzero(((kept[:, 1] & 0x40000000) == 0) & (np.concatenate([[True], krow[1:] != krow[:-1]])))[0]
        bm = np.stack([opens, rsp[krow[opens] + 1] - rsp[krow[opens]], krow[opens]], axis=1).astype(np.int64)
        optr = _np(oth.indptr)
        mask = flag[_np(oth.idx).astype(np.int64)] != 0
        row_of = np.repeat(np.arange(oth.nrows), optr[1:] - optr[:-1])
        o_idx, o_y = _np(oth.idx)[mask], _np(oth.y)[mask]
        cnt = np.bincount(row_of[mask], minlength=oth.nrows).astype(np.int64)
        rows = np.nonzero(cnt > 0)[0]
        c = cnt[rows]
        start = np.cumsum(c) - c
        ns = (c + cap - 1) // cap
        sg0 = np.cumsum(ns) - ns
        nseg = int(ns.sum())
        local = np.repeat(np.arange(rows.shape[0]), ns)
        within = np.arange(nseg) - sg0[local]
        begin = start[local] + within * cap
        length = np.minimum(c[local] - within * cap, cap) | np.where(ns[local] == 1, 0x40000000, 0)
        osegs = np.stack([begin, length | (rows[local] << 32)], axis=1).astype(np.int64)
        multi = np.nonzero(ns > 1)[0]
        om = np.stack([sg0[multi], ns[multi], rows[multi]], axis=1).astype(np.int64)
        present = np.zeros(oth.nrows, np.uint8)            # 1: present in one segment, 2: a split row
        present[rows] = np.where(ns > 1, 2, 1)
        return kept, bm, o_idx, o_y, present, osegs, om, rows.shape[0]

    def svi_batch_prepare(self, ws):
        """hpf_hip_svi_batch_prepare in numpy: same outputs in the same layout (tests compare them with the kernels')."""
        own, oth, cap = ws.own, ws.oth, int(ws.seg_cap)
        flag = _np(ws.flag_own)
        if ws.prev_ids is not None:
            flag[_np(ws.prev_ids)] = 0
        ids = _np(ws.ids)
        rsp_own = _np(own.row_seg_ptr)
        flag[ids] = np.where(rsp_own[ids + 1] - rsp_own[ids] == 1, 1, 2)      # 2: a split row or a row without nonzeros
        ptr = _np(own.indptr)
        empty = ids[ptr[ids + 1] == ptr[ids]]
        _np(ws.acc_own)[empty] = 0
        sizes = _np(ws.sizes)
        sizes[:] = 0
        kept, bm, o_idx, o_y, present, osegs, om, nrows_present = self._batch_structures(own, oth, cap, flag)
        assert kept.shape[0] <= ws.b_cap and o_idx.shape[0] <= ws.o_cap and osegs.shape[0] <= ws.o_segs_cap
        _np(ws.b_segs)[: kept.shape[0]] = kept
        _np(ws.b_multi)[: bm.shape[0]] = bm
        _np(ws.o_idx)[: o_idx.shape[0]] = o_idx
        _np(ws.o_y)[: o_idx.shape[0]] = o_y
        _np(ws.flag_oth)[:] = present
        _np(ws.o_segs)[: osegs.shape[0]] = osegs
        _np(ws.o_multi)[: om.shape[0]] = om
        sizes[:6] = kept.shape[0], bm.shape[0], osegs.shape[0], om.shape[0], o_idx.shape[0], nrows_present

    def svi_epoch_scratch_words(self, nb):
        return 8

    def svi_epoch_prepare(self, ws):
        """hpf_hip_svi_epoch_prepare in numpy: every batch of the epoch through the per-batch construction, the batches'
        nonzeros laid end to end in e_idx / e_y (the other side's segments index them as a whole)."""
        own, oth, cap, nb, per = ws.own, ws.oth, int(ws.seg_cap), int(ws.nb), int(ws.per)
        order = _np(ws.order)
        ptr = _np(own.indptr)
        _np(ws.acc_own)[order[ptr[order + 1] == ptr[order]]] = 0
        _np(ws.flag_own)[:] = 0
        sizes = _np(ws.sizes)
        base = 0
        for b in range(nb):
            ids = order[b * per: min(own.nrows, (b + 1) * per)]
            flag = _np(ws.flag_own)[b]
            rsp_own = _np(own.row_seg_ptr)
            flag[ids] = np.where(rsp_own[ids + 1] - rsp_own[ids] == 1, 1, 2)
            _np(ws.batch_of)[ids] = b
            kept, bm, o_idx, o_y, present, osegs, om, nrows_present = self._batch_structures(own, oth, cap, flag)
            assert kept.shape[0] <= ws.b_cap and osegs.shape[0] <= ws.o_segs_cap
            osegs = osegs.copy()
            osegs[:, 0] += base
            _np(ws.b_segs)[b, : kept.shape[0]] = kept
            _np(ws.b_multi)[b, : bm.shape[0]] = bm
            _np(ws.e_idx)[base: base + o_idx.shape[0]] = o_idx
            _np(ws.e_y)[base: base + o_idx.shape[0]] = o_y
            _np(ws.flag_oth)[b] = present
            _np(ws.o_segs)[b, : osegs.shape[0]] = osegs
            _np(ws.o_multi)[b, : om.shape[0]] = om
            sizes[b, :6] = kept.shape[0], bm.shape[0], osegs.shape[0], om.shape[0], o_idx.shape[0], nrows_present
            base += o_idx.shape[0]

    def svi_coo_narrow(self, ids, limit, out, err):
        a = _np(ids)
        bad = (a < 0) | (a >= limit)
        _np(out)[:] = np.where(bad, 0, a).astype(np.int32)
        if bad.any():
            _np(err)[0] = 1

    def svi_coo_prepare(self, key, nrows, seg_cap, flag, row_start, row_cnt, segs, multi, sizes, tiles):
        """hpf_hip_svi_coo_prepare in numpy: the layout of one grouping of a COO batch from its sorted row ids."""
        kk, cap = _np(key).astype(np.int64), int(seg_cap)
        n = kk.shape[0]
        cnt = np.bincount(kk, minlength=nrows).astype(np.int64)
        rows = np.nonzero(cnt > 0)[0]
        c = cnt[rows]
        start = np.cumsum(c) - c
        ns = (c + cap - 1) // cap
        sg0 = np.cumsum(ns) - ns
        nseg = int(ns.sum())
        local = np.repeat(np.arange(rows.shape[0]), ns)
        within = np.arange(nseg) - sg0[local]
        begin = start[local] + within * cap
        length = np.minimum(c[local] - within * cap, cap) | np.where(ns[local] == 1, 0x40000000, 0)
        sg = np.stack([begin, length | (rows[local] << 32)], axis=1).astype(np.int64)
        mi = np.nonzero(ns > 1)[0]
        om = np.stack([sg0[mi], ns[mi], rows[mi]], axis=1).astype(np.int64)
        f = _np(flag)
        f[:nrows] = 0
        f[rows] = np.where(ns > 1, 2, 1)
        sz = _np(sizes)
        sz[:7] = 0
        if nseg > segs.shape[0] or om.shape[0] > multi.shape[0]:
            sz[7] = 1
        _np(segs)[: min(nseg, segs.shape[0])] = sg[: segs.shape[0]]
        _np(multi)[: min(om.shape[0], multi.shape[0])] = om[: multi.shape[0]]
        sz[2], sz[3], sz[4], sz[5] = min(nseg, segs.shape[0]), min(om.shape[0], multi.shape[0]), n, rows.shape[0]

    def segsum_desc(self, part, desc, ndesc_dev, ndesc_max, acc, ld):
        nd = min(int(ndesc_max), int(_np(ndesc_dev)[0]))
        P, D = _np(part).astype(np.float64), _np(desc)
        for d in range(nd):
            first, n, r = int(D[d, 0]), int(D[d, 1]), int(D[d, 2])
            _np(acc)[r] = P[first: first + n].sum(axis=0).astype(np.float32)

    def mt19937_words(self, state, raw):
        """numpy's own generator positioned at `state`; its outputs, un-tempered, are the stream's state words."""
        st = _np(state).view(np.uint32)
        bg = np.random.MT19937()
        bg.state = {"bit_generator": "MT19937", "state": {"key": st[:624].copy(), "pos": int(st[624])}}
        y = bg.random_raw(raw.numel()).astype(np.uint32)
        new = bg.state["state"]
        st[:624] = new["key"]
        st[624] = new["pos"]
        y ^= y >> 18
        y ^= (y << 15) & np.uint32(0xefc60000)
        t = y.copy()
        for _ in range(5):
            t = y ^ ((t << 7) & np.uint32(0x9d2c5680))
        y = t
        for _ in range(3):
            t = y ^ (t >> 11)
        _np(raw).view(np.uint32)[:] = t

    def uniform_rows(self, raw, out, nrows, k, ld, base, scale, den=None, ratio=None):
        y = _np(raw).view(np.uint32)[: nrows * k].copy()
        y ^= y >> 11
        y ^= (y << 7) & np.uint32(0x9d2c5680)
        y ^= (y << 15) & np.uint32(0xefc60000)
        y ^= y >> 18
        u = (y >> 8).astype(np.float32) * np.float32(1.0 / 16777216.0)
        vals = (np.float32(base) + np.float32(scale) * u).reshape(nrows, k)
        _np(out)[:nrows, :k] = vals
        if ratio is not None:
            _np(ratio)[:nrows, :k] = vals / _np(den)[:nrows, :k]

    # -- stochastic-VI row kernels (float32 arithmetic, statement for statement) ------------------
    def refresh_grid(self, nrows):
        return self.finalize_grid(nrows)

    def svi_shape_rows(self, row_list, acc, e, shp, prior, w_new, w_old, k, ld, acc_by_row=False):
        rows = _np(row_list).astype(np.int64)
        if rows.shape[0] == 0:
            return
        f = np.float32
        if acc is None:
            a = np.zeros((rows.shape[0], k), np.float32)
        else:
            a = _np(acc)[rows, :k] if acc_by_row else _np(acc)[: rows.shape[0], :k]
        fresh = (f(prior) + _np(e)[rows, :k] * a).astype(np.float32)
        S = _np(shp)
        if w_old == 0:
            S[rows, :k] = f(w_new) * fresh
        else:
            S[rows, :k] = f(w_new) * fresh + f(w_old) * S[rows, :k]

    def svi_refresh(self, nrows, shp, rte, fac, rs, cs_other, cs_partial, top, add, step, step_prev, refresh_rte,
                    blend_rs, k, ld):
        f = np.float32
        R = _np(rte)
        if refresh_rte:
            R[:, :k] = f(top) / _np(rs)[:, None] + _np(cs_other)[None, :k]
        F = _np(fac)
        F[:, :k] = _np(shp)[:, :k] / R[:, :k]
        F[:, k:] = 0
        if blend_rs:
            _np(rs)[:] = f(step) * (f(add) + F[:, :k].sum(axis=1)) + f(step_prev) * _np(rs)
        cp = _np(cs_partial)
        cp[:] = 0
        cp[0] = F.astype(np.float64).sum(axis=0).astype(np.float32)

    sweep_blocks = 1          # (rows of partial column sums the fused sweep writes)

    def sweep_svi(self, side, tab_self, tab_other, part, e_new, shp, rte, fac, rs, cs_other, cs_partial, prior, w_new,
                  w_old, top, add, step, step_prev, k, ld):
        """hpf_hip_sweep_svi_f32: the plain sweep, then the flagged-row statements of svi_side (rate_mode 1, rs_mode 1)
        for the rows present in ONE segment, on a compacted copy of their rows (so that only they enter the column sums)."""
        nseg = _live_nseg(side)
        _np(cs_partial)[:] = 0
        if nseg == 0:
            return
        acc = torch.zeros_like(shp)
        self.sweep(side, tab_self, tab_other, part, k, ld, acc_rows=acc, acc_ld=ld)
        begin, length, row = _decode_segs(side)
        whole = (_np(side.segs)[:nseg, 1] & 0x40000000) != 0
        rows = torch.from_numpy(np.sort(row[whole]).astype(np.int64))
        if rows.shape[0] == 0:
            return
        sub = {n: t[rows].clone() for n, t in (("acc", acc), ("e", tab_self), ("shp", shp), ("rte", rte), ("rs", rs))}
        fac_s = torch.zeros_like(sub["shp"])
        ones = torch.ones(rows.shape[0], dtype=torch.uint8)
        e_s = sub["e"].clone() if e_new is not None else None
        self.svi_side(rows.shape[0], ones, sub["acc"], sub["e"], sub["shp"], sub["rte"], fac_s, sub["rs"], cs_other,
                      cs_partial, prior, w_new, w_old, top, add, step, step_prev, 1, 1, k, ld, e_out=e_s)
        shp[rows], rte[rows], rs[rows] = sub["shp"], sub["rte"], sub["rs"]
        if fac is not None:
            fac[rows] = fac_s
        if e_new is not None:
            e_new[rows] = e_s

    def sweep_svi_batch(self, side, e_self, tab_other, part, shp, rte_in, rte_out, fac, rs, rs_prev_out, factored, cs_other,
                        cs_partial, prior, w_new, w_old, top, add, step, step_prev, k, ld):
        """hpf_hip_sweep_svi_batch_f32: expect over the rows that have a segment (prologue), the plain sweep, then the
        flagged-row statements of svi_side (rate_mode 0, rs_mode 1) for the rows present in ONE segment, on a compacted
        copy of their rows (so that only they enter the column sums)."""
        nseg = _live_nseg(side)
        _np(cs_partial)[:] = 0
        if nseg == 0:
            return
        begin, length, row = _decode_segs(side)
        rows_all = torch.from_numpy(np.unique(row).astype(np.int64))
        self.expect(shp, rte_in, e_self, int(rows_all.shape[0]), k, ld, row_list=rows_all, factored=factored)
        acc = torch.zeros_like(shp)
        self.sweep(side, e_self, tab_other, part, k, ld, acc_rows=acc, acc_ld=ld)
        whole = (_np(side.segs)[:nseg, 1] & 0x40000000) != 0
        rows = torch.from_numpy(np.sort(row[whole]).astype(np.int64))
        if rows.shape[0] == 0:
            return
        sub = {n: t[rows].clone() for n, t in (("acc", acc), ("e", e_self), ("shp", shp), ("rs", rs))}
        rte_s, fac_s = torch.zeros_like(sub["shp"]), torch.zeros_like(sub["shp"])
        rsp_s = torch.zeros(rows.shape[0], dtype=torch.float32)
        ones = torch.ones(rows.shape[0], dtype=torch.uint8)
        self.svi_side(rows.shape[0], ones, sub["acc"], sub["e"], sub["shp"], rte_s, fac_s, sub["rs"], cs_other, cs_partial,
                      prior, w_new, w_old, top, add, step, step_prev, 0, 1, k, ld, rs_prev_out=rsp_s)
        shp[rows], rs[rows] = sub["shp"], sub["rs"]
        if rte_out is not None:
            rte_out[rows] = rte_s
        if fac is not None:
            fac[rows] = fac_s
        if rs_prev_out is not None:
            rs_prev_out[rows] = rsp_s

    def svi_side(self, nrows, flag, acc, e, shp, rte, fac, rs, cs_other, cs_partial, prior, w_new, w_old, top, add, step,
                 step_prev, rate_mode, rs_mode, k, ld, rs_rate=None, rs_prev_out=None, e_out=None, done_flag=0):
        """hpf_hip_svi_side_f32 = the separate stand-in statements in the reference's order.  rte / fac None: computed
        into scratch tables and dropped; rs_rate / rs_prev_out: the factored-rate plumbing of the lazy epochs; e_out: the
        flagged rows' new E rows (= expect over them afterwards); done_flag: rows whose flag equals it are left out
        altogether (the pass runs on a compacted copy of the others)."""
        if done_flag:
            assert flag is not None and rs_rate is None
            rows = torch.nonzero(flag[:nrows] != done_flag).reshape(-1)
            _np(cs_partial)[:] = 0
            if rows.shape[0] == 0:
                return
            sub = {n: t[rows].clone() for n, t in (("acc", acc), ("e", e), ("shp", shp), ("rs", rs))}
            rte_s = rte[rows].clone() if rte is not None else None
            fac_s = torch.zeros_like(sub["shp"])
            e_s = sub["e"].clone() if e_out is not None else None
            rsp_s = torch.zeros(rows.shape[0], dtype=torch.float32) if rs_prev_out is not None else None
            self.svi_side(rows.shape[0], (flag[rows] != 0).to(torch.uint8), sub["acc"], sub["e"], sub["shp"], rte_s,
                          fac_s, sub["rs"], cs_other, cs_partial, prior, w_new, w_old, top, add, step, step_prev, rate_mode,
                          rs_mode, k, ld, rs_prev_out=rsp_s, e_out=e_s)
            shp[rows], rs[rows] = sub["shp"], sub["rs"]
            if rte is not None:
                rte[rows] = rte_s
            if fac is not None:
                fac[rows] = fac_s
            if e_out is not None:
                e_out[rows] = e_s
            if rs_prev_out is not None:
                rs_prev_out[rows] = rsp_s
            return
        rows = torch.nonzero(flag[:nrows] != 0).reshape(-1) if flag is not None else torch.empty(0, dtype=torch.int64)
        self.svi_shape_rows(rows, acc, e, shp, prior, w_new, w_old, k, ld, acc_by_row=True)
        rte_t = rte if rte is not None else torch.zeros_like(shp)
        fac_t = fac if fac is not None else torch.zeros_like(shp)
        if rate_mode == 1:
            self.svi_rate_rows(rows, rte_t, None, rs, cs_other, top, 0.0, step, step_prev, 0, k, ld)
        rs_before = rs.clone()
        if rs_prev_out is not None:
            _np(rs_prev_out)[:nrows] = _np(rs_rate if rs_rate is not None else rs)[:nrows]
        rs_used = rs if rs_rate is None else rs_rate.clone()
        self.svi_refresh(nrows, shp, rte_t, fac_t, rs_used, cs_other if rate_mode == 0 else None, cs_partial, top, add,
                         step, step_prev, rate_mode == 0, rs_mode == 2 and rs_rate is None, k, ld)
        assert not (rs_mode == 2 and rs_rate is not None)
        if rs_mode == 1:
            assert torch.equal(rs, rs_before)
            self.svi_rate_rows(rows, None, fac_t, rs, None, 0.0, add, step, step_prev, 1, k, ld)
        if e_out is not None:
            self.expect(shp, rte_t, e_out, nrows, k, ld, flag=flag)

    def svi_rate_rows(self, row_list, rte, fac, rs, cs_other, top, add, step, step_prev, mode, k, ld):
        rows = _np(row_list).astype(np.int64)
        if rows.shape[0] == 0:
            return
        f = np.float32
        if mode == 0:
            R = _np(rte)
            R[rows, :k] = f(step) * (f(top) / _np(rs)[rows][:, None] + _np(cs_other)[None, :k]) + f(step_prev) * R[rows, :k]
        else:
            _np(rs)[rows] = f(step) * (f(add) + _np(fac)[rows, :k].sum(axis=1)) + f(step_prev) * _np(rs)[rows]
