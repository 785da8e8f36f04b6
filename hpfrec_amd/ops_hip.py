"""HipOps: the kernel launchers of include/hpf_hip.h on torch device tensors.

This is the only implementation of the op set that ships.  (tests/cpu_ops.py holds a
numpy stand-in with the same method names so that host logic -- layouts, sharding, the
driver loop, the HPF class -- can be exercised on machines without a GPU; it is never
importable from this package.)
"""
import os

import torch

from . import _lib


def _ptr(t):
    return None if t is None else t.data_ptr()


def _svi_batch_struct():
    import ctypes
    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32

    class SviBatchDesc(ctypes.Structure):
        """hpf_svi_batch (include/hpf_hip.h), field for field."""
        _fields_ = [("own_segs", vp), ("own_nseg", i64), ("own_row_seg_ptr", vp), ("own_indptr", vp), ("own_nrows", i64),
                    ("oth_idx", vp), ("oth_y", vp), ("oth_indptr", vp), ("oth_nrows", i64), ("oth_nnz", i64),
                    ("ids", vp), ("nids", i64), ("prev_ids", vp), ("nprev", i64),
                    ("flag_own", vp), ("flag_oth", vp), ("acc_own", vp), ("ld", i32), ("seg_cap", i32),
                    ("b_segs", vp), ("b_segs_cap", i64), ("b_multi", vp), ("multi_cap", i64),
                    ("o_idx", vp), ("o_y", vp), ("o_cap", i64), ("o_segs", vp), ("o_segs_cap", i64), ("o_multi", vp),
                    ("sizes", vp), ("mask", vp), ("chunk_pre", vp), ("tile_cnt", vp), ("tile_off", vp), ("row_start", vp), ("row_cnt", vp),
                    ("tiles", vp), ("flag_bits", vp)]
    return SviBatchDesc


SviBatchDesc = _svi_batch_struct()


def _svi_epoch_struct():
    import ctypes
    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32

    class SviEpochDesc(ctypes.Structure):
        """hpf_svi_epoch (include/hpf_hip.h), field for field."""
        _fields_ = [("own_segs", vp), ("own_nseg", i64), ("own_row_seg_ptr", vp), ("own_indptr", vp), ("own_nrows", i64),
                    ("oth_segs", vp), ("oth_nseg", i64), ("oth_row_seg_ptr", vp),
                    ("oth_idx", vp), ("oth_y", vp), ("oth_nrows", i64), ("oth_nnz", i64),
                    ("order", vp), ("per", i64), ("nb", i32), ("ld", i32), ("seg_cap", i32), ("reserved", i32),
                    ("acc_own", vp), ("batch_of", vp), ("flag_own", vp), ("flag_oth", vp),
                    ("b_segs", vp), ("b_segs_cap", i64), ("b_multi", vp), ("multi_cap", i64),
                    ("e_idx", vp), ("e_y", vp), ("o_segs", vp), ("o_segs_cap", i64), ("o_multi", vp),
                    ("sizes", vp), ("key", vp), ("seg_cnt", vp), ("seg_pos", vp), ("tiles", vp)]
    return SviEpochDesc


SviEpochDesc = _svi_epoch_struct()


def _svi_coo_struct():
    import ctypes
    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32

    class SviCooDesc(ctypes.Structure):
        """hpf_svi_coo (include/hpf_hip.h), field for field."""
        _fields_ = [("key", vp), ("n", i64), ("nrows", i64), ("seg_cap", i32), ("reserved", i32),
                    ("flag", vp), ("row_start", vp), ("row_cnt", vp),
                    ("segs", vp), ("segs_cap", i64), ("multi", vp), ("multi_cap", i64), ("sizes", vp), ("tiles", vp)]
    return SviCooDesc


SviCooDesc = _svi_coo_struct()


class HipOps:
    name = "hip"

    def __init__(self, device=None):
        if not torch.cuda.is_available():
            raise _lib.HpfHipError("hpfrec_amd: no ROCm device visible (torch.cuda.is_available() is False); "
                                   "the HIP path has no CPU fallback")
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self.L = _lib.lib()
        with torch.cuda.device(self.device):
            self.cu_count, self.arch = _lib.device_info()
        # memory-bound grid: blocks per CU (a multiple of the 4 resident ones), grid-stride over the rest (CDNA guide,
        # guideline 11); 16 rather than 8 evens out the tail: +0.8 % at C3 (6 -- not a multiple -- costs 6 %)
        self.sweep_blocks = max(1, self.cu_count) * int(os.environ.get("HPF_SWEEP_BPC", "16"))
        self.finalize_blocks = max(1, self.cu_count) * int(os.environ.get("HPF_FIN_BPC", "4"))
        self.refresh_blocks = max(1, self.cu_count) * int(os.environ.get("HPF_REFRESH_BPC", "8"))

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    # -- every method mirrors one C entry point ------------------------------------------
    def sweep(self, side, tab_self, tab_other, part, k, ld, acc_rows=None, acc_ld=0, grid_blocks=None):
        """`side.nseg_dev` (optional, device int64[1]): the live segment count of a batch built on the device; side.nseg
        is then the capacity of its segment list."""
        _lib.check(self.L.hpf_hip_sweep_f32(_ptr(side.segs), side.nseg, _ptr(side.idx), _ptr(side.y),
                                            _ptr(tab_self), _ptr(tab_other), _ptr(part), _ptr(acc_rows), int(acc_ld),
                                            k, ld, int(getattr(side, "short_rows", 0)),
                                            grid_blocks or self.sweep_blocks, _ptr(getattr(side, "nseg_dev", None)),
                                            self._stream()),
                   "hpf_hip_sweep_f32")

    def sweep_grid(self, nseg, blocks=None):
        return int(max(1, min(blocks or self.sweep_blocks, (nseg + 3) // 4)))

    def sweep_finalize(self, side, tab_self, tab_other, part, e_new, shp, rte, fac, rs, cs_other, cs_partial,
                       prior_shp, top_shp, add_rte, k, ld, rs_prev=None):
        """sweep + fused row finalize of single-segment rows; cs_partial must have sweep_grid(nseg) rows."""
        _lib.check(self.L.hpf_hip_sweep_finalize_f32(_ptr(side.segs), side.nseg, _ptr(side.idx), _ptr(side.y),
                                                     _ptr(tab_self), _ptr(tab_other), _ptr(part), _ptr(e_new),
                                                     _ptr(shp), _ptr(rte), _ptr(fac), _ptr(rs), _ptr(rs_prev),
                                                     _ptr(cs_other), _ptr(cs_partial), float(prior_shp),
                                                     float(top_shp), float(add_rte), k, ld, cs_partial.shape[0],
                                                     self._stream()), "hpf_hip_sweep_finalize_f32")

    def finalize_grid(self, nrows):
        return int(max(1, min(self.finalize_blocks, (nrows + 3) // 4)))

    def row_finalize(self, part, row_seg_ptr, nrows, e_old, e_new, shp, rte, fac, rs, cs_other, cs_partial,
                     prior_shp, top_shp, add_rte, k, ld, row_list=None, part_ld=None, rs_prev=None):
        grid = cs_partial.shape[0]
        _lib.check(self.L.hpf_hip_row_finalize_f32(_ptr(part), _ptr(row_seg_ptr), _ptr(row_list), nrows, _ptr(e_old),
                                                   _ptr(e_new),
                                                   _ptr(shp), _ptr(rte), _ptr(fac), _ptr(rs), _ptr(rs_prev),
                                                   _ptr(cs_other), _ptr(cs_partial), float(prior_shp), float(top_shp),
                                                   float(add_rte), k, ld, ld if part_ld is None else part_ld, grid,
                                                   self._stream()),
                   "hpf_hip_row_finalize_f32")

    def row_finalize_ranges(self, acc, ranges, e_old, e_new, shp, rte, fac, rs, cs_other, cs_partial, prior_shp,
                            top_shp, add_rte, k, ld, acc_ld, rs_prev=None, e_new_ld=None):
        """Dense finalize of several row ranges in one launch; ranges = [(rows, first acc row, first table row)];
        e_new_ld: row stride of e_new (ld, or k for a packed all-gather send buffer)."""
        import ctypes
        n = len(ranges)
        arr = (ctypes.c_int64 * n)
        rows, t0, r0 = (arr(*[int(r[i]) for r in ranges]) for i in range(3))
        _lib.check(self.L.hpf_hip_row_finalize_ranges_f32(
            _ptr(acc), n, ctypes.addressof(rows), ctypes.addressof(t0), ctypes.addressof(r0), _ptr(e_old), _ptr(e_new),
            _ptr(shp), _ptr(rte), _ptr(fac), _ptr(rs), _ptr(rs_prev), _ptr(cs_other), _ptr(cs_partial),
            float(prior_shp), float(top_shp), float(add_rte), k, ld, int(acc_ld), int(ld if e_new_ld is None else e_new_ld),
            cs_partial.shape[0], self._stream()),
            "hpf_hip_row_finalize_ranges_f32")

    @staticmethod
    def gather_payload_ld(k):
        """Row stride of the gather-early all-gather payload: k numerators + the base rate, rounded up to 4 floats."""
        return ((int(k) + 1 + 3) // 4) * 4

    def item_shape_rows(self, acc, ranges, e_old, shp_out, send, rs, prior_shp, top_shp, k, ld, rs_prev=None):
        """Part 1 of the split item finalizer (gather-early exchange): shapes (from the packed [.][k] `acc`) into the
        padded `shp_out` [.][ld], the [exp(psi(shp)) row-scaled | top_shp / rs | 0..] rows into `send`
        ([.][gather_payload_ld(k)]); ranges as in row_finalize_ranges."""
        import ctypes
        n = len(ranges)
        arr = (ctypes.c_int64 * n)
        rows, t0, r0 = (arr(*[int(r[i]) for r in ranges]) for i in range(3))
        grid = self.finalize_grid(sum(int(r[0]) for r in ranges))
        _lib.check(self.L.hpf_hip_item_shape_rows_f32(
            _ptr(acc), n, ctypes.addressof(rows), ctypes.addressof(t0), ctypes.addressof(r0), _ptr(e_old), _ptr(shp_out),
            _ptr(send), _ptr(rs), _ptr(rs_prev), float(prior_shp), float(top_shp), k, ld, grid, self._stream()),
            "hpf_hip_item_shape_rows_f32")

    def item_apply_rows(self, recv, shp_own, e_tab, shp, fac, rs, cs_other, cs_partial, add_rte, k, ld, rank, world, nrows,
                        range_rows):
        """Part 2: E rows of ALL items from the gathered [numerators | base] rows and colsum(Theta) (cs_other); the rows
        this rank owns also get their means, scalar rates, optional shape / mean stores and colsum partials.
        range_rows = [(lo, hi)] in issue order."""
        import ctypes
        n = len(range_rows)
        arr = (ctypes.c_int64 * n)
        lo, hi = arr(*[int(r[0]) for r in range_rows]), arr(*[int(r[1]) for r in range_rows])
        _lib.check(self.L.hpf_hip_item_apply_rows_f32(
            _ptr(recv), _ptr(shp_own), _ptr(e_tab), _ptr(shp), _ptr(fac), _ptr(rs), _ptr(cs_other), _ptr(cs_partial),
            float(add_rte), k, ld, int(rank), int(world), int(nrows), n, ctypes.addressof(lo), ctypes.addressof(hi),
            cs_partial.shape[0], self._stream()), "hpf_hip_item_apply_rows_f32")

    def colsum_reduce(self, cs_partial, cs_out, ld):
        _lib.check(self.L.hpf_hip_colsum_reduce_f32(_ptr(cs_partial), cs_partial.shape[0], _ptr(cs_out), ld,
                                                    self._stream()), "hpf_hip_colsum_reduce_f32")

    def colsum(self, tab, nrows, ld, cs_partial):
        _lib.check(self.L.hpf_hip_colsum_f32(_ptr(tab), nrows, ld, _ptr(cs_partial), cs_partial.shape[0],
                                             self._stream()), "hpf_hip_colsum_f32")

    def colsum_sequential(self, tab, nrows, ld, cs_out):
        """cs_out = tab[:nrows].sum(axis=0) in numpy's order (float32, row after row): the reference's sums, bit for bit."""
        _lib.check(self.L.hpf_hip_colsum_sequential_f32(_ptr(tab), int(nrows), ld, _ptr(cs_out), self._stream()),
                   "hpf_hip_colsum_sequential_f32")

    def expect(self, shp, rte, e, nrows, k, ld, row_list=None, flag=None, factored=None, rte_out=None):
        """flag (uint8 per table row): only rows with a non-zero flag.  factored = (rs, cs, top): the rate is
        top / rs[r] + cs[c] instead of a table (rte may be None); rte_out: that rate is also stored as table rows."""
        rs, cs, top = factored if factored is not None else (None, None, 0.0)
        _lib.check(self.L.hpf_hip_expect_f32(_ptr(shp), _ptr(rte), _ptr(e), _ptr(row_list), _ptr(flag), nrows, k, ld,
                                             _ptr(rs), _ptr(cs), float(top), _ptr(rte_out), self._stream()),
                   "hpf_hip_expect_f32")

    def segsum(self, part, row_seg_ptr, nrows, acc, ld, row_list=None, acc_ld=None, acc_by_row=False):
        _lib.check(self.L.hpf_hip_segsum_f32(_ptr(part), _ptr(row_seg_ptr), _ptr(row_list), nrows, _ptr(acc), ld,
                                             ld if acc_ld is None else acc_ld, int(bool(acc_by_row)),
                                             self._stream()), "hpf_hip_segsum_f32")

    def pair_llk(self, T, B, ix_u, ix_i, y, k, ld, full_llk):
        """-> float64 tensor [3]: sum y*log(yhat) [- lgamma(y+1)], sum (y-yhat)^2, sum yhat."""
        n = int(ix_u.shape[0])
        grid = int(max(1, min(self.sweep_blocks, (n + 15) // 16)))
        partial = torch.empty((grid, 4), dtype=torch.float64, device=self.device)
        _lib.check(self.L.hpf_hip_pair_llk_f32(_ptr(T), _ptr(B), _ptr(ix_u), _ptr(ix_i), _ptr(y), n, _ptr(partial),
                                               k, ld, int(bool(full_llk)), grid, self._stream()),
                   "hpf_hip_pair_llk_f32")
        return partial.sum(dim=0)[:3]

    def llk_sweep(self, side, T, B, k, ld, full_llk):
        """pair_llk over the nonzeros of a row-grouped side (rows of T = side rows)."""
        grid = self.sweep_grid(side.nseg)
        partial = torch.empty((grid, 4), dtype=torch.float64, device=self.device)
        _lib.check(self.L.hpf_hip_llk_sweep_f32(_ptr(side.segs), side.nseg, _ptr(side.idx), _ptr(side.y), _ptr(T),
                                                _ptr(B), _ptr(partial), k, ld, int(bool(full_llk)), grid,
                                                self._stream()), "hpf_hip_llk_sweep_f32")
        return partial.sum(dim=0)[:3]

    def pair_dot(self, T, B, ix_u, ix_i, out, k, ld):
        _lib.check(self.L.hpf_hip_pair_dot_f32(_ptr(T), _ptr(B), _ptr(ix_u), _ptr(ix_i), int(ix_u.shape[0]),
                                               _ptr(out), k, ld, self._stream()), "hpf_hip_pair_dot_f32")

    def score_rows(self, vec, tab, out, k, ld):
        _lib.check(self.L.hpf_hip_score_rows_f32(_ptr(vec), _ptr(tab), int(tab.shape[0]), _ptr(out), k, ld,
                                                 self._stream()), "hpf_hip_score_rows_f32")

    def fold_in(self, idx, y, e_items, cs_other, shp, rte, fac, e_last, rounds, prior, top, add, rs, stop_thr, maxiter,
                k, ld):
        """The local coordinate ascent of one user (calc_user_factors, PXI:505-513) as one launch; shp / rte / fac are
        [ld] vectors updated in place, e_last the E row of the last round, rounds[0] the rounds executed."""
        _lib.check(self.L.hpf_hip_fold_in_f32(_ptr(idx), _ptr(y), int(y.shape[0]), _ptr(e_items), _ptr(cs_other),
                                              _ptr(shp), _ptr(rte), _ptr(fac), _ptr(e_last), _ptr(rounds), float(prior),
                                              float(top), float(add), float(rs), float(stop_thr), int(maxiter), k, ld,
                                              self._stream()), "hpf_hip_fold_in_f32")

    # -- index structures of a stochastic batch ------------------------------------------------
    def svi_batch_prepare(self, ws):
        """Everything a stochastic batch needs besides the dense algebra, built on the device by ONE call
        (hpf_hip_svi_batch_prepare) into the workspace `ws` (svi.BatchWorkspace): flags of the batch's rows (ws.ids; the
        workspace's previous batch, ws.prev_ids, is unmarked), the own side's compacted segment list, the other side's
        filtered copy, all sizes in ws.sizes -- nothing is read back."""
        import ctypes
        d = ws.__dict__.get("_hip_desc")
        if d is None:
            d = ws._hip_desc = SviBatchDesc()
            own, oth = ws.own, ws.oth
            d.own_segs, d.own_nseg, d.own_row_seg_ptr = _ptr(own.segs), own.nseg, _ptr(own.row_seg_ptr)
            d.own_indptr, d.own_nrows = _ptr(own.indptr), own.nrows
            d.oth_idx, d.oth_y, d.oth_indptr = _ptr(oth.idx), _ptr(oth.y), _ptr(oth.indptr)
            d.oth_nrows, d.oth_nnz = oth.nrows, oth.nnz
            assert ctypes.sizeof(d) == int(self.L.hpf_hip_svi_batch_sizeof())
            d.flag_own, d.flag_oth, d.acc_own = _ptr(ws.flag_own), _ptr(ws.flag_oth), _ptr(ws.acc_own)
            d.ld, d.seg_cap = int(ws.ld), int(ws.seg_cap)
            d.b_segs, d.b_segs_cap, d.b_multi, d.multi_cap = _ptr(ws.b_segs), ws.b_cap, _ptr(ws.b_multi), ws.multi_cap
            d.o_idx, d.o_y, d.o_cap = _ptr(ws.o_idx), _ptr(ws.o_y), ws.o_cap
            d.o_segs, d.o_segs_cap, d.o_multi = _ptr(ws.o_segs), ws.o_segs_cap, _ptr(ws.o_multi)
            d.sizes, d.mask, d.tile_cnt, d.tile_off = _ptr(ws.sizes), _ptr(ws.mask), _ptr(ws.tile_cnt), _ptr(ws.tile_off)
            d.chunk_pre, d.flag_bits = _ptr(ws.chunk_pre), _ptr(ws.flag_bits)
            d.row_start, d.row_cnt, d.tiles = _ptr(ws.row_start), _ptr(ws.row_cnt), _ptr(ws.tiles)
        d.ids, d.nids = _ptr(ws.ids), int(ws.ids.shape[0])
        d.prev_ids, d.nprev = (_ptr(ws.prev_ids), int(ws.prev_ids.shape[0])) if ws.prev_ids is not None else (None, 0)
        _lib.check(self.L.hpf_hip_svi_batch_prepare(ctypes.byref(d), self._stream()), "hpf_hip_svi_batch_prepare")

    def svi_prep_scratch_words(self):
        return int(self.L.hpf_hip_svi_prep_scratch_words())

    def svi_epoch_scratch_words(self, nb):
        return int(self.L.hpf_hip_svi_epoch_scratch_words(int(nb)))

    def svi_epoch_prepare(self, ws):
        """The index structures of EVERY batch of an epoch (svi.EpochWorkspace; ws.order = the epoch's shuffled rows), by
        one call (hpf_hip_svi_epoch_prepare): the other side's nonzeros partitioned by batch in one labelled pass instead
        of one filter pass per batch; nothing is read back."""
        import ctypes
        d = ws.__dict__.get("_hip_desc")
        if d is None:
            d = ws._hip_desc = SviEpochDesc()
            assert ctypes.sizeof(d) == int(self.L.hpf_hip_svi_epoch_sizeof())
            own, oth = ws.own, ws.oth
            d.own_segs, d.own_nseg, d.own_row_seg_ptr = _ptr(own.segs), own.nseg, _ptr(own.row_seg_ptr)
            d.own_indptr, d.own_nrows = _ptr(own.indptr), own.nrows
            d.oth_segs, d.oth_nseg, d.oth_row_seg_ptr = _ptr(oth.segs), oth.nseg, _ptr(oth.row_seg_ptr)
            d.oth_idx, d.oth_y, d.oth_nrows, d.oth_nnz = _ptr(oth.idx), _ptr(oth.y), oth.nrows, oth.nnz
            d.per, d.nb, d.ld, d.seg_cap = int(ws.per), int(ws.nb), int(ws.ld), int(ws.seg_cap)
            d.acc_own, d.batch_of, d.flag_own, d.flag_oth = _ptr(ws.acc_own), _ptr(ws.batch_of), _ptr(ws.flag_own), _ptr(ws.flag_oth)
            d.b_segs, d.b_segs_cap, d.b_multi, d.multi_cap = _ptr(ws.b_segs), ws.b_cap, _ptr(ws.b_multi), ws.multi_cap
            d.e_idx, d.e_y = _ptr(ws.e_idx), _ptr(ws.e_y)
            d.o_segs, d.o_segs_cap, d.o_multi = _ptr(ws.o_segs), ws.o_segs_cap, _ptr(ws.o_multi)
            d.sizes, d.key, d.seg_cnt, d.seg_pos, d.tiles = _ptr(ws.sizes), _ptr(ws.key), _ptr(ws.seg_cnt), _ptr(ws.seg_pos), _ptr(ws.tiles)
        assert int(ws.order.shape[0]) == ws.own.nrows and ws.order.dtype == torch.int64
        d.order = _ptr(ws.order)
        _lib.check(self.L.hpf_hip_svi_epoch_prepare(ctypes.byref(d), self._stream()), "hpf_hip_svi_epoch_prepare")

    def svi_coo_narrow(self, ids, limit, out, err):
        """out = ids as int32 row ids; err[0] = 1 when an id lies outside [0, limit) (hpf_hip_svi_coo_narrow)."""
        _lib.check(self.L.hpf_hip_svi_coo_narrow(_ptr(ids), int(ids.shape[0]), int(limit), _ptr(out), _ptr(err),
                                                 self._stream()), "hpf_hip_svi_coo_narrow")

    def svi_coo_prepare(self, key, nrows, seg_cap, flag, row_start, row_cnt, segs, multi, sizes, tiles):
        """One grouping of a COO batch from its stably sorted row ids `key` (hpf_hip_svi_coo_prepare): flags, segments,
        split-row descriptors, sizes -- nothing read back."""
        import ctypes
        d = SviCooDesc()
        assert ctypes.sizeof(d) == int(self.L.hpf_hip_svi_coo_sizeof())
        d.key, d.n, d.nrows, d.seg_cap = _ptr(key), int(key.shape[0]), int(nrows), int(seg_cap)
        d.flag, d.row_start, d.row_cnt = _ptr(flag), _ptr(row_start), _ptr(row_cnt)
        d.segs, d.segs_cap, d.multi, d.multi_cap = _ptr(segs), int(segs.shape[0]), _ptr(multi), int(multi.shape[0])
        d.sizes, d.tiles = _ptr(sizes), _ptr(tiles)
        _lib.check(self.L.hpf_hip_svi_coo_prepare(ctypes.byref(d), self._stream()), "hpf_hip_svi_coo_prepare")

    def segsum_desc(self, part, desc, ndesc_dev, ndesc_max, acc, ld):
        """acc[row] = sum of a split row's part[] rows for the {first, n, row} descriptors of a device-built batch."""
        _lib.check(self.L.hpf_hip_segsum_desc_f32(_ptr(part), _ptr(desc), _ptr(ndesc_dev), int(ndesc_max), _ptr(acc), ld,
                                                  self._stream()), "hpf_hip_segsum_desc_f32")

    def mt19937_words(self, state, raw):
        """raw[:] = the next raw.numel() state words of the MT19937 stream in `state` (int32[625] device tensor:
        numpy's key + pos, advanced in place); the sequential half of initialize_parameters' draws (PXI:127-138)."""
        n = int(raw.numel())
        words = int(self.L.hpf_hip_mt19937_scratch_words(n))     # > 0: a long draw, walked by many workgroups at once
        scratch = torch.empty(words, dtype=torch.int32, device=raw.device) if words > 0 else None
        _lib.check(self.L.hpf_hip_mt19937_words(_ptr(state), _ptr(raw), n, _ptr(scratch), self._stream()),
                   "hpf_hip_mt19937_words")
        if scratch is not None:
            scratch.record_stream(torch.cuda.current_stream(self.device))

    def uniform_rows(self, raw, out, nrows, k, ld, base, scale, den=None, ratio=None):
        """out[r, :k] = base + scale*U for the nrows*k stored words `raw` (numpy's float32 uniforms, bit for bit);
        ratio = out / den when given."""
        assert raw.numel() >= nrows * k
        _lib.check(self.L.hpf_hip_uniform_rows_f32(_ptr(raw), _ptr(out), _ptr(den), _ptr(ratio), int(nrows), float(base),
                                                   float(scale), k, ld, self._stream()), "hpf_hip_uniform_rows_f32")

    # -- stochastic-VI row kernels ------------------------------------------------------------
    def svi_shape_rows(self, row_list, acc, e, shp, prior, w_new, w_old, k, ld, acc_by_row=False):
        _lib.check(self.L.hpf_hip_svi_shape_rows_f32(_ptr(row_list), int(row_list.shape[0]), _ptr(acc), _ptr(e),
                                                     _ptr(shp), float(prior), float(w_new), float(w_old), k, ld,
                                                     int(bool(acc_by_row)), self._stream()), "hpf_hip_svi_shape_rows_f32")

    def refresh_grid(self, nrows):
        """Grid (= column-sum partial rows) of the whole-table SVI refresh: a pure streaming kernel."""
        return int(max(1, min(self.refresh_blocks, (nrows + 15) // 16)))

    def svi_refresh(self, nrows, shp, rte, fac, rs, cs_other, cs_partial, top, add, step, step_prev, refresh_rte,
                    blend_rs, k, ld):
        _lib.check(self.L.hpf_hip_svi_refresh_f32(nrows, _ptr(shp), _ptr(rte), _ptr(fac), _ptr(rs), _ptr(cs_other),
                                                  _ptr(cs_partial), float(top), float(add), float(step),
                                                  float(step_prev), int(bool(refresh_rte)), int(bool(blend_rs)), k,
                                                  ld, cs_partial.shape[0], self._stream()), "hpf_hip_svi_refresh_f32")

    def svi_side(self, nrows, flag, acc, e, shp, rte, fac, rs, cs_other, cs_partial, prior, w_new, w_old, top, add, step,
                 step_prev, rate_mode, rs_mode, k, ld, rs_rate=None, rs_prev_out=None, e_out=None, done_flag=0):
        """rte / fac None: not stored (rte: rate_mode 0 only); rs_prev_out: the scalar each row's rate was formed with;
        rs_rate: form the rate from these scalars instead of rs (expanding a factored rate); e_out: the flagged rows' new
        E rows (what `expect` would compute from the tables afterwards); done_flag: rows whose flag equals it were finished
        by sweep_svi and are skipped."""
        _lib.check(self.L.hpf_hip_svi_side_f32(nrows, _ptr(flag), _ptr(acc), _ptr(e), _ptr(shp), _ptr(rte), _ptr(fac),
                                               _ptr(rs), _ptr(cs_other), _ptr(cs_partial), float(prior), float(w_new),
                                               float(w_old), float(top), float(add), float(step), float(step_prev),
                                               int(rate_mode), int(rs_mode), k, ld, cs_partial.shape[0], _ptr(rs_rate),
                                               _ptr(rs_prev_out), _ptr(e_out), int(done_flag), self._stream()),
                   "hpf_hip_svi_side_f32")

    def sweep_svi(self, side, tab_self, tab_other, part, e_new, shp, rte, fac, rs, cs_other, cs_partial, prior, w_new,
                  w_old, top, add, step, step_prev, k, ld):
        """The sweep over a batch grouped by its OTHER side's rows with that side's step fused in (hpf_hip_sweep_svi_f32):
        rows present in one segment are finished by the wavefront that swept them; cs_partial: one row per block."""
        # (8 gathers in flight even for short rows: the epilogue caps the occupancy at 4 waves per SIMD either way, and
        #  the plain sweep's short-row launch buys its occupancy with half the gathers -- 25.7 -> 25.1 ms per C5 epoch)
        short = int(os.environ.get("HPF_SVI_FUSED_SHORT", "0"))
        _lib.check(self.L.hpf_hip_sweep_svi_f32(_ptr(side.segs), side.nseg, _ptr(side.idx), _ptr(side.y), _ptr(tab_self),
                                                _ptr(tab_other), _ptr(part), _ptr(e_new), _ptr(shp), _ptr(rte), _ptr(fac),
                                                _ptr(rs), _ptr(cs_other), _ptr(cs_partial), float(prior), float(w_new),
                                                float(w_old), float(top), float(add), float(step), float(step_prev), k, ld,
                                                short, cs_partial.shape[0], _ptr(getattr(side, "nseg_dev", None)),
                                                self._stream()), "hpf_hip_sweep_svi_f32")

    def sweep_svi_batch(self, side, e_self, tab_other, part, shp, rte_in, rte_out, fac, rs, rs_prev_out, factored, cs_other,
                        cs_partial, prior, w_new, w_old, top, add, step, step_prev, k, ld):
        """The sweep over a batch's OWN rows with both ends of a row's step fused in (hpf_hip_sweep_svi_batch_f32): the E
        row is formed in the prologue (factored = (rs, cs, top) of a rank-1 rate, or None: the table rte_in) and written to
        e_self; rows present in one segment are finished in the epilogue; cs_partial: one row per block."""
        r_rs, r_cs, r_top = factored if factored is not None else (None, None, 0.0)
        _lib.check(self.L.hpf_hip_sweep_svi_batch_f32(
            _ptr(side.segs), side.nseg, _ptr(side.idx), _ptr(side.y), _ptr(e_self), _ptr(tab_other), _ptr(part), _ptr(shp),
            _ptr(rte_in), _ptr(rte_out), _ptr(fac), _ptr(rs), _ptr(rs_prev_out), _ptr(r_rs), _ptr(r_cs), float(r_top),
            _ptr(cs_other), _ptr(cs_partial), float(prior), float(w_new), float(w_old), float(top), float(add), float(step),
            float(step_prev), k, ld, int(getattr(side, "short_rows", 0)), cs_partial.shape[0],
            _ptr(getattr(side, "nseg_dev", None)), self._stream()), "hpf_hip_sweep_svi_batch_f32")

    def svi_rate_rows(self, row_list, rte, fac, rs, cs_other, top, add, step, step_prev, mode, k, ld):
        _lib.check(self.L.hpf_hip_svi_rate_rows_f32(_ptr(row_list), int(row_list.shape[0]), _ptr(rte), _ptr(fac),
                                                    _ptr(rs), _ptr(cs_other), float(top), float(add), float(step),
                                                    float(step_prev), int(mode), k, ld, self._stream()),
                   "hpf_hip_svi_rate_rows_f32")
