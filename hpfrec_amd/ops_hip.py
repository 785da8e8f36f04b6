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
        _lib.check(self.L.hpf_hip_sweep_f32(_ptr(side.segs), side.nseg, _ptr(side.idx), _ptr(side.y),
                                            _ptr(tab_self), _ptr(tab_other), _ptr(part), _ptr(acc_rows), int(acc_ld),
                                            k, ld, int(getattr(side, "short_rows", 0)),
                                            grid_blocks or self.sweep_blocks, self._stream()),
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

    def sweep_prefinalize(self, side, tab_self, tab_other, part, acc_rows, acc_ld, shp, rte, fac, rs, cs_other,
                          cs_partial, prior_shp, top_shp, add_rte, k, ld, rs_prev=None):
        """sharded item pass: whole-row segments first finish their row from acc_rows (last iteration's reduced
        statistics), then sweep it and leave this iteration's local accumulator in acc_rows."""
        _lib.check(self.L.hpf_hip_sweep_prefinalize_f32(
            _ptr(side.segs), side.nseg, _ptr(side.idx), _ptr(side.y), _ptr(tab_self), _ptr(tab_other), _ptr(part),
            _ptr(acc_rows), int(acc_ld), _ptr(shp), _ptr(rte), _ptr(fac), _ptr(rs), _ptr(rs_prev), _ptr(cs_other),
            _ptr(cs_partial),
            float(prior_shp), float(top_shp), float(add_rte), k, ld, cs_partial.shape[0], self._stream()),
            "hpf_hip_sweep_prefinalize_f32")

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

    def unpack_rows(self, src, dst, nrows, k, ld):
        """dst[r, :k] = src[r, :k]: a packed [nrows, k] table into a padded [nrows, ld] one."""
        _lib.check(self.L.hpf_hip_unpack_rows_f32(_ptr(src), _ptr(dst), int(nrows), k, ld, self._stream()),
                   "hpf_hip_unpack_rows_f32")

    def colsum_reduce(self, cs_partial, cs_out, ld):
        _lib.check(self.L.hpf_hip_colsum_reduce_f32(_ptr(cs_partial), cs_partial.shape[0], _ptr(cs_out), ld,
                                                    self._stream()), "hpf_hip_colsum_reduce_f32")

    def colsum(self, tab, nrows, ld, cs_partial):
        _lib.check(self.L.hpf_hip_colsum_f32(_ptr(tab), nrows, ld, _ptr(cs_partial), cs_partial.shape[0],
                                             self._stream()), "hpf_hip_colsum_f32")

    def expect(self, shp, rte, e, nrows, k, ld, row_list=None):
        _lib.check(self.L.hpf_hip_expect_f32(_ptr(shp), _ptr(rte), _ptr(e), _ptr(row_list), nrows, k, ld,
                                             self._stream()), "hpf_hip_expect_f32")

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

    # -- index plumbing of a stochastic batch ---------------------------------------------------
    def gather_rows(self, src_begin, dst_begin, row_ids, src_idx, src_y, out_idx, out_y, out_row):
        """Nonzeros of the listed rows (row t: src[src_begin[t] ...) -> out[dst_begin[t] .. dst_begin[t+1]))."""
        _lib.check(self.L.hpf_hip_gather_rows(_ptr(src_begin), _ptr(dst_begin), _ptr(row_ids), int(row_ids.shape[0]),
                                              _ptr(src_idx), _ptr(src_y), _ptr(out_idx), _ptr(out_y), _ptr(out_row),
                                              self._stream()), "hpf_hip_gather_rows")

    def fill_segments(self, start, count, row_seg_ptr, row_ids, seg_cap, segs):
        """hpf_segment descriptors (int64 [nseg,2] image) of rows (start, count, id) cut at seg_cap nonzeros."""
        _lib.check(self.L.hpf_hip_fill_segments(_ptr(start), _ptr(count), _ptr(row_seg_ptr), _ptr(row_ids),
                                                int(row_ids.shape[0]), int(seg_cap), _ptr(segs), self._stream()),
                   "hpf_hip_fill_segments")

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
                 step_prev, rate_mode, rs_mode, k, ld):
        _lib.check(self.L.hpf_hip_svi_side_f32(nrows, _ptr(flag), _ptr(acc), _ptr(e), _ptr(shp), _ptr(rte), _ptr(fac),
                                               _ptr(rs), _ptr(cs_other), _ptr(cs_partial), float(prior), float(w_new),
                                               float(w_old), float(top), float(add), float(step), float(step_prev),
                                               int(rate_mode), int(rs_mode), k, ld, cs_partial.shape[0], self._stream()),
                   "hpf_hip_svi_side_f32")

    def svi_rate_rows(self, row_list, rte, fac, rs, cs_other, top, add, step, step_prev, mode, k, ld):
        _lib.check(self.L.hpf_hip_svi_rate_rows_f32(_ptr(row_list), int(row_list.shape[0]), _ptr(rte), _ptr(fac),
                                                    _ptr(rs), _ptr(cs_other), float(top), float(add), float(step),
                                                    float(step_prev), int(mode), k, ld, self._stream()),
                   "hpf_hip_svi_rate_rows_f32")
