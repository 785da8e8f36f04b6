/*
 * hpf_hip.h -- C ABI of libhpf_hip.so, the MI355X (gfx950) replacement for the
 * Cython/OpenMP loops of david-cortes/hpfrec (hpfrec/cython_loops.pxi, "PXI").
 *
 * The reference has no FFI of its own: hpfrec/__init__.py calls module-level Python
 * functions of the compiled extension hpfrec.cython_loops_float, which in turn call
 * `cdef ... nogil` loops.  Those cdef loops are the functions replaced here; each
 * entry point below cites the loop(s) it stands in for.  The Python-level drop-in
 * (same names/arity as the extension module) lives in hpfrec_amd/cython_loops_float.py
 * and binds this ABI through ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (e.g. torch.Tensor.data_ptr()) unless stated otherwise; no device
 *     memory is allocated, retained or freed by the library (a shard plan, last section, keeps host-side state:
 *     a descriptor and a few events); all launches are asynchronous on
 *     `stream` (a hipStream_t passed as void*; NULL = the default stream);
 *   - return value: 0 on success, otherwise the hipError_t of the failed launch or a
 *     negative HPF_E* code for argument errors; no exception crosses the ABI;
 *   - factor tables are row-major float32 [nrows][ld] with ld = hpf_hip_ld_for_k(k)
 *     (>= k, power of two >= 32, rows 128-byte aligned); columns k..ld-1 of the
 *     e_* / fac tables are zero and must stay zero;
 *   - index arrays are int32 (row ids of the *other* side), segment descriptors are
 *     hpf_segment.  The reference uses size_t indices (cython_float_nonwindows.pyx:8);
 *     the Python layer narrows them after a range check.
 */
#ifndef HPF_HIP_H
#define HPF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HPF_HIP_ABI_VERSION 24

#define HPF_EINVAL (-1)  /* bad argument (null pointer, k<=0, ld mismatch ...) */
#define HPF_EUNSUPPORTED (-2) /* k larger than the kernels are instantiated for */
#define HPF_ENOLIB (-3)       /* RCCL entry points not bound (hpf_hip_rccl_open failed or was not called) */
#define HPF_ETIMEOUT (-4)     /* direct exchange: a peer's flag did not arrive within the region's time-out */
#define HPF_ERCCL_BASE (-1000) /* an RCCL call failed: the return value is HPF_ERCCL_BASE - ncclResult_t */

/* A contiguous run of nonzeros belonging to one sparse row (CSR row of a user, or CSC
 * column of an item).  Rows longer than the segment cap are split into several
 * segments so that no wavefront owns an unbounded amount of work. */
typedef struct hpf_segment {
    int64_t begin; /* offset of the first nonzero in idx[] / y[]                          */
    int32_t len;   /* number of nonzeros (> 0) in the low 24 bits, HPF_SEG_* flags above  */
    int32_t row;   /* row of tab_self this segment belongs to                             */
} hpf_segment;

#define HPF_SEG_LEN_MASK 0x00FFFFFF
#define HPF_SEG_WHOLE_ROW 0x40000000 /* the segment is the only one of its row */

int hpf_hip_abi_version(void);

/* Padded leading dimension for k latent factors, or HPF_EUNSUPPORTED. */
int hpf_hip_ld_for_k(int k);

/* Device name ("gfx950...") and compute-unit count of the current device. */
int hpf_hip_device_info(int *cu_count, char *arch, int arch_len);

/*
 * Fused phi + shape accumulation for one side of the bipartite graph.
 * Replaces update_phi (PXI:551-591) + update_G_n_L_sh (PXI:613-621) for the rows of
 * tab_self, without materialising phi (reference: np.empty((nY,k)), PXI:187):
 *
 *   for every segment g, nonzero n in g (other-side row c = idx[n], count y[n]):
 *       s = <tab_self[row(g)], tab_other[c]>;   w = y[n] / s
 *       part[g][:] += w * tab_other[c][:]
 *
 * with tab_self/tab_other holding E = exp(psi(shape) - log(rate)) rows (row-scaled,
 * see hpf_hip_row_finalize_f32), so that shape_row = prior + tab_self[row] (*) sum of
 * that row's part[] entries.  One wavefront per segment, 64/(ld/4) nonzeros per step.  Call once per side (the
 * deterministic two-pass scheme; a one-pass form with fp32 atomics for the other side was measured at 4x the time
 * of the second pass and removed, profiles/r02_atomic_one_pass.txt).
 * acc_rows (optional): segments flagged HPF_SEG_WHOLE_ROW write their accumulator to
 * acc_rows[row][0:acc_ld] (k <= acc_ld <= ld, packed) instead of part[g] -- the multi-GPU exchange buffer.
 * short_rows: tuning hint (0/1) for rows that average a few dozen nonzeros at most: the same kernel with half the
 * gathers in flight per wavefront (smaller register file, more wavefronts resident).  Results are identical.
 * nseg_dev (optional, device): the live number of segments, <= nseg -- a stochastic batch whose size only the device
 * knows (hpf_hip_svi_batch_prepare); nseg is then the capacity of segs[] / part[].
 */
int hpf_hip_sweep_f32(const hpf_segment *segs, int64_t nseg, const int32_t *idx, const float *y,
                      const float *tab_self, const float *tab_other, float *part, float *acc_rows, int acc_ld, int k,
                      int ld, int short_rows, int grid_blocks, const int64_t *nseg_dev, void *stream);

/*
 * hpf_hip_sweep_f32 with the row finalizer (next entry) fused in: a segment flagged
 * HPF_SEG_WHOLE_ROW is finished by the wavefront that swept it -- shp/rte/fac/rs/e_new of its row are
 * written directly (e_old = tab_self; e_new may alias tab_self) and nothing goes to part[]; other
 * segments write part[] as usual and their rows (and rows without any nonzero) are finished by a
 * following hpf_hip_row_finalize_f32 call with a row_list.  cs_partial must have grid_blocks rows;
 * all are written.  This overlaps the fp64 transcendental work with the gathers of other waves.
 */
int hpf_hip_sweep_finalize_f32(const hpf_segment *segs, int64_t nseg, const int32_t *idx, const float *y,
                               const float *tab_self, const float *tab_other, float *part, float *e_new, float *shp,
                               float *rte, float *fac, float *rs, float *rs_prev, const float *cs_other,
                               float *cs_partial, float prior_shp, float top_shp, float add_rte, int k, int ld,
                               int grid_blocks, void *stream);

/*
 * Closed-form updates for the rows of one side.  Replaces the numpy statements of
 * fit_hpf PXI:236-259 (and the psi/log/exp hoisted out of update_phi, PXI:588):
 *
 *   r        = row_list ? row_list[t] : t,   t = 0 .. nrows-1   (row_list: finish only the listed rows)
 *   acc      = sum of part[g] over the row's segments g in [row_seg_ptr[r], row_seg_ptr[r+1])
 *              (row_seg_ptr == NULL: part is already one accumulator row per table row)
 *   shp[r]   = prior_shp + e_old[r] (*) acc              Gamma_shp / Lambda_shp
 *   rte[r]   = top_shp / rs[r] + cs_other                 Gamma_rte (PXI:236) / Lambda_rte (PXI:255)
 *   fac[r]   = shp[r] / rte[r]                            Theta (PXI:251) / Beta (PXI:256)
 *   rs[r]    = add_rte + sum_k fac[r]                     k_rte (PXI:258) / t_rte (PXI:259)
 *   e_new[r] = exp(psi(shp[r]) - log(rte[r])) / max_k(.)  input of the next sweep
 *   cs_partial[block] = per-block column sums of fac      -> hpf_hip_colsum_reduce_f32
 *
 * e_new may alias e_old.  rs is updated in place.  shp/rte/fac may be NULL (skip store).
 * rs_prev (optional) receives the row's OLD rs: rte = top_shp/rs_prev + cs_other is rank-1, so a caller can keep
 * rs_prev (nrows floats) + cs_other (k floats) instead of the [nrows][ld] rte table and expand it on output.
 * part_ld is the row stride of part[] (ld, or k for the packed all-reduce payload of the multi-GPU path).
 */
int hpf_hip_row_finalize_f32(const float *part, const int64_t *row_seg_ptr, const int64_t *row_list, int64_t nrows,
                             const float *e_old, float *e_new, float *shp, float *rte, float *fac, float *rs,
                             float *rs_prev, const float *cs_other, float *cs_partial, float prior_shp, float top_shp,
                             float add_rte, int k, int ld, int part_ld, int grid_blocks, void *stream);

/*
 * hpf_hip_row_finalize_f32 for accumulator rows that are dense and already reduced (row_seg_ptr == row_list == NULL),
 * over SEVERAL row ranges in one launch: range i covers range_rows[i] rows, accumulator rows
 * acc[range_acc_begin[i] + j][0:acc_ld] <-> table rows range_row_begin[i] + j of e_old/shp/rte/fac/rs/rs_prev; the
 * new E row is written to e_new[range_acc_begin[i] + j] (e_new is indexed like acc: it is the all-gather send
 * buffer).  The multi-GPU "scatter" exchange leaves each rank with one slice of every item range (the
 * reduce-scatter outputs, concatenated in acc); this finishes all of them at once.  The three range arrays are
 * HOST arrays of nranges <= HPF_MAX_ROW_RANGES entries, read during the call.  cs_partial: grid_blocks rows, all
 * written.  e_new_ld: row stride of e_new -- ld, or k <= e_new_ld < ld for a packed send buffer (only the first
 * e_new_ld columns are written).
 */
#define HPF_MAX_ROW_RANGES 8
int hpf_hip_row_finalize_ranges_f32(const float *acc, int nranges, const int64_t *range_rows,
                                    const int64_t *range_acc_begin, const int64_t *range_row_begin,
                                    const float *e_old, float *e_new, float *shp, float *rte, float *fac, float *rs,
                                    float *rs_prev, const float *cs_other, float *cs_partial, float prior_shp,
                                    float top_shp, float add_rte, int k, int ld, int acc_ld, int e_new_ld,
                                    int grid_blocks, void *stream);

/*
 * The item finalizer of the sharded path in two parts (the "gather-early" exchange: the all-gather of the new item
 * expectations runs UNDER the user sweep; hpf_hip.hip item_shape_kernel / item_apply_kernel).  Same statements as
 * hpf_hip_row_finalize_ranges_f32 (PXI:239-259 for the items) in a different order of evaluation:
 * part 1, hpf_hip_item_shape_rows_f32 (owner of a slice, needs only the reduce-scattered statistics): for the rows of the
 *   ranges (as in hpf_hip_row_finalize_ranges_f32; acc is packed [.][k]): shp = prior + e_old (*) acc -> shp_out[t] (padded
 *   [.][ld], pads zero); send[t][0:k] = exp(psi(shp)) row-scaled, send[t][k] = top_shp / rs[r], zero up to the row stride
 *   hpf_hip_gather_payload_ld(k) = k+1 rounded up to a multiple of 4 (the all-gather payload, read as float4 by part 2);
 *   rs_prev[r] = rs[r].
 * part 2, hpf_hip_item_apply_rows_f32 (every rank, all items, after colsum(Theta) is known): recv = the all-gathered
 *   payload, [world][rows per rank][payload ld] with a rank's slices in range (issue) order; for every table row r < nrows:
 *   e_tab[r] = recv.num / (recv.base + cs_other), row-scaled to [1,2); rows this rank owns also get fac = shp/rte
 *   (shp_own: the padded shapes of part 1), rs[r] = add_rte + sum_k fac, optional shp / fac stores, and the
 *   per-block column sums of fac in cs_partial (grid_blocks rows, all written; grid_blocks must be a multiple of
 *   world: the grid is (grid_blocks / world) x world, one column of blocks per rank's block of the gathered buffer).
 *   The E rows use reciprocal-and-multiply (a few ulps; identical on every rank), the means the correctly rounded
 *   division of the one-part finalizer.  range_lo / range_hi: HOST arrays.
 */
int hpf_hip_gather_payload_ld(int k);
int hpf_hip_item_shape_rows_f32(const float *acc, int nranges, const int64_t *range_rows, const int64_t *range_acc_begin,
                                const int64_t *range_row_begin, const float *e_old, float *shp_out, float *send,
                                const float *rs, float *rs_prev, float prior_shp, float top_shp, int k, int ld,
                                int grid_blocks, void *stream);
int hpf_hip_item_apply_rows_f32(const float *recv, const float *shp_own, float *e_tab, float *shp, float *fac, float *rs,
                                const float *cs_other, float *cs_partial, float add_rte, int k, int ld, int rank,
                                int world, int64_t nrows, int nranges, const int64_t *range_lo, const int64_t *range_hi,
                                int grid_blocks, void *stream);


/* cs_out[c] = sum_b cs_partial[b][c], fixed order, double accumulation (Beta.sum(axis=0), PXI:236,255). */
int hpf_hip_colsum_reduce_f32(const float *cs_partial, int nblk, float *cs_out, int ld, void *stream);

/* Per-block column sums of a table (first iteration / partial_fit: Beta.sum(axis=0) from host-initialised Beta). */
int hpf_hip_colsum_f32(const float *tab, int64_t nrows, int ld, float *cs_partial, int grid_blocks, void *stream);

/* cs_out[c] = tab[0][c] + tab[1][c] + ... in float32, the rows IN SEQUENCE: numpy's own order for Beta.sum(axis=0) /
 * Theta.sum(axis=0) (PXI:236,255; the stochastic steps' PXI:300,320 / 352,372), reproduced bit for bit -- one lane per
 * column, a chain of nrows dependent adds, fed through LDS by the other waves of the workgroup (16 columns per workgroup):
 * 4.25 ns per row (a dependent v_add_f32 issues every 8 cycles: 3.3 ns is the floor).  HPF_COLSUM_ORDER=reference makes the
 * drivers use it instead of the partial sums of the sweeps + hpf_hip_colsum_reduce_f32: the mode for callers who need the
 * reference's sums -- it is the summation order of these statements that separates the default path from the reference at
 * 1e5..1e6 rows (tests/test_hip_parity.py::test_large_vs_golden, tests/test_svi_paths.py::test_svi_large_vs_golden_with_
 * reference_order_sums_on_gpu).  tab must be 16-byte aligned. */
int hpf_hip_colsum_sequential_f32(const float *tab, int64_t nrows, int ld, float *cs_out, void *stream);

/* e[r] = exp(psi(shp[r]) - log(rte[r])) / rowmax, pads zeroed: the hoisted transcendental part of
 * update_phi (PXI:570,588,685) for rows whose shape/rate did not come out of hpf_hip_row_finalize_f32
 * (initialisation PXI:134-138; the rows of an SVI batch).  r = row_list ? row_list[t] : t, t < nrows; with `flag`
 * (optional, one byte per table row) only rows with flag[r] != 0 are touched.  rate_rs (optional): the rate is not read
 * from a table but formed as rate_top / rate_rs[r] + rate_cs[c] -- the factored (rank-1) form a stochastic epoch keeps for
 * its batch side (hpf_hip_svi_side_f32: rs_prev_out); rte may then be NULL.  rte_out (optional, with rate_rs): the rate
 * rows formed that way are also STORED there -- the expansion of a factored rate into its table (what the other side's
 * blend of a stochastic step needs, PXI:320 / 372) rides on the pass that refreshes the side's E table at an epoch
 * boundary instead of being a second pass over the shapes. */
int hpf_hip_expect_f32(const float *shp, const float *rte, float *e, const int64_t *row_list, const uint8_t *flag,
                       int64_t nrows, int k, int ld, const float *rate_rs, const float *rate_cs, float rate_top,
                       float *rte_out, void *stream);

/* acc[t][0:acc_ld] (or acc[r][0:acc_ld] when acc_by_row) = sum of the part[] segments of row
 * r = row_list ? row_list[t] : t, t < nrows (multi-GPU item side before the all-reduce -- acc_ld = k packs the
 * payload; the batch rows of an SVI step). */
int hpf_hip_segsum_f32(const float *part, const int64_t *row_seg_ptr, const int64_t *row_list, int64_t nrows,
                       float *acc, int ld, int acc_ld, int acc_by_row, void *stream);

/*
 * Poisson log-likelihood terms over listed pairs.  Replaces llk_plus_rmse (PXI:627-658)
 * and sum_prediction (PXI:816-825): per block b, partial[4b+0] = sum y*log(yhat)
 * [- lgamma(y+1) when full_llk], partial[4b+1] = sum (y-yhat)^2, partial[4b+2] = sum yhat,
 * accumulated in double (reference: long double).  The caller adds the blocks.
 */
int hpf_hip_pair_llk_f32(const float *T, const float *B, const int32_t *ix_u, const int32_t *ix_i, const float *y,
                         int64_t n, double *partial, int k, int ld, int full_llk, int grid_blocks, void *stream);

/* The same three sums over the training nonzeros in the row-grouped layout of hpf_hip_sweep_f32 (rows of T
 * read once per segment, only B rows gathered): the train-llk evaluation of assess_convergence (PXI:75-79). */
int hpf_hip_llk_sweep_f32(const hpf_segment *segs, int64_t nseg, const int32_t *idx, const float *y, const float *T,
                          const float *B, double *partial, int k, int ld, int full_llk, int grid_blocks,
                          void *stream);

/* out[n] = <T[ix_u[n]], B[ix_i[n]]>.  Replaces predict_multiple (PXI:803-810). */
int hpf_hip_pair_dot_f32(const float *T, const float *B, const int32_t *ix_u, const int32_t *ix_i, int64_t n,
                         float *out, int k, int ld, void *stream);

/*
 * Stochastic VI (fit_hpf SVI epochs PXI:262-377, partial_fit PXI:423-473): the dense statements around the
 * batch's phi, as three row kernels.  `acc` holds sum_n (w_n * other-side E row) per listed row, i.e. the
 * output of hpf_hip_sweep_f32 + hpf_hip_segsum_f32 over the batch's rows, aligned with row_list.
 *
 * shape rows:  shp[r] = w_new*(prior + e[r] (*) acc[t]) + w_old*shp[r],  r = row_list[t]
 *              (batch side: w_new=1, w_old=0 -- PXI:304-314; other side: step*multiplier, 1-step -- PXI:316,368)
 * refresh:     every row of a side: [rte = top/rs + cs_other] ; fac = shp/rte ;
 *              [rs = step*(add + sum_k fac) + step_prev*rs] ; cs_partial = per-block column sums of fac
 *              (PXI:300,318,322 / 352,370,374; the rs blend over all rows is partial_fit's PXI:472-473)
 *              acc row of list entry t: acc[t] (acc_by_row = 0) or acc[row_list[t]] (acc_by_row = 1: a full-height
 *              accumulator table the batch sweeps wrote straight into)
 * rate rows:   mode 0: rte[r] = step*(top/rs[r] + cs_other) + step_prev*rte[r]        (PXI:320,372)
 *              mode 1: rs[r]  = step*(add + sum_k fac[r])   + step_prev*rs[r]         (PXI:324-325,376-377)
 */
int hpf_hip_svi_shape_rows_f32(const int64_t *row_list, int64_t nrows, const float *acc, const float *e, float *shp,
                               float prior, float w_new, float w_old, int k, int ld, int acc_by_row, void *stream);
int hpf_hip_svi_refresh_f32(int64_t nrows, const float *shp, float *rte, float *fac, float *rs, const float *cs_other,
                            float *cs_partial, float top, float add, float step, float step_prev, int refresh_rte,
                            int blend_rs, int k, int ld, int grid_blocks, void *stream);
/*
 * The three statements groups above for ONE side in ONE pass over all of its rows (what the drivers use): flag[r] != 0
 * marks the rows of the step (the batch's rows / the rows the batch touched; flag == NULL: none),
 *   flagged:        shp[r] = w_new*(prior + e[r] (*) acc[r]) + w_old*shp[r]          (acc, e: full-height tables)
 *   rate_mode 0:    rte[r] = top/rs[r] + cs_other for every row   (the batch side, PXI:300 / 352)
 *   rate_mode 1:    rte[r] = step*(top/rs[r] + cs_other) + step_prev*rte[r] for flagged rows   (PXI:320 / 372)
 *   every row:      fac[r] = shp[r]/rte[r];  cs_partial = per-block column sums of fac
 *   rs_mode 0/1/2:  rs[r] = step*(add + sum_k fac[r]) + step_prev*rs[r] for no / flagged / all rows
 * Row-local, the same float32 operations in the same order as the separate kernels.
 * fac may be NULL (means not stored: only their column sums and the scalar rates are needed between checks); with
 * rate_mode 0 rte may be NULL too and rs_prev_out[r] (optional, every row) receives the scalar the rate was formed with,
 * so that rte = top/rs_prev_out + cs_other can be expanded later: rs_rate (optional) forms the rate from THAT scalar
 * instead of rs[r] -- the expansion is this same call with flag = NULL, rs_mode 0, rs_rate = the kept scalars.
 * e_out (optional; may be e): the flagged rows also get their NEW expectation row exp(psi(shp[r]))/rte[r], row-scaled,
 * from the shape and rate just formed -- bit for bit what hpf_hip_expect_f32 would compute from the tables afterwards
 * (the psi/log/exp hoisted out of update_phi, PXI:588), without reading them back.  In a rate_mode 1 pass nothing else of
 * the side changes, so an E table that was current for all rows stays current and the next step needs no expectation pass
 * over this side.
 */
int hpf_hip_svi_side_f32(int64_t nrows, const uint8_t *flag, const float *acc, const float *e, float *shp, float *rte,
                         float *fac, float *rs, const float *cs_other, float *cs_partial, float prior, float w_new,
                         float w_old, float top, float add, float step, float step_prev, int rate_mode, int rs_mode,
                         int k, int ld, int grid_blocks, const float *rs_rate, float *rs_prev_out, float *e_out,
                         int done_flag, void *stream);
/*
 * done_flag (0: none): rows whose flag EQUALS done_flag were finished elsewhere and are skipped entirely -- no load, no
 * store, no share of the column sums.  That elsewhere is the next entry: the OTHER side's pass of a stochastic step fused
 * into the sweep that forms its phi-sums.  hpf_hip_sweep_svi_f32 = hpf_hip_sweep_f32 over the batch grouped by the other
 * side's rows + for every HPF_SEG_WHOLE_ROW segment the flagged-row statements of hpf_hip_svi_side_f32 with rate_mode 1,
 * rs_mode 1, applied by the wavefront that swept the row while it still holds the row's phi-sum and E row: shp and rte
 * are read and rewritten, fac (optional) written, rs blended, e_new (optional; must be tab_self) = the row's new E row,
 * cs_partial[grid_blocks][ld] = per-block column sums of fac over the rows finished here (all blocks write theirs).
 * Split rows write part[] as in hpf_hip_sweep_f32 and are finished, with the rows the batch does not touch, by a
 * hpf_hip_svi_side_f32 call with done_flag = the flag value of whole rows (the preparations above write 1 for a row
 * present in one segment, 2 for a split row); the two calls' partial column sums add up to the side's (PXI:370-374).
 * Same float32 statements through the same device functions: a row finished here and one finished there get the same
 * shapes, rates, means and E row bit for bit (tests/test_hip_parity.py::test_sweep_svi_op_row_for_row); its scalar rate holds
 * a k-term sum that the two kernels fold in different orders (1e-7).
 */
int hpf_hip_sweep_svi_f32(const hpf_segment *segs, int64_t nseg, const int32_t *idx, const float *y, const float *tab_self,
                          const float *tab_other, float *part, float *e_new, float *shp, float *rte, float *fac, float *rs,
                          const float *cs_other, float *cs_partial, float prior, float w_new, float w_old, float top,
                          float add, float step, float step_prev, int k, int ld, int short_rows, int grid_blocks,
                          const int64_t *nseg_dev, void *stream);
int hpf_hip_svi_rate_rows_f32(const int64_t *row_list, int64_t nrows, float *rte, const float *fac, float *rs,
                              const float *cs_other, float top, float add, float step, float step_prev, int mode,
                              int k, int ld, void *stream);

/*
 * The BATCH side of a stochastic step in the sweep that forms its phi-sums (PXI:292-314,324 for a user batch, 344-366,377 for
 * an item batch; partial_fit PXI:438-457,472-473).  hpf_hip_sweep_svi_batch_f32 = for every segment of the batch's rows
 *   prologue:  the row's E row exp(psi(shp[row]))/rate[row], row-scaled (hpf_hip_expect_f32's statements: the psi/log/exp
 *              of update_phi_csr, PXI:683-692, hoisted to the row), from the row's CURRENT shape and rate -- the rate either
 *              factored, rate_top / rate_rs[row] + rate_cs[c] (rate_rs != NULL), or the stored table rte_in -- written to
 *              e_self[row] (the other side's sweep of the same step gathers it) and kept in registers;
 *   sweep:     hpf_hip_sweep_f32's loop over the segment's nonzeros (gathers from tab_other);
 *   epilogue (HPF_SEG_WHOLE_ROW segments): the flagged-row statements of hpf_hip_svi_side_f32 with rate_mode 0 --
 *              shp = w_new*(prior + E (*) phi-sum), rate = top/rs[row] + cs_other (stored to rte_out / the mean to fac
 *              when those are given), rs[row] = step*(add + sum_k mean) + step_prev*rs[row], rs_prev_out[row] (optional) =
 *              the scalar the rate was formed with, cs_partial[grid_blocks][ld] = per-block column sums of the means of
 *              the rows finished here (all blocks write theirs).
 * Split rows write part[] as in hpf_hip_sweep_f32 and are finished, with every row outside the batch, by a
 * hpf_hip_svi_side_f32 call with done_flag = the flag value of whole rows (the batch preparations write 1 for a batch row
 * present in one segment, 2 for a split row or a row without nonzeros); the two calls' partial column sums add up to the
 * side's (PXI:318 / 370).  Same float32 statements through the same device functions as the separate passes.
 * rate_rs may alias rs_prev_out and rte_in may alias rte_out (a row is read in its prologue, written in its epilogue).
 */
int hpf_hip_sweep_svi_batch_f32(const hpf_segment *segs, int64_t nseg, const int32_t *idx, const float *y, float *e_self,
                                const float *tab_other, float *part, float *shp, const float *rte_in, float *rte_out,
                                float *fac, float *rs, float *rs_prev_out, const float *rate_rs, const float *rate_cs,
                                float rate_top, const float *cs_other, float *cs_partial, float prior, float w_new,
                                float w_old, float top, float add, float step, float step_prev, int k, int ld,
                                int short_rows, int grid_blocks, const int64_t *nseg_dev, void *stream);

/*
 * calc_user_factors (PXI:476-520): the local coordinate ascent of ONE user against fixed item parameters, looping on
 * the device.  The user's n items are rows idx[t] of the items' E table `e_items` ([rows][ld], hpf_hip_expect_f32 of
 * Lambda_shp / Lambda_rte), with counts y[t]; cs_other = Beta.sum(axis=0).  shp / rte / fac ([ld] each) hold the
 * user's initial Gamma_shp, Gamma_rte and Theta (PXI:490-497) and return the final ones; rs is the initial k_rte.
 * Per round (PXI:505-513): E row from (shp, rte); acc = sum_t y_t * e_items[idx_t] / <E, e_items[idx_t]>;
 * rte = top/rs + cs_other; shp = prior + E (*) acc; fac = shp/rte; rs = add + sum_k fac; stop when
 * ||fac - fac_prev||_2 < stop_thr, after at most maxiter rounds.  e_last ([ld]) = the E row of the LAST round (the
 * reference returns phi/Y of that round), rounds[0] = rounds executed.  One workgroup.
 */
int hpf_hip_fold_in_f32(const int32_t *idx, const float *y, int64_t n, const float *e_items, const float *cs_other,
                        float *shp, float *rte, float *fac, float *e_last, int32_t *rounds, float prior, float top,
                        float add, float rs, float stop_thr, int maxiter, int k, int ld, void *stream);

/*
 * Index structures of a stochastic batch, built on the device by one call (hpfrec_amd/csrc/hpf_svi_prep.hip).  The
 * reference slices its CSR / CSC per batch with numpy fancy indexing on the host (PXI:280-290, 332-342) and collects the
 * other side's rows with get_unique_items_batch (PXI:27-42).  Here, for the batch made of rows `ids` of the "own" side
 * (users for a user batch, items for an item batch; both sides in the layout of hpf_hip_sweep_f32):
 *   own side:   b_segs = the stable compaction of the side's global segment list by the batch's rows -- descriptors
 *               unchanged, they index the side's global idx / y; b_multi[m] = {first compacted segment, segments, row}
 *               of each split row; flag_own[row] = 1 for the batch's rows (the previous batch of this workspace,
 *               prev_ids, is unmarked first); acc_own[row] is zeroed for batch rows without nonzeros;
 *   other side: the batch's nonzeros grouped by the other side's rows, in the order the other side's global layout holds
 *               them (ascending own-side ids: stable, no sort): o_idx = own-side row id, o_y = count, o_segs / o_multi as
 *               above with `begin` indexing o_idx / o_y; flag_oth[row] = 1 for rows present in one segment, 2 for split
 *               rows, 0 for all others.
 * sizes[0..8) (device int64): segments own, split rows own, segments other, split rows other, nonzeros, rows other, -,
 * overflow (a capacity was too small; never with capacities from the side's largest rows; STICKY: entries 0..6 are reset
 * by every call, the overflow flag only by the caller, so one read after many batches sees any of them).  Nothing is read back: the
 * consumers take their counts from `sizes` (hpf_hip_sweep_f32 nseg_dev, hpf_hip_segsum_desc_f32, hpf_hip_expect_f32 flag).
 * How: the own side is a stable compaction of its segment list; the other side is the other side's flat nonzero array
 * (already grouped by its rows, own-side ids ascending inside a row) filtered by flag_own -- a keep-bitmask pass, prefix
 * sums per 1024-entry tile, a scan over the rows, a write pass; bandwidth-bound, no sort, no atomics on floats.
 * All pointers are device memory.
 */
typedef struct hpf_svi_batch {
    const hpf_segment *own_segs; int64_t own_nseg; const int64_t *own_row_seg_ptr; const int64_t *own_indptr;
    int64_t own_nrows;
    const int32_t *oth_idx; const float *oth_y; const int64_t *oth_indptr; int64_t oth_nrows; int64_t oth_nnz;
    const int64_t *ids; int64_t nids; const int64_t *prev_ids; int64_t nprev;
    uint8_t *flag_own; uint8_t *flag_oth; float *acc_own; int32_t ld, seg_cap;
    hpf_segment *b_segs; int64_t b_segs_cap; int64_t *b_multi; int64_t multi_cap;
    int32_t *o_idx; float *o_y; int64_t o_cap; hpf_segment *o_segs; int64_t o_segs_cap; int64_t *o_multi;
    int64_t *sizes;
    /* scratch: mask [16 * ceil(oth_nnz/1024)] 8-byte words (a keep bit per nonzero); chunk_pre [same count] uint16;
     * tile_cnt [ceil(oth_nnz/1024)+1] int32; tile_off [ceil(oth_nnz/1024)+1] int64; row_start [oth_nrows] int64;
     * row_cnt [oth_nrows] int32; tiles [hpf_hip_svi_prep_scratch_words()] int64 */
    uint64_t *mask; uint16_t *chunk_pre; int32_t *tile_cnt; int64_t *tile_off; int64_t *row_start; int32_t *row_cnt;
    int64_t *tiles; uint32_t *flag_bits;   /* flag_bits [ceil(own_nrows/32)]: flag_own as a bitset (it fits a CU's LDS) */
} hpf_svi_batch;
int64_t hpf_hip_svi_batch_sizeof(void);   /* sizeof(hpf_svi_batch), for a binding's layout check */
int64_t hpf_hip_svi_prep_scratch_words(void);
int hpf_hip_svi_batch_prepare(const hpf_svi_batch *batch, void *stream);
/*
 * The same structures for ALL batches of one epoch, built once (hpfrec_amd/csrc/hpf_svi_prep.hip, second half).  An
 * epoch's batches partition the rows of its side (`order` is the epoch's shuffle, PXI:277 / 329; batch b = rows
 * order[b*per .. min(own_nrows, (b+1)*per))), so every nonzero belongs to exactly one batch: instead of filtering the
 * other side's 48M ids once per batch (11-16 passes per C5 epoch), one pass labels every nonzero with its batch
 * (key[e] = batch_of[idx[e]]), a scan over {batch, segment of the other side} counts gives every segment's place in
 * every batch, and one scatter writes e_idx / e_y = the other side's layout PARTITIONED by batch (stable: inside a batch
 * the order of hpf_hip_svi_batch_prepare's o_idx / o_y).  Each batch then owns fixed-capacity slices:
 *   flag_own [nb][own_nrows], flag_oth [nb][oth_nrows], b_segs [nb][b_segs_cap], b_multi [nb][multi_cap][3],
 *   o_segs [nb][o_segs_cap] (their `begin` indexes e_idx / e_y as a whole), o_multi [nb][multi_cap][3], sizes [nb][8]
 * with the meaning hpf_hip_svi_batch_prepare gives them for that batch (the per-batch call and this one produce the
 * same segment lists, flags and nonzero order -- tests/test_svi_paths.py).  acc_own rows of rows without nonzeros are zeroed.
 * nb <= 255.  Scratch: batch_of [own_nrows] u8, key [oth_nnz] u8, seg_cnt [nb*oth_nseg + 1] i32,
 * seg_pos [nb*oth_nseg + 1] i64, tiles [hpf_hip_svi_epoch_scratch_words(nb)] i64.  sizes[b][7] is sticky as above.
 */
typedef struct hpf_svi_epoch {
    const hpf_segment *own_segs; int64_t own_nseg; const int64_t *own_row_seg_ptr; const int64_t *own_indptr;
    int64_t own_nrows;
    const hpf_segment *oth_segs; int64_t oth_nseg; const int64_t *oth_row_seg_ptr;
    const int32_t *oth_idx; const float *oth_y; int64_t oth_nrows; int64_t oth_nnz;
    const int64_t *order; int64_t per; int32_t nb, ld, seg_cap, reserved;
    float *acc_own; uint8_t *batch_of; uint8_t *flag_own; uint8_t *flag_oth;
    hpf_segment *b_segs; int64_t b_segs_cap; int64_t *b_multi; int64_t multi_cap;
    int32_t *e_idx; float *e_y; hpf_segment *o_segs; int64_t o_segs_cap; int64_t *o_multi;
    int64_t *sizes;
    uint8_t *key; int32_t *seg_cnt; int64_t *seg_pos; int64_t *tiles;
} hpf_svi_epoch;
int64_t hpf_hip_svi_epoch_sizeof(void);
int64_t hpf_hip_svi_epoch_scratch_words(int nb);
int hpf_hip_svi_epoch_prepare(const hpf_svi_epoch *epoch, void *stream);
/*
 * A batch handed over as COO triplets in the caller's order (the extension's partial_fit, PXI:423-434): no resident CSR to
 * slice.  hpf_hip_svi_coo_narrow: the reference's size_t ids (as int64) -> int32 row ids + the range check the reference
 * does not make (err[0] = 1 when an id is < 0 or >= limit; the id is replaced by 0 so that nothing reads out of bounds).
 * hpf_hip_svi_coo_prepare: ONE grouping of the batch from its row ids STABLY SORTED (key[n], ascending; the nonzeros'
 * other-side ids and counts permuted alike by the caller): flag[nrows] (0: row absent, 1: present in one segment, 2: a
 * split row), segs (their `begin` indexes the sorted arrays), multi = {first segment, segments, row} per split row, sizes
 * as hpf_hip_svi_batch_prepare gives them for the OTHER side ([2] segments, [3] split rows, [4] nonzeros, [5] rows present,
 * [7] sticky overflow flag).  Scratch: row_start [nrows] int64, row_cnt [nrows] int32, tiles
 * [hpf_hip_svi_prep_scratch_words()] int64.  Nothing is read back.
 */
typedef struct hpf_svi_coo {
    const int32_t *key; int64_t n; int64_t nrows; int32_t seg_cap, reserved;
    uint8_t *flag; int64_t *row_start; int32_t *row_cnt;
    hpf_segment *segs; int64_t segs_cap; int64_t *multi; int64_t multi_cap;
    int64_t *sizes; int64_t *tiles;
} hpf_svi_coo;
int64_t hpf_hip_svi_coo_sizeof(void);
int hpf_hip_svi_coo_narrow(const int64_t *ids, int64_t n, int64_t limit, int32_t *out, int64_t *err, void *stream);
int hpf_hip_svi_coo_prepare(const hpf_svi_coo *coo, void *stream);
/* acc[row][0:ld] = sum of part[first .. first+n) for the descriptors {first, n, row} (b_multi / o_multi above), the
 * first min(ndesc_max, *ndesc_dev) of them: the split rows of a batch sweep (hpf_hip_segsum_f32 with device-side lists). */
int hpf_hip_segsum_desc_f32(const float *part, const int64_t *desc, const int64_t *ndesc_dev, int64_t ndesc_max,
                            float *acc, int ld, void *stream);

/*
 * initialize_parameters (PXI:127-141) without the host, in two steps.
 * hpf_hip_mt19937_words: the next n words of an MT19937 stream, raw[0..n) -- the recurrence's state words, before the
 *   tempering that turns a word into an output.  `state` is uint32[625] in device memory: numpy's 624 key words + its
 *   `pos` (bit_generator.state); it is left at the stream's new position, so consecutive calls continue one stream like
 *   consecutive numpy draws.  The recurrence only parallelises over 227 words, so short draws (and any draw without
 *   scratch) are walked by ONE workgroup (0.27 us per 624 words).  Long draws are cut into 512-1024 chunks, each walked
 *   by its own workgroup from a state obtained by polynomial jump-ahead (x^n mod the characteristic polynomial of the
 *   generator, applied as a 19937-term XOR over a window of the stream; hpfrec_amd/csrc/hpf_mt19937.hip): `scratch` must
 *   then hold hpf_hip_mt19937_scratch_words(n) uint32 words of device memory (0: no scratch needed for this n).
 * hpf_hip_mt19937_jump_poly: out[0..624) = the coefficients of x^(624 * 2^q) mod phi, bit i of word i/32 (HOST memory;
 *   lets a test check the jump against a generator's own stepping without a GPU).
 * hpf_hip_uniform_rows_f32: nrows*k stored words -> numpy's Generator.random(dtype=float32) values (tempering, then
 *   (y >> 8) * 2^-24), laid out row-major with leading dimension ld:
 *     out[r*ld + j] = base + scale*U[r*k + j]       (two float32 roundings, like `a_prime + 0.01 * draw`)
 *   and, when `ratio` is given, ratio[...] = out[...] / den[...] correctly rounded (Theta = Gamma_shp / Gamma_rte,
 *   PXI:140-141; den and ratio are laid out like out).  Pad columns are not written.  A rank of a sharded fit passes
 *   the words of its own rows (raw + row0*k).
 */
int64_t hpf_hip_mt19937_scratch_words(int64_t n);
int hpf_hip_mt19937_jump_poly(int q, uint32_t *out);
int hpf_hip_mt19937_words(uint32_t *state, uint32_t *raw, int64_t n, uint32_t *scratch, void *stream);
int hpf_hip_uniform_rows_f32(const uint32_t *raw, float *out, const float *den, float *ratio, int64_t nrows, float base,
                             float scale, int k, int ld, void *stream);

/* Measurement aid, not part of the replaced path: the sweep's gather pattern (256-byte rows of a [rows][64]
 * table, row ids from idx[], 4 rows per wave step, 8 steps in flight) with the arithmetic stripped; used by
 * tools/gather_probe.py to measure the gather-bandwidth ceiling the sweep kernel is held against. */
int hpf_hip_gather_probe_f32(const int32_t *idx, int64_t n, const float *tab, float *sink, int grid_blocks,
                             void *stream);

/* out[r] = <vec, tab[r]>, r < nrows; vec is one padded row (ld floats).  Replaces the scoring product of
 * HPF.topN, Theta[user].dot(Beta.T) (hpfrec/__init__.py:1337). */
int hpf_hip_score_rows_f32(const float *vec, const float *tab, int64_t nrows, float *out, int k, int ld,
                           void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Multi-GPU: one rank's whole sharded iteration issued by ONE call (hpfrec_amd/csrc/hpf_shard.hip).
 *
 * The reference is single-node OpenMP (PXI:227-259); SURVEY.md section 8(e) shards the users over the GPUs of a node
 * and exchanges the item statistics once per iteration.  The Python driver (hpfrec_amd/cavi.py, _iterate_scatter)
 * issues ~12 kernel launches, 6 collectives and ~12 stream/event operations per iteration through ctypes and
 * torch.distributed: 0.26 ms of host time inside a 0.6 ms iteration at 8 ranks.  The entries below issue the same
 * schedule from C: the kernels through the entry points above, the collectives through RCCL's C API on a communicator
 * of the caller's, the dependencies through HIP events on two streams.
 *
 * RCCL is NOT linked: hpf_hip_rccl_open() resolves its entry points from the librccl.so the host process has already
 * loaded (PyTorch ships and loads its own), so that one RCCL instance serves torch.distributed and this library.
 * A plan retains host-side state only (the descriptor, a few hipEvents); all device memory stays the caller's.
 * ------------------------------------------------------------------------------------------------------------------ */
int hpf_hip_rccl_open(const char *librccl_path);   /* NULL: "librccl.so" by the loader's search; idempotent */
int hpf_hip_rccl_unique_id(uint8_t id[128]);        /* ncclGetUniqueId (rank 0; the id travels by the caller's means) */
int hpf_hip_rccl_comm_init(void **comm, int world, int rank, const uint8_t id[128]);  /* ncclCommInitRank, current device */
int hpf_hip_rccl_comm_count(void *comm, int *count);                                  /* ncclCommCount */
int hpf_hip_rccl_comm_destroy(void *comm);
/* float32 sum collectives on `stream` (in place all-reduce; recv_n / send_n = elements PER RANK) */
int hpf_hip_rccl_all_reduce_f32(void *comm, float *buf, int64_t n, void *stream);
int hpf_hip_rccl_reduce_scatter_f32(void *comm, const float *send, float *recv, int64_t recv_n, void *stream);
int hpf_hip_rccl_all_gather_f32(void *comm, const float *send, float *recv, int64_t send_n, void *stream);

#define HPF_COLL_ALL_REDUCE 0      /* recv == send, count elements                                  */
#define HPF_COLL_REDUCE_SCATTER 1  /* count = elements received per rank (send holds world*count)   */
#define HPF_COLL_ALL_GATHER 2      /* count = elements sent per rank (recv holds world*count)       */
/* A stand-in for RCCL (tests: gloo ranks sharing one GPU): called instead of RCCL when given; must leave the result
 * usable by work queued on `stream` afterwards.  Returns 0 on success. */
typedef int (*hpf_collective_fn)(void *ctx, int op, const float *send, float *recv, int64_t count, void *stream);

/* One item range of the exchange: table rows [lo, hi), (hi - lo) % world == 0 (the last range may run past nI into pad
 * rows of the item tables); its segments are segs[seg_lo .. seg_lo + nseg) of the item side. */
typedef struct hpf_shard_range {
    int64_t lo, hi;
    int64_t seg_lo, nseg;
    const int64_t *multi_rows; /* device: rows of the range that are split into several segments or have none */
    int64_t nmulti;
    int32_t short_rows;        /* launch hint of hpf_hip_sweep_f32 for this range */
    int32_t reserved;
} hpf_shard_range;

/* Everything one rank's iteration touches ("scatter" exchange).  All table pointers are device
 * memory owned by the caller and must stay valid while the plan lives. */
typedef struct hpf_shard_desc {
    int32_t world, rank, k, ld;
    int64_t nU, nI;                /* this rank's users; all items (the item tables hold ranges[last].hi rows) */
    /* user side (CSR of this rank's users) */
    const hpf_segment *u_segs; int64_t u_nseg; const int32_t *u_idx; const float *u_y;
    const int64_t *u_row_seg_ptr; const int64_t *u_multi_rows; int64_t u_nmulti;
    /* item side (CSC of this rank's nonzeros) */
    const hpf_segment *i_segs; const int32_t *i_idx; const float *i_y; const int64_t *i_row_seg_ptr;
    int32_t nranges, pad0; hpf_shard_range ranges[HPF_MAX_ROW_RANGES];   /* in issue order */
    /* tables ([rows][ld] unless noted) */
    float *eB; float *part_u; float *part_i;
    float *Gamma_shp, *Theta, *k_rte, *k_rte_prev;       /* user outputs ([nU] scalars) */
    float *Lambda_shp, *Beta, *t_rte, *t_rte_prev;       /* item outputs, current on the owning rank */
    float *csT, *csB, *csB_used;                          /* [ld] column sums */
    float *csT_part; int32_t csT_part_rows, user_sweep_grid, user_multi_grid, pad1;
    float *csB_part; int32_t csB_part_rows, pad2;        /* = finalize grid of this rank's item slices */
    float *acc_i;                  /* [ranges[last].hi][k] packed exchange buffer (reduce-scatter input) */
    float *acc_own;                /* [sum of slice lengths][k] reduce-scatter outputs, range after range */
    float *e_own;                  /* [sum of slice lengths][e_own_ld] new E rows of the slices (all-gather input) */
    int32_t e_own_ld, item_sweep_grid;   /* e_own_ld = ld (schedule 0: all-gather straight into eB) or
                                            hpf_hip_gather_payload_ld(k) (schedules 1 and 3: [numerators | base] rows) */
    float *ag_recv;                /* schedules 1 and 3: [world][sum of slice lengths][e_own_ld] gathered rows */
    float a, k_shp, add_k_rte, c, t_shp, add_t_rte;
    void *comm;                    /* ncclComm_t (hpf_hip_rccl_comm_init), or NULL */
    hpf_collective_fn coll; void *coll_ctx;   /* used instead of RCCL when coll != NULL */
    void *xstream;                 /* the exchange stream (hipStream_t) */
    int32_t dry_run;               /* 1: this rank alone -- every collective is its one-rank form (local copy of the
                                      rank's slice) + a 1-element all-reduce on comm if given: the compute-only
                                      schedule of a rank, for probes and the bench's exposed-exchange figure */
    int32_t schedule;              /* HPF_SCHEDULE_FINALIZE_THEN_GATHER (0), _GATHER_EARLY (1), _DIRECT (3) */
    float *shp_own;                /* gather-early: [sum of slice lengths][ld] shapes between the finalizer's halves */
    float dry_run_busbw_GBps;      /* dry run only, > 0: every collective additionally occupies its stream for
                                      latency + bytes * (world-1)/world / busbw -- one idle-spinning wavefront (the links
                                      do the work on a real node), so that a one-GPU probe shows what each schedule hides */
    float dry_run_latency_us;
    int32_t dry_run_footprint_blocks;  /* dry run with busbw > 0: the stand-in of a bulk collective is this many workgroups
                                      of 256 threads, 128 VGPRs and 64 KB of LDS each (what a collective library's
                                      kernel needs to be RESIDENT beside the sweeps), each holding its slot for the link
                                      time; 0: one wavefront */
    int32_t direct_prefetch;       /* schedule 3: 1 = the peers' finished rows are copied into ag_recv by ONE pull launch
                                      on the exchange stream (under the user sweep) and the apply kernel reads local
                                      memory; 0 = the apply kernel reads the owners' buffers itself */
    int32_t direct_pull_grid, direct_gather_gx;   /* schedule 3: workgroups of a slice's pull-reduce launch; workgroups per
                                      OWNER of the pull of the finished rows (its workgroups poll: keep it small) */
    void *p2p_region;              /* schedule 3: the rank's connected exchange region (hpf_hip_p2p_region_create /
                                      _connect); acc_i and e_own must lie INSIDE its data buffer, at the offsets below */
    int64_t p2p_acc_offset, p2p_send_offset;   /* bytes from the start of the region's data buffer */
} hpf_shard_desc;

/* Three schedules of the same exchange.
 * 0, finalize-then-gather: after the user side, all-reduce colsum(Theta), finish this rank's item slices
 *    (hpf_hip_row_finalize_ranges_f32), all-gather the new E rows range by range; the next iteration's sweep of a range waits
 *    for that range's all-gather -- the all-gather hides only under the item sweeps of the other ranges.
 * 1, gather-early: the finalizer is split (hpf_hip_item_shape_rows_f32 / hpf_hip_item_apply_rows_f32): the shape / psi
 *    half runs right after the reduce-scatters and its [k numerators | base rate] rows (e_own, e_own_ld =
 *    hpf_hip_gather_payload_ld(k)) are all-gathered in ONE collective into ag_recv ([world][sum of slice lengths][that
 *    ld]) WHILE THE USER SWEEP RUNS, the shapes wait in shp_own ([sum of slice lengths][ld]); after the
 *    all-reduce of colsum(Theta) (on the compute stream: no stream hand-over) every rank applies the rates to all items
 *    locally.  csB_part then has the grid of the apply kernel (csB_part_rows blocks over nI rows).  The exchange leaves
 *    the critical path except for two k-float all-reduces.
 * (2 was "gather-carried", rounds 3-4: gather-early with the exchange running on into the next iteration on two RCCL
 *    communicators; the direct exchange hides more link time without a collective library and replaced it.)
 * 3, direct: gather-early WITHOUT collectives (section "Multi-GPU, direct exchange" below).  acc_i and e_own live in the
 *    rank's peer-mapped region.  Compute stream: item sweeps (the launch after a range's sweep tells every rank, on entry,
 *    that the range is complete -- no stream event ties the two streams), user side, colsum(Theta) summed over the ranks
 *    inside its reduction kernel (granules), the apply half, colsum(Beta) the same way.  Exchange stream, per range: one
 *    wavefront waits for every rank's flag, then the slice's N partial accumulator rows are PULLED out of the N ranks'
 *    buffers and summed in rank order into acc_own (the reduce-scatter); the shape half of all slices; then either one
 *    pull launch copies every owner's finished rows into ag_recv under the user sweep (direct_prefetch) or the apply
 *    kernel reads them from the owners' buffers itself (the all-gather).  No RCCL call, no collective kernel beside the
 *    sweeps.  comm / coll are not used; a dry run is a region connected to itself (hpf_hip_p2p_region_connect(region, NULL)). */
#define HPF_SCHEDULE_FINALIZE_THEN_GATHER 0
#define HPF_SCHEDULE_GATHER_EARLY 1
#define HPF_SCHEDULE_DIRECT 3

/* dry_run == 2, "trace": nothing is issued and no device is needed -- every kernel launch, collective, event record /
 * wait and copy of hpf_hip_shard_iterate / _join / _exchange_only is appended to the plan's trace instead, in issue order
 * (table pointers are never dereferenced, streams are opaque values, comm / coll are ignored).  What a multi-rank run
 * depends on -- identical collective sequences on all ranks, every wait after its record -- can then be checked on any
 * machine (tests/test_host_logic.py). */
#define HPF_TRACE_KERNEL 1      /* id = HPF_TRACE_K_*; arg = 1 + flag kind when the launch raises a flag on entry */
#define HPF_TRACE_COLLECTIVE 2  /* id = HPF_COLL_* (| 0x100: a k-float all-reduce), arg = count    */
#define HPF_TRACE_RECORD 3      /* arg = event                                                     */
#define HPF_TRACE_WAIT 4        /* arg = event                                                     */
#define HPF_TRACE_COPY 5        /* arg = bytes                                                     */
#define HPF_TRACE_K_SWEEP 1
#define HPF_TRACE_K_SEGSUM 2
#define HPF_TRACE_K_SWEEP_FINALIZE 3
#define HPF_TRACE_K_ROW_FINALIZE 4
#define HPF_TRACE_K_ROW_FINALIZE_RANGES 5
#define HPF_TRACE_K_COLSUM_REDUCE 6
#define HPF_TRACE_K_ITEM_SHAPE 7
#define HPF_TRACE_K_ITEM_APPLY 8
#define HPF_TRACE_K_PULL_REDUCE 10       /* arg = the item range whose slice is pulled and summed            */
#define HPF_TRACE_K_GATHER_PULL 11       /* arg = signal kind | done kind << 8                              */
#define HPF_TRACE_K_COLSUM_ALLREDUCE 12  /* arg = HPF_P2P_VEC_* | (1 + own flag kind waited for afterwards) << 8 | peer kinds << 16 */
#define HPF_TRACE_K_SIGNAL 13            /* arg = flag kind (a launch that only raises a flag)              */
#define HPF_TRACE_K_WAIT 14              /* arg = the flag kinds one wavefront waits for, from every rank    */
/* events of a traced plan: HPF_TRACE_EVENT_BASE + 2j (range j swept), + 2j + 1 (range j's all-gather done), then, after the
 * 2 * nranges of them: colsum(Theta) ready, iteration start, apply done, colsum(Beta) done */
#define HPF_TRACE_EVENT_BASE 0x1000
typedef struct hpf_shard_trace_rec {
    int32_t kind, id;
    int64_t stream;     /* the stream argument as given (compute stream of the call, xstream, sstream) */
    int64_t arg;
} hpf_shard_trace_rec;
/* *n = records waiting; when cap >= *n they are copied to out and the trace is emptied (cap < *n: a size query) */
int hpf_hip_shard_trace(void *plan, hpf_shard_trace_rec *out, int64_t cap, int64_t *n);

/* {sizeof(hpf_shard_desc), offsetof ranges, offsetof acc_i, offsetof dry_run}: lets a foreign-function binding check
 * its mirror of the struct against the compiled one. */
int hpf_hip_shard_desc_layout(int64_t out[4]);
int hpf_hip_shard_plan_create(const hpf_shard_desc *desc, void **plan);
int hpf_hip_shard_plan_destroy(void *plan);
/* One iteration on compute_stream (+ the plan's exchange stream): item sweeps per range -> reduce-scatter; the user
 * side (sweep fused with its finalizer, split rows, colsum) under the exchange; then on the exchange stream the
 * all-reduce of colsum(Theta), the finalizer of this rank's item slices, all-gathers of the new E rows (each waited
 * for by the next iteration's sweep of that range only) and the all-reduce of colsum(Beta).  eT: the users' current E
 * table, eT_next: receives the new one (the caller swaps).  store = 0 skips the six [n][k] output tables. */
int hpf_hip_shard_iterate(void *plan, const float *eT, float *eT_next, int store, void *compute_stream);
/* `stream` waits for everything the plan has in flight on its exchange stream (call before reading the item tables,
 * before work of ANOTHER communicator, and before destroying the plan); the next iterate re-synchronises. */
int hpf_hip_shard_join(void *plan, void *stream);
/* schedule 3: synchronises the device and returns HPF_ETIMEOUT when a wait of this rank ran out (0 otherwise, and for
 * the other schedules) */
int hpf_hip_shard_status(void *plan);
/* 1 (one collective of op HPF_COLL_*) of `count` per-rank elements between scratch regions of the plan's buffers, on
 * `stream`: the exchange-only timing pass of bench.py.  range < 0: the k-float all-reduce. */
int hpf_hip_shard_exchange_only(void *plan, int op, int range, void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Multi-GPU, direct exchange (hpfrec_amd/csrc/hpf_p2p.hip, hpf_p2p_dev.h): the item statistics move between the GPUs
 * of a node WITHOUT a collective library.  SURVEY.md section 8(e): "must use a direct reduce-scatter + all-gather ...
 * all 7 links ... rather than a ring".  Every rank owns a REGION -- a data buffer (caller-defined layout; coarse-grained
 * device memory) and a control block (flags, vector slots, an error word; fine-grained device memory) -- exports both
 * (hipIpcGetMemHandle) and maps every peer's (hipIpcOpenMemHandle).  The kernels of the iteration then read the peers'
 * buffers directly: data is only ever PULLED by its consumer after the producer's flag (a monotonic epoch) has arrived
 * in the consumer's own control block; k-float vectors travel as 8-byte {value, epoch} granules.  Every wait is bounded
 * by the region's time-out: a missing peer sets the error word (HPF_ETIMEOUT at the next status call), it cannot hang
 * the GPU.  The reference has no counterpart (single-node OpenMP, cython_loops.pxi:4).
 * ------------------------------------------------------------------------------------------------------------------ */
#define HPF_P2P_MAX_RANKS 16
#define HPF_P2P_HANDLE_BYTES 64     /* sizeof(hipIpcMemHandle_t) */
#define HPF_P2P_NKINDS 32           /* flag kinds per control block (the schedule's: HPF_P2P_FLAG_*) */
#define HPF_P2P_NVEC 2              /* vector slots: HPF_P2P_VEC_CST, HPF_P2P_VEC_CSB */
#define HPF_P2P_VEC_CST 0
#define HPF_P2P_VEC_CSB 1
#define HPF_P2P_FLAG_SWEPT(j) (j)                          /* item range j of this epoch is complete in the rank's acc buffer */
#define HPF_P2P_FLAG_SHAPED(j) (HPF_MAX_ROW_RANGES + (j))  /* the rank's [numerators | base] rows of range j are complete */
#define HPF_P2P_FLAG_GATHERED 29                           /* (local) every owner's rows of this epoch are in ag_recv */
#define HPF_P2P_FLAG_USER 30                               /* free for callers (tests, probes) */
int64_t hpf_hip_p2p_ctrl_bytes(int ld);
/* Allocates the region of `rank` of `world` on the current device (data_bytes of zeroed data + the control block for
 * vectors of ld floats). */
int hpf_hip_p2p_region_create(int world, int rank, int ld, int64_t data_bytes, void **region);
/* out: [control-block handle | data handle], 2 x HPF_P2P_HANDLE_BYTES */
int hpf_hip_p2p_region_handles(void *region, uint8_t out[2 * HPF_P2P_HANDLE_BYTES]);
/* handles: world x 2 x HPF_P2P_HANDLE_BYTES bytes, rank after rank (this rank's own entry is ignored).  NULL: this rank
 * ALONE -- every peer is mapped to the local memory and no flag is ever waited for (probes; the compute-only twin of the
 * bench).  Call once, after every rank has created its region. */
int hpf_hip_p2p_region_connect(void *region, const uint8_t *handles);
int hpf_hip_p2p_region_data(void *region, int peer, void **ptr);       /* peer's data buffer as mapped here */
int hpf_hip_p2p_region_set_timeout(void *region, float timeout_ms);    /* default 20 s */
int hpf_hip_p2p_region_next_epoch(void *region, uint32_t *epoch);      /* ++epoch (starts at 1; every rank counts alike) */
/* synchronises the device and reads the error word: 0, or HPF_ETIMEOUT with *err = the bit set (bit f: a flag of kind
 * f mod 16 never came; bit 16 + v: a vector granule).  The word is sticky for the life of the region, and once it is set no
 * later wait of this rank polls at all: the launches already queued drain at once (their results are garbage), so a dead
 * peer costs one time-out budget, not one per queued wait. */
int hpf_hip_p2p_region_status(void *region, uint32_t *err);
int hpf_hip_p2p_region_destroy(void *region);
/* primitive stream operations (tests, probes; the iteration has them fused into its kernels):
 * signal: after everything queued on `stream` so far, flags[kind][this rank] = epoch in every rank's control block;
 * wait:   `stream` continues once flags[kind][src] >= epoch for every src in src_mask;
 * allreduce_vec: vec[0..ld) summed over the ranks in rank order (in place; every rank gets the same floats);
 * pull:   dst[0..n) = n floats at src_offset_bytes of src_rank's data buffer, after flags[kind][src_rank] >= epoch
 *         (kind < 0: no wait). */
int hpf_hip_p2p_signal(void *region, int kind, uint32_t epoch, void *stream);
int hpf_hip_p2p_wait(void *region, int kind, uint32_t epoch, uint32_t src_mask, void *stream);
int hpf_hip_p2p_allreduce_vec_f32(void *region, int which, uint32_t epoch, float *vec, void *stream);
int hpf_hip_p2p_pull_f32(void *region, int kind, uint32_t epoch, int src_rank, int64_t src_offset_bytes, float *dst,
                         int64_t n, int grid_blocks, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* HPF_HIP_H */
