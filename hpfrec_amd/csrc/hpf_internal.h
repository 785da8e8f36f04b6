// hpf_internal.h -- launchers shared between the translation units of libhpf_hip.so (C++ linkage, not part of the C ABI):
// the direct-exchange forms of the iteration's kernels (hpf_hip.hip), used by the C-issued schedule (hpf_shard.hip).
// Each is the public entry of the same name in include/hpf_hip.h plus the peer-memory operands of hpf_p2p_dev.h.
#ifndef HPF_INTERNAL_H
#define HPF_INTERNAL_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hpf_hip.h"
#include "hpf_p2p_dev.h"

namespace hpf_direct {

// "what this stream wrote into the rank's exchange buffer so far is complete": raised by block 0 of the launch that
// carries it, on entry (peers_dev: a hpf_p2p::Peers in device memory; null: no signal)
struct Signal {
    const hpf_p2p::Peers *peers_dev;
    int kind;
    uint32_t epoch;
};

// hpf_hip_sweep_f32 + a signal
int sweep(const hpf_segment *segs, int64_t nseg, const int32_t *idx, const float *y, const float *tab_self,
          const float *tab_other, float *part, float *acc_rows, int acc_ld, int k, int ld, int short_rows, int grid_blocks,
          Signal sig, hipStream_t st);
// hpf_hip_sweep_finalize_f32 + a signal + a copy of cs_other (what the launch used), written by block 0
int sweep_finalize(const hpf_segment *segs, int64_t nseg, const int32_t *idx, const float *y, const float *tab_self,
                   const float *tab_other, float *part, float *e_new, float *shp, float *rte, float *fac, float *rs,
                   float *rs_prev, const float *cs_other, float *cs_partial, float prior_shp, float top_shp,
                   float add_rte, int k, int ld, int grid_blocks, float *cs_other_copy, Signal sig, hipStream_t st);
// hpf_hip_item_apply_rows_f32 with the rows of owner o read from blocks[o] (wait_kind >= 0: after flags[wait_kind][o],
// or -- local_flag -- this rank's own flags[wait_kind][rank], has reached epoch)
int item_apply_blocks(const float *const *blocks, int nblocks, int wait_kind, int local_flag, uint32_t epoch,
                      const hpf_p2p::Peers &pp, const float *shp_own, float *e_tab, float *shp, float *fac, float *rs,
                      const float *cs_other, float *cs_partial, float add_rte, int k, int ld, int rank, int world,
                      int64_t nrows, int nranges, const int64_t *range_lo, const int64_t *range_hi, int grid_blocks,
                      hipStream_t st);
// hpf_hip_colsum_reduce_f32 whose result is the sum over ALL ranks (granules, rank order; which = HPF_P2P_VEC_*); the
// launch then also WAITS (one workgroup) for the flag kinds of then_wait_kinds from every peer and for this rank's own
// flag then_wait_self (-1: none), on behalf of the large-grid launch that follows it on the stream
int colsum_reduce_allreduce(const float *cs_partial, int nblk, float *cs_out, int ld, const hpf_p2p::Peers *peers_dev,
                            int which, uint32_t epoch, uint32_t then_wait_kinds, int then_wait_self, hipStream_t st);

}  // namespace hpf_direct

// hpf_p2p.hip: the finished [numerators | base] rows of every owner copied into the local gathered buffer
// (dst + o * rows_per_rank * row_floats for owner o), one launch, grid (gx, world).  Block (0, 0) first raises
// flags[signal_kind][rank] = epoch in every peer (this rank's rows are complete: they were written by earlier launches of
// this stream); the blocks of owner o then wait for o's flag.  The last block to finish raises this rank's OWN
// flags[done_kind][rank] (the apply kernel on the compute stream polls it instead of a stream event).  copy_own = 0: this
// rank's own rows are not copied (their reader takes them from the source buffer).
namespace hpf_p2p {
int gather_pull(void *region, int64_t src_offset_bytes, float *dst, int64_t floats_per_rank, int signal_kind, int done_kind,
                uint32_t epoch, int gx, int copy_own, hipStream_t st);
// dst[0..n) = the rank-order sum over the ranks of the n floats at src_offset_bytes of every rank's data buffer (ranks
// outside sum_mask are read, not counted); wide loads when offset, n and dst allow.  Put wait_flags() ahead of it.
int pull_reduce(void *region, int64_t src_offset_bytes, float *dst, int64_t n, uint32_t sum_mask, int grid_blocks,
                hipStream_t st);
// `st` continues once every peer in src_mask has raised every flag kind of the bit mask `kinds` to epoch and (self_kind >= 0)
// this rank its own flag self_kind: ONE waiting wavefront.  Put in front of every consumer with a large grid.
int wait_flags(void *region, uint32_t kinds, uint32_t epoch, uint32_t src_mask, int self_kind, hipStream_t st);
// this rank's Peers in device memory (plan-lifetime copy inside the region's control block header is not possible:
// the block is fine-grained; a small plain allocation owned by the region)
const Peers *region_peers_dev(void *region);
}  // namespace hpf_p2p
#endif  // HPF_INTERNAL_H
