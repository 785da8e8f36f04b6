// hpf_svi_prep.hip -- the index structures of one stochastic batch, built on the device by ONE host call.
//
// What this replaces (reference: /root/reference/hpfrec/cython_loops.pxi = "PXI"): an SVI epoch slices its data per batch
// with numpy fancy indexing on the host -- the batch's rows of the CSR (PXI:280-290) or of scipy's CSC (PXI:332-342),
// and the set of other-side rows they touch (get_unique_items_batch, PXI:27-42).  Rounds 1-2 of this build did the
// equivalent with host numpy over a host copy of the row pointers, a pinned upload, a device sort, a histogram and a few
// size read-backs per batch: 0.35 s of host work in a 0.56 s epoch loop at BASELINE config C5
// (profiles/r02_final_svi_c5.txt).  Here a batch costs the host one call and no read-back:
//
//   own side   (the batch's rows of the side it is drawn from): nothing is copied -- the batch's segment list is the
//              stable compaction of the side's GLOBAL segment list by a per-row flag, its descriptors keep pointing into the
//              global idx / y arrays;
//   other side (the same nonzeros grouped by the other side's rows): the other side's global layout already holds every
//              row's nonzeros in ascending order of the own side's ids, so the batch's view of it is that layout FILTERED by
//              the flag -- a count pass, a scan over rows, a write pass: stable, deterministic, no sort.
//
// Every data-dependent size stays on the device (`sizes`); the consumers (hpf_hip_sweep_f32, hpf_hip_expect_f32,
// hpf_hip_segsum_desc_f32) read their trip counts from there, and the buffers are sized once per fit from a bound.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hpf_hip.h"

namespace {

constexpr int WAVE = 64;
constexpr int BLOCK = 256;
constexpr int WPB = BLOCK / WAVE;
constexpr int TILES = 1024;           // scan tiles (= blocks of the two-launch scans)

// exclusive prefix of NC int64 components over the threads of a block, plus the block totals
template <int NC>
__device__ __forceinline__ void block_exclusive_scan(const long long (&v)[NC], long long (&excl)[NC], long long (&total)[NC]) {
    __shared__ long long wsum[WPB][NC];
    const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x >> 6;
    long long inc[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        long long x = v[c];
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) {
            const long long y = __shfl_up(x, d);
            if (lane >= d) x += y;
        }
        inc[c] = x;
    }
    __syncthreads();                  // (wsum may still be read by the previous call)
    if (lane == WAVE - 1) {
#pragma unroll
        for (int c = 0; c < NC; c++) wsum[wid][c] = inc[c];
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NC; c++) {
        long long before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < WPB; w++) {
            const long long s = wsum[w][c];
            if (w < wid) before += s;
            all += s;
        }
        excl[c] = before + inc[c] - v[c];
        total[c] = all;
    }
}

// sum over tiles before `tile` of the per-tile totals (NC components each) -- every block does its own: TILES is small
template <int NC>
__device__ __forceinline__ void tiles_before(const long long *__restrict__ tiles, int tile, long long (&before)[NC],
                                             long long (&all)[NC], int ntile = TILES) {
    long long mine[NC], tot[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) mine[c] = tot[c] = 0;
    for (int t = threadIdx.x; t < ntile; t += BLOCK) {
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const long long x = tiles[(size_t)t * NC + c];
            tot[c] += x;
            if (t < tile) mine[c] += x;
        }
    }
    long long e1[NC], e2[NC];
    block_exclusive_scan<NC>(mine, e1, before);
    block_exclusive_scan<NC>(tot, e2, all);
}

// (1) flags of the batch's rows: the previous batch that used this workspace is unmarked, the new one marked; a batch row
// without any nonzero gets its accumulator row zeroed (the dense step reads acc[row] of every flagged row)
// (flag values: 1 = the row is present in exactly one segment -- the batch-side sweep finishes it itself, hpf_hip_sweep_svi_batch_f32;
//  2 = a split row or a row without nonzeros: finished by the whole-table pass; every consumer that only asks "is the row in
//  the batch" tests flag != 0)
__global__ __launch_bounds__(BLOCK) void svi_mark_kernel(const int64_t *__restrict__ prev_ids, int64_t nprev,
                                                         const int64_t *__restrict__ ids, int64_t nids,
                                                         uint8_t *__restrict__ flag, const int64_t *__restrict__ indptr,
                                                         const int64_t *__restrict__ row_seg_ptr,
                                                         float *__restrict__ acc, int ld, int phase) {
    const int64_t stride = (int64_t)gridDim.x * BLOCK;
    if (phase == 0) {
        for (int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x; t < nprev; t += stride) flag[prev_ids[t]] = 0;
        return;
    }
    for (int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x; t < nids; t += stride) {
        const int64_t r = ids[t];
        flag[r] = (row_seg_ptr[r + 1] - row_seg_ptr[r] == 1) ? 1 : 2;
        if (indptr[r + 1] == indptr[r]) {           // (ld is a multiple of 32 and rows are 128-byte aligned)
            float4 *row = reinterpret_cast<float4 *>(acc + (size_t)r * ld);
            for (int c = 0; c < ld / 4; c++) row[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}

// (2a) own side: per tile of the side's global segment list, how many segments belong to flagged rows and how many of
// those open a split row
// EPOCH (second half of this file): `flag` holds the BATCH of every row, blockIdx.y is the batch this block compacts for,
// gridDim.x tiles per batch; every per-batch output is a fixed-capacity slice
template <bool EPOCH>
__global__ __launch_bounds__(BLOCK) void svi_own_count_kernel(const hpf_segment *__restrict__ segs, int64_t nseg,
                                                              const uint8_t *__restrict__ flag,
                                                              long long *__restrict__ tiles) {
    const int ntile = gridDim.x, want = EPOCH ? (int)blockIdx.y : 1;
    tiles += (size_t)blockIdx.y * ntile * 2;
    const int64_t per = (nseg + ntile - 1) / ntile;
    const int64_t s0 = (int64_t)blockIdx.x * per, s1 = min(nseg, s0 + per);
    long long v[2] = {0, 0};
    for (int64_t s = s0 + threadIdx.x; s < s1; s += BLOCK) {
        const hpf_segment g = segs[s];
        if (EPOCH ? flag[g.row] == want : flag[g.row] != 0) {
            v[0]++;
            // the first segment of a split row: not flagged whole-row, and the previous segment is another row's
            if (!(g.len & HPF_SEG_WHOLE_ROW) && (s == 0 || segs[s - 1].row != g.row)) v[1]++;
        }
    }
    long long e[2], tot[2];
    block_exclusive_scan<2>(v, e, tot);
    if (threadIdx.x == 0) {
        tiles[(size_t)blockIdx.x * 2 + 0] = tot[0];
        tiles[(size_t)blockIdx.x * 2 + 1] = tot[1];
    }
}

// (2b) own side: the stable compaction itself.  b_segs = the flagged rows' descriptors (unchanged: they index the global
// idx / y); b_multi[m] = {first compacted segment, segments, row} of every split row of the batch
template <bool EPOCH>
__global__ __launch_bounds__(BLOCK) void svi_own_write_kernel(const hpf_segment *__restrict__ segs, int64_t nseg,
                                                              const int64_t *__restrict__ row_seg_ptr,
                                                              const uint8_t *__restrict__ flag,
                                                              const long long *__restrict__ tiles,
                                                              hpf_segment *__restrict__ b_segs, int64_t b_cap,
                                                              int64_t *__restrict__ b_multi, int64_t m_cap,
                                                              int64_t *__restrict__ sizes) {
    const int ntile = gridDim.x, want = EPOCH ? (int)blockIdx.y : 1;
    tiles += (size_t)blockIdx.y * ntile * 2;
    b_segs += (size_t)blockIdx.y * b_cap;
    b_multi += (size_t)blockIdx.y * m_cap * 3;
    sizes += (size_t)blockIdx.y * 8;
    long long base[2], all[2];
    tiles_before<2>(tiles, blockIdx.x, base, all, ntile);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        sizes[0] = min((long long)b_cap, all[0]);
        sizes[1] = min((long long)m_cap, all[1]);
        if (all[0] > b_cap || all[1] > m_cap) sizes[7] = 1;      // overflow (cannot happen with the caller's bound)
    }
    const int64_t per = (nseg + ntile - 1) / ntile;
    const int64_t s0 = (int64_t)blockIdx.x * per, s1 = min(nseg, s0 + per);
    for (int64_t c0 = s0; c0 < s1; c0 += BLOCK) {                 // chunks of BLOCK segments, in order
        const int64_t s = c0 + threadIdx.x;
        long long v[2] = {0, 0};
        hpf_segment g;
        g.begin = 0;
        g.len = 0;
        g.row = 0;
        bool keep = false, opens = false;
        if (s < s1) {
            g = segs[s];
            keep = EPOCH ? flag[g.row] == want : flag[g.row] != 0;
            opens = keep && !(g.len & HPF_SEG_WHOLE_ROW) && (s == 0 || segs[s - 1].row != g.row);
            v[0] = keep;
            v[1] = opens;
        }
        long long e[2], tot[2];
        block_exclusive_scan<2>(v, e, tot);
        const long long pos = base[0] + e[0], mpos = base[1] + e[1];
        if (keep && pos < b_cap) b_segs[pos] = g;
        if (opens && mpos < m_cap) {
            b_multi[mpos * 3 + 0] = pos;
            b_multi[mpos * 3 + 1] = row_seg_ptr[g.row + 1] - row_seg_ptr[g.row];
            b_multi[mpos * 3 + 2] = g.row;
        }
        base[0] += tot[0];
        base[1] += tot[1];
    }
}

// ---- other side: the side's flat nonzero array (grouped by its rows, own-side ids ascending inside a row) FILTERED by the
// own side's flag.  Flat order is row order, so the compaction keeps exactly the order a stable sort by row would give.
// Entry e is kept iff flag[idx[e]]; its output position is P(e) = kept entries before e, read off a bitmask + a prefix per
// tile of 1024 entries: no pass over the nonzeros ever depends on a row's length.
constexpr int CHUNKS_PER_TILE = 16;       // 64-entry chunks per tile
#ifndef HPF_PREP_INFLIGHT
#define HPF_PREP_INFLIGHT 8               // chunks whose id (and count) loads a wave keeps in flight in the mask / write passes
#endif
constexpr int PIF = HPF_PREP_INFLIGHT;    // (4: the id stream of the mask pass ran at 1.1 TB/s -- latency-bound)

// (3a') the own side's flags as a bitset (a 1M-row side: 128 KB -- it fits the LDS of a CU, the byte table does not)
__global__ __launch_bounds__(BLOCK) void svi_flag_bits_kernel(const uint8_t *__restrict__ flag, int64_t nrows,
                                                              uint32_t *__restrict__ bits) {
    const int64_t nwords = (nrows + 31) / 32;
    const bool wide = (reinterpret_cast<uintptr_t>(flag) & 15) == 0;
    for (int64_t w = (int64_t)blockIdx.x * BLOCK + threadIdx.x; w < nwords; w += (int64_t)gridDim.x * BLOCK) {
        uint32_t v = 0;
        if (wide && w * 32 + 32 <= nrows) {       // 32 flags = two 16-byte loads (one byte load each: 0.1 ms per C5 batch)
            const uint4 q[2] = {reinterpret_cast<const uint4 *>(flag + w * 32)[0],
                                reinterpret_cast<const uint4 *>(flag + w * 32)[1]};
            const uint32_t d[8] = {q[0].x, q[0].y, q[0].z, q[0].w, q[1].x, q[1].y, q[1].z, q[1].w};
#pragma unroll
            for (int i = 0; i < 8; i++)
#pragma unroll
                for (int b = 0; b < 4; b++)
                    if ((d[i] >> (8 * b)) & 0xFFu) v |= 1u << (4 * i + b);
        } else {
            for (int b = 0; b < 32; b++) {
                const int64_t r = w * 32 + b;
                if (r < nrows && flag[r]) v |= 1u << b;
            }
        }
        bits[w] = v;
    }
}

// (3a) one wavefront per tile of 1024 entries: mask[c] = keep bits of entries [64c, 64c+64) for its 16 chunks, chunk_pre[c] =
// kept entries of the tile before chunk c, tile_cnt[t] = kept entries of the tile.  Four chunks' loads are in flight at once.
// The 48M flag look-ups are the cost: as byte gathers from memory they bound the kernel at ~4x the time the id stream
// takes; LDS_BITS: every workgroup (1024 threads, one per CU) first copies the bitset into LDS and looks the ids up there.
template <bool LDS_BITS>
__global__ __launch_bounds__(1024) void svi_oth_mask_kernel(const int32_t *__restrict__ idx, int64_t nnz,
                                                            const uint32_t *__restrict__ bits, int64_t nwords,
                                                            unsigned long long *__restrict__ mask,
                                                            uint16_t *__restrict__ chunk_pre,
                                                            int32_t *__restrict__ tile_cnt) {
    extern __shared__ uint32_t lbits[];
    const uint32_t *tab = bits;
    if constexpr (LDS_BITS) {
        for (int64_t w = threadIdx.x; w < nwords; w += blockDim.x) lbits[w] = bits[w];
        __syncthreads();
        tab = lbits;
    }
    const int lane = threadIdx.x & (WAVE - 1);
    const int wpb = blockDim.x / WAVE;
    const int64_t ntiles = (nnz + 1023) / 1024;
    const int64_t nwaves = (int64_t)gridDim.x * wpb;
    for (int64_t t = (int64_t)blockIdx.x * wpb + (threadIdx.x >> 6); t < ntiles; t += nwaves) {
        unsigned long long mine = 0;               // lane j < 16 ends up holding the mask of chunk j
#pragma unroll
        for (int j0 = 0; j0 < CHUNKS_PER_TILE; j0 += PIF) {
            int32_t id[PIF];
            bool in[PIF];
#pragma unroll
            for (int u = 0; u < PIF; u++) {
                const int64_t e = (t * CHUNKS_PER_TILE + j0 + u) * WAVE + lane;
                in[u] = e < nnz;
                id[u] = in[u] ? idx[e] : 0;
            }
#pragma unroll
            for (int u = 0; u < PIF; u++) {
                const unsigned long long m = __ballot(in[u] && ((tab[id[u] >> 5] >> (id[u] & 31)) & 1u));
                if (lane == j0 + u) mine = m;
            }
        }
        int pc = (lane < CHUNKS_PER_TILE) ? __popcll(mine) : 0, inc = pc;
#pragma unroll
        for (int d = 1; d < CHUNKS_PER_TILE; d <<= 1) {
            const int y = __shfl_up(inc, d);
            if (lane >= d) inc += y;
        }
        if (lane < CHUNKS_PER_TILE) {
            mask[t * CHUNKS_PER_TILE + lane] = mine;
            chunk_pre[t * CHUNKS_PER_TILE + lane] = (uint16_t)(inc - pc);
        }
        if (lane == CHUNKS_PER_TILE - 1) tile_cnt[t] = inc;
    }
}

// (3b) exclusive scan of the tile counts, two launches like the other scans: group sums, then the offsets
__global__ __launch_bounds__(BLOCK) void svi_tile_sums_kernel(const int32_t *__restrict__ tile_cnt, int64_t ntiles,
                                                              long long *__restrict__ groups) {
    const int64_t per = (ntiles + TILES - 1) / TILES;
    const int64_t t0 = (int64_t)blockIdx.x * per, t1 = min(ntiles, t0 + per);
    long long v[1] = {0};
    for (int64_t t = t0 + threadIdx.x; t < t1; t += BLOCK) v[0] += tile_cnt[t];
    long long e[1], tot[1];
    block_exclusive_scan<1>(v, e, tot);
    if (threadIdx.x == 0) groups[blockIdx.x] = tot[0];
}

__global__ __launch_bounds__(BLOCK) void svi_tile_offsets_kernel(const int32_t *__restrict__ tile_cnt, int64_t ntiles,
                                                                 const long long *__restrict__ groups,
                                                                 int64_t *__restrict__ tile_off) {
    long long base[1], all[1];
    tiles_before<1>(groups, blockIdx.x, base, all);
    const int64_t per = (ntiles + TILES - 1) / TILES;
    const int64_t t0 = (int64_t)blockIdx.x * per, t1 = min(ntiles, t0 + per);
    for (int64_t c0 = t0; c0 < t1; c0 += BLOCK) {
        const int64_t t = c0 + threadIdx.x;
        long long v[1] = {0};
        if (t < t1) v[0] = tile_cnt[t];
        long long e[1], tot[1];
        block_exclusive_scan<1>(v, e, tot);
        if (t < t1) tile_off[t] = base[0] + e[0];
        base[0] += tot[0];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) tile_off[ntiles] = all[0];
}

// kept entries before entry e
__device__ __forceinline__ long long kept_before(const unsigned long long *__restrict__ mask,
                                                 const uint16_t *__restrict__ chunk_pre,
                                                 const int64_t *__restrict__ tile_off, int64_t e) {
    const int64_t chunk = e >> 6;
    long long p = tile_off[chunk / CHUNKS_PER_TILE] + chunk_pre[chunk];
    const int b = (int)(e & 63);
    if (b) p += __popcll(mask[chunk] & ((1ull << b) - 1));
    return p;
}

// (3c) other side: per row, where its kept nonzeros start and how many there are; per tile of rows, the totals of
// {rows present, batch segments, split rows}
__global__ __launch_bounds__(BLOCK) void svi_oth_rows_kernel(const int64_t *__restrict__ indptr, int64_t nrows, int64_t nnz,
                                                             const unsigned long long *__restrict__ mask,
                                                             const uint16_t *__restrict__ chunk_pre,
                                                             const int64_t *__restrict__ tile_off, int cap,
                                                             int64_t *__restrict__ row_start, int32_t *__restrict__ row_cnt,
                                                             long long *__restrict__ tiles) {
    const int64_t per = (nrows + TILES - 1) / TILES;
    const int64_t r0 = (int64_t)blockIdx.x * per, r1 = min(nrows, r0 + per);
    long long v[3] = {0, 0, 0};
    for (int64_t r = r0 + threadIdx.x; r < r1; r += BLOCK) {
        const int64_t e0 = indptr[r], e1 = indptr[r + 1];
        long long p0 = 0, p1 = 0;
        if (e1 > e0) {
            p0 = kept_before(mask, chunk_pre, tile_off, e0);
            p1 = (e1 < nnz) ? kept_before(mask, chunk_pre, tile_off, e1) : tile_off[(nnz + 1023) / 1024];
        }
        const int c = (int)(p1 - p0);
        row_start[r] = p0;
        row_cnt[r] = c;
        if (c > 0) {
            const int ns = (c + cap - 1) / cap;
            v[0]++;
            v[1] += ns;
            v[2] += ns > 1;
        }
    }
    long long e[3], tot[3];
    block_exclusive_scan<3>(v, e, tot);
    if (threadIdx.x == 0)
        for (int c = 0; c < 3; c++) tiles[(size_t)blockIdx.x * 3 + c] = tot[c];
}

// (3d) other side: the scan over rows applied.  Every row gets its flag (present or not); a present row gets its batch
// segments {begin in o_idx / o_y, length, row} cut at `cap`, split rows their {first segment, segments, row} descriptor
__global__ __launch_bounds__(BLOCK) void svi_oth_layout_kernel(int64_t nrows, const int64_t *__restrict__ row_start,
                                                               const int32_t *__restrict__ row_cnt, int cap,
                                                               const long long *__restrict__ tiles,
                                                               const int64_t *__restrict__ tile_off, int64_t ntiles_e,
                                                               uint8_t *__restrict__ flag_oth,
                                                               hpf_segment *__restrict__ o_segs, int64_t o_segs_cap,
                                                               int64_t *__restrict__ o_multi, int64_t m_cap, int64_t o_cap,
                                                               int64_t *__restrict__ sizes) {
    long long base[3], all[3];
    tiles_before<3>(tiles, blockIdx.x, base, all);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const long long entries = tile_off[ntiles_e];
        sizes[5] = all[0];
        sizes[4] = min((long long)o_cap, entries);
        sizes[2] = min((long long)o_segs_cap, all[1]);
        sizes[3] = min((long long)m_cap, all[2]);
        if (entries > o_cap || all[1] > o_segs_cap || all[2] > m_cap) sizes[7] = 1;
    }
    const int64_t per = (nrows + TILES - 1) / TILES;
    const int64_t r0 = (int64_t)blockIdx.x * per, r1 = min(nrows, r0 + per);
    for (int64_t c0 = r0; c0 < r1; c0 += BLOCK) {
        const int64_t r = c0 + threadIdx.x;
        long long v[3] = {0, 0, 0};
        int c = 0, ns = 0;
        if (r < r1) {
            c = row_cnt[r];
            ns = (c + cap - 1) / cap;
            flag_oth[r] = c > 0 ? (ns > 1 ? 2 : 1) : 0;      // (a split row is told apart: hpf_hip_sweep_svi_f32)
            v[0] = c > 0;
            v[1] = ns;
            v[2] = ns > 1;
        }
        long long e[3], tot[3];
        block_exclusive_scan<3>(v, e, tot);
        if (c > 0) {
            const long long start = row_start[r], sg0 = base[1] + e[1], m0 = base[2] + e[2];
            for (int q = 0; q < ns; q++) {
                if (sg0 + q >= o_segs_cap) break;
                hpf_segment g;
                g.begin = start + (long long)q * cap;
                const int left = c - q * cap;
                g.len = (left < cap ? left : cap) | (ns == 1 ? HPF_SEG_WHOLE_ROW : 0);
                g.row = (int32_t)r;
                o_segs[sg0 + q] = g;
            }
            if (ns > 1 && m0 < m_cap) {
                o_multi[m0 * 3 + 0] = sg0;
                o_multi[m0 * 3 + 1] = ns;
                o_multi[m0 * 3 + 2] = r;
            }
        }
        for (int q = 0; q < 3; q++) base[q] += tot[q];
    }
}

// (3e) other side: the kept nonzeros written in order -- o_idx = the own side's row id, o_y = the count.  One wavefront
// per tile: the tile's 16 masks and prefixes arrive in one load each, PIF chunks' gathers are in flight at once
__global__ __launch_bounds__(BLOCK) void svi_oth_write_kernel(const int32_t *__restrict__ idx, const float *__restrict__ y,
                                                              int64_t nnz, const unsigned long long *__restrict__ mask,
                                                              const uint16_t *__restrict__ chunk_pre,
                                                              const int64_t *__restrict__ tile_off,
                                                              int32_t *__restrict__ o_idx, float *__restrict__ o_y,
                                                              int64_t o_cap) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int64_t ntiles = (nnz + 1023) / 1024;
    const int64_t nwaves = (int64_t)gridDim.x * WPB;
    for (int64_t t = (int64_t)blockIdx.x * WPB + (threadIdx.x >> 6); t < ntiles; t += nwaves) {
        const long long base = tile_off[t];
        if (tile_off[t + 1] == base) continue;     // nothing kept in this tile
        unsigned long long mine = 0;
        int pre = 0;
        if (lane < CHUNKS_PER_TILE) {
            mine = mask[t * CHUNKS_PER_TILE + lane];
            pre = chunk_pre[t * CHUNKS_PER_TILE + lane];
        }
#pragma unroll
        for (int j0 = 0; j0 < CHUNKS_PER_TILE; j0 += PIF) {
            int32_t id[PIF];
            float yy[PIF];
            long long pos[PIF];
            bool keep[PIF];
#pragma unroll
            for (int u = 0; u < PIF; u++) {
                const unsigned long long m = __shfl(mine, j0 + u);
                const int p = __shfl(pre, j0 + u);
                keep[u] = (m >> lane) & 1ull;
                pos[u] = base + p + __popcll(m & ((1ull << lane) - 1));
                const int64_t e = (t * CHUNKS_PER_TILE + j0 + u) * WAVE + lane;
                id[u] = keep[u] ? idx[e] : 0;
                yy[u] = keep[u] ? y[e] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < PIF; u++) {
                if (keep[u] && pos[u] < o_cap) {
                    o_idx[pos[u]] = id[u];
                    o_y[pos[u]] = yy[u];
                }
            }
        }
    }
}

// acc[row] = sum of the part[] rows of a split row, for the descriptors {first segment, segments, row} a batch
// preparation produced; *ndesc_dev descriptors (device-side count)
__global__ __launch_bounds__(BLOCK) void segsum_desc_kernel(const float *__restrict__ part,
                                                            const int64_t *__restrict__ desc,
                                                            const int64_t *__restrict__ ndesc_dev, int64_t ndesc_max,
                                                            float *__restrict__ acc, int ld) {
    const int64_t nd = min(ndesc_max, ndesc_dev[0]);
    const int64_t total = nd * ld;
    for (int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x; t < total; t += (int64_t)gridDim.x * BLOCK) {
        const int64_t d = t / ld;
        const int c = (int)(t - d * ld);
        int64_t sg = desc[d * 3 + 0];
        const int64_t s1 = sg + desc[d * 3 + 1];
        const int64_t row = desc[d * 3 + 2];
        float a = 0.f;
        for (; sg + 8 <= s1; sg += 8) {  // same fold order as row_finalize_kernel / segsum_kernel
            float p[8];
#pragma unroll
            for (int u = 0; u < 8; u++) p[u] = part[(size_t)(sg + u) * ld + c];
            a += ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
        }
        for (; sg < s1; sg++) a += part[(size_t)sg * ld + c];
        acc[(size_t)row * ld + c] = a;
    }
}


// ============================================================================================================
// One preparation per EPOCH (hpf_hip_svi_epoch_prepare).  The batches of an epoch partition the rows of its side, so the
// per-batch filter above -- a pass over all of the other side's ids for every batch, 11-16 of them per C5 epoch, 20 % of
// the epoch's kernel time -- is one labelled partition of those ids: key, count per {batch, segment}, one scan, one
// scatter.  What each batch then needs is a slice of the epoch's arrays; the per-batch host call disappears.
// ============================================================================================================
constexpr int ETILES = 256;           // scan tiles per batch of the row / segment compactions of an epoch

// (E0) the epoch's order -> batch of every row, the batch's row flags, zeroed accumulator rows of rows without nonzeros
__global__ __launch_bounds__(BLOCK) void svi_epoch_mark_kernel(const int64_t *__restrict__ order, int64_t nrows, int64_t per,
                                                               uint8_t *__restrict__ batch_of, uint8_t *__restrict__ flag,
                                                               const int64_t *__restrict__ indptr,
                                                               const int64_t *__restrict__ row_seg_ptr, float *__restrict__ acc,
                                                               int ld) {
    const int64_t stride = (int64_t)gridDim.x * BLOCK;
    for (int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x; t < nrows; t += stride) {
        const int64_t r = order[t], b = t / per;
        batch_of[r] = (uint8_t)b;
        flag[(size_t)b * nrows + r] = (row_seg_ptr[r + 1] - row_seg_ptr[r] == 1) ? 1 : 2;      // (see svi_mark_kernel)
        if (indptr[r + 1] == indptr[r]) {
            float4 *row = reinterpret_cast<float4 *>(acc + (size_t)r * ld);
            for (int c = 0; c < ld / 4; c++) row[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}

__device__ __forceinline__ long long readlane64(long long x, int l) {
    const int lo = __builtin_amdgcn_readlane((int)(x & 0xffffffffll), l);
    const int hi = __builtin_amdgcn_readlane((int)(x >> 32), l);
    return ((long long)hi << 32) | (unsigned int)lo;
}

// (E1) one wavefront per segment of the other side's global layout: key[e] = batch of the own-side row of nonzero e (a
// byte gather from a table the L2 holds), cnt[b][s] = nonzeros of segment s that belong to batch b.  Lane j counts
// bucket j (+64w): the distinct keys of a 64-entry chunk are walked with ballots, no LDS, no atomics.
template <int NBW>
__global__ __launch_bounds__(BLOCK) void svi_epoch_key_kernel(const hpf_segment *__restrict__ segs, int64_t nseg,
                                                              const int32_t *__restrict__ idx,
                                                              const uint8_t *__restrict__ batch_of,
                                                              uint8_t *__restrict__ key, int32_t *__restrict__ cnt, int nb) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int64_t nwaves = (int64_t)gridDim.x * WPB;
    for (int64_t s = (int64_t)blockIdx.x * WPB + (threadIdx.x >> 6); s < nseg; s += nwaves) {
        const hpf_segment g = segs[s];
        const int len = g.len & HPF_SEG_LEN_MASK;
        int c[NBW];
#pragma unroll
        for (int w = 0; w < NBW; w++) c[w] = 0;
        for (int o = 0; o < len; o += 4 * WAVE) {
            int32_t id[4];
            int kb[4];
            bool in[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                in[u] = o + u * WAVE + lane < len;
                id[u] = in[u] ? idx[g.begin + o + u * WAVE + lane] : 0;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) kb[u] = in[u] ? (int)batch_of[id[u]] : 0;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (in[u]) key[g.begin + o + u * WAVE + lane] = (uint8_t)kb[u];
                unsigned long long rem = __ballot(in[u]);
                while (rem) {
                    const int v = __builtin_amdgcn_readlane(kb[u], __ffsll((long long)rem) - 1);
                    const unsigned long long m = __ballot(in[u] && kb[u] == v);
                    const int n = __popcll(m);
#pragma unroll
                    for (int w = 0; w < NBW; w++)
                        if ((v >> 6) == w && lane == (v & 63)) c[w] += n;
                    rem &= ~m;
                }
            }
        }
#pragma unroll
        for (int w = 0; w < NBW; w++)
            if (w * WAVE + lane < nb) cnt[(int64_t)(w * WAVE + lane) * nseg + s] = c[w];
    }
}

// (E2) the scatter: pos[b][s] (the exclusive scan of cnt in {batch, segment} order) is where segment s's nonzeros of
// batch b go; inside the segment they keep their order (rank among the same key's lanes of the chunk + the running base)
template <int NBW>
__global__ __launch_bounds__(BLOCK) void svi_epoch_scatter_kernel(const hpf_segment *__restrict__ segs, int64_t nseg,
                                                                  const int32_t *__restrict__ idx, const float *__restrict__ y,
                                                                  const uint8_t *__restrict__ key,
                                                                  const int64_t *__restrict__ pos, int nb,
                                                                  int32_t *__restrict__ e_idx, float *__restrict__ e_y) {
    const int lane = threadIdx.x & (WAVE - 1);
    const unsigned long long below = (1ull << lane) - 1;
    const int64_t nwaves = (int64_t)gridDim.x * WPB;
    for (int64_t s = (int64_t)blockIdx.x * WPB + (threadIdx.x >> 6); s < nseg; s += nwaves) {
        const hpf_segment g = segs[s];
        const int len = g.len & HPF_SEG_LEN_MASK;
        long long base[NBW];
#pragma unroll
        for (int w = 0; w < NBW; w++) base[w] = (w * WAVE + lane < nb) ? pos[(int64_t)(w * WAVE + lane) * nseg + s] : 0;
        for (int o = 0; o < len; o += 4 * WAVE) {
            int32_t id[4];
            float yy[4];
            int kb[4];
            bool in[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int64_t e = g.begin + o + u * WAVE + lane;
                in[u] = o + u * WAVE + lane < len;
                id[u] = in[u] ? idx[e] : 0;
                yy[u] = in[u] ? y[e] : 0.f;
                kb[u] = in[u] ? (int)key[e] : 0;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                long long p = 0;
                unsigned long long rem = __ballot(in[u]);
                while (rem) {
                    const int v = __builtin_amdgcn_readlane(kb[u], __ffsll((long long)rem) - 1);
                    const unsigned long long m = __ballot(in[u] && kb[u] == v);
                    long long bv = 0;
#pragma unroll
                    for (int w = 0; w < NBW; w++)
                        if ((v >> 6) == w) bv = readlane64(base[w], v & 63);
                    if (in[u] && kb[u] == v) p = bv + __popcll(m & below);
#pragma unroll
                    for (int w = 0; w < NBW; w++)
                        if ((v >> 6) == w && lane == (v & 63)) base[w] += __popcll(m);
                    rem &= ~m;
                }
                if (in[u]) {
                    e_idx[p] = id[u];
                    e_y[p] = yy[u];
                }
            }
        }
    }
}

// (E3) per batch (blockIdx.y) and tile of the other side's rows: totals of {rows present, batch segments, split rows}.
// A row's nonzeros of batch b sit at pos[b][first segment of the row] .. pos[b][first segment of the next row]
__global__ __launch_bounds__(BLOCK) void svi_epoch_oth_count_kernel(const int64_t *__restrict__ row_seg_ptr, int64_t nrows,
                                                                    int64_t nseg, const int64_t *__restrict__ pos, int cap,
                                                                    long long *__restrict__ tiles) {
    const int ntile = gridDim.x;
    const int64_t *pb = pos + (int64_t)blockIdx.y * nseg;
    const int64_t per = (nrows + ntile - 1) / ntile;
    const int64_t r0 = (int64_t)blockIdx.x * per, r1 = min(nrows, r0 + per);
    long long v[3] = {0, 0, 0};
    for (int64_t r = r0 + threadIdx.x; r < r1; r += BLOCK) {
        const long long c = pb[row_seg_ptr[r + 1]] - pb[row_seg_ptr[r]];
        if (c > 0) {
            const long long ns = (c + cap - 1) / cap;
            v[0]++;
            v[1] += ns;
            v[2] += ns > 1;
        }
    }
    long long e[3], tot[3];
    block_exclusive_scan<3>(v, e, tot);
    if (threadIdx.x == 0)
        for (int c = 0; c < 3; c++) tiles[((size_t)blockIdx.y * ntile + blockIdx.x) * 3 + c] = tot[c];
}

// (E4) the layout of every batch's other side: flags of the rows present, segments cut at `cap`, split-row descriptors,
// sizes -- svi_oth_layout_kernel with a batch dimension
__global__ __launch_bounds__(BLOCK) void svi_epoch_oth_layout_kernel(const int64_t *__restrict__ row_seg_ptr, int64_t nrows,
                                                                     int64_t nseg, const int64_t *__restrict__ pos, int cap,
                                                                     const long long *__restrict__ tiles,
                                                                     uint8_t *__restrict__ flag_oth,
                                                                     hpf_segment *__restrict__ o_segs, int64_t o_segs_cap,
                                                                     int64_t *__restrict__ o_multi, int64_t m_cap,
                                                                     int64_t *__restrict__ sizes) {
    const int ntile = gridDim.x;
    const int64_t *pb = pos + (int64_t)blockIdx.y * nseg;
    tiles += (size_t)blockIdx.y * ntile * 3;
    flag_oth += (size_t)blockIdx.y * nrows;
    o_segs += (size_t)blockIdx.y * o_segs_cap;
    o_multi += (size_t)blockIdx.y * m_cap * 3;
    sizes += (size_t)blockIdx.y * 8;
    long long base[3], all[3];
    tiles_before<3>(tiles, blockIdx.x, base, all, ntile);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        sizes[5] = all[0];
        sizes[4] = pb[nseg] - pb[0];
        sizes[2] = min((long long)o_segs_cap, all[1]);
        sizes[3] = min((long long)m_cap, all[2]);
        if (all[1] > o_segs_cap || all[2] > m_cap) sizes[7] = 1;
    }
    const int64_t per = (nrows + ntile - 1) / ntile;
    const int64_t r0 = (int64_t)blockIdx.x * per, r1 = min(nrows, r0 + per);
    for (int64_t c0 = r0; c0 < r1; c0 += BLOCK) {
        const int64_t r = c0 + threadIdx.x;
        long long v[3] = {0, 0, 0};
        long long c = 0, ns = 0, start = 0;
        if (r < r1) {
            start = pb[row_seg_ptr[r]];
            c = pb[row_seg_ptr[r + 1]] - start;
            ns = (c + cap - 1) / cap;
            flag_oth[r] = c > 0 ? (ns > 1 ? 2 : 1) : 0;      // (a split row is told apart: hpf_hip_sweep_svi_f32)
            v[0] = c > 0;
            v[1] = ns;
            v[2] = ns > 1;
        }
        long long e[3], tot[3];
        block_exclusive_scan<3>(v, e, tot);
        if (c > 0) {
            const long long sg0 = base[1] + e[1], m0 = base[2] + e[2];
            for (long long q = 0; q < ns; q++) {
                if (sg0 + q >= o_segs_cap) break;
                hpf_segment g;
                g.begin = start + q * cap;
                const long long left = c - q * cap;
                g.len = (int32_t)(left < cap ? left : cap) | (ns == 1 ? HPF_SEG_WHOLE_ROW : 0);
                g.row = (int32_t)r;
                o_segs[sg0 + q] = g;
            }
            if (ns > 1 && m0 < m_cap) {
                o_multi[m0 * 3 + 0] = sg0;
                o_multi[m0 * 3 + 1] = ns;
                o_multi[m0 * 3 + 2] = r;
            }
        }
        for (int q = 0; q < 3; q++) base[q] += tot[q];
    }
}

// ============================================================================================================
// One batch handed over as COO triplets (partial_fit, PXI:423-434: Y_batch, ix_u_batch, ix_i_batch in the caller's order):
// there is no resident CSR to slice, so each grouping is a stable sort of the batch by that side's row id (the caller's:
// a device radix sort) + the layout below -- heads and tails of the sorted runs, then the other-side layout of the batch
// preparations above (svi_oth_layout_kernel: flags, segments cut at `cap`, split-row descriptors, sizes).
// ============================================================================================================
// (C0) reference ids (size_t, arriving as int64: an id >= 2^63 is negative) -> int32 row ids, with the range check the
// reference does not make (bounds checks off, PXI:547-550: an id past the table is undefined behaviour there)
__global__ __launch_bounds__(BLOCK) void svi_coo_narrow_kernel(const int64_t *__restrict__ ids, int64_t n, int64_t limit,
                                                               int32_t *__restrict__ out, int64_t *__restrict__ err) {
    bool bad = false;
    for (int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x; t < n; t += (int64_t)gridDim.x * BLOCK) {
        const int64_t v = ids[t];
        const bool ok = v >= 0 && v < limit;
        bad |= !ok;
        out[t] = ok ? (int32_t)v : 0;
    }
    if (__ballot(bad) != 0 && (threadIdx.x & (WAVE - 1)) == 0) err[0] = 1;      // (every writer writes the same value)
}

// (C1) sorted keys -> where each row's run starts and ends (row_cnt holds the END until C2 turns it into the count; it is
// zero-filled beforehand, and a run's end is >= 1: zero = the row is absent)
__global__ __launch_bounds__(BLOCK) void svi_coo_heads_kernel(const int32_t *__restrict__ key, int64_t n,
                                                              int64_t *__restrict__ row_start, int32_t *__restrict__ row_end) {
    for (int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x; t < n; t += (int64_t)gridDim.x * BLOCK) {
        const int32_t r = key[t];
        if (t == 0 || key[t - 1] != r) row_start[r] = t;
        if (t == n - 1 || key[t + 1] != r) row_end[r] = (int32_t)(t + 1);
    }
}

// (C2) per row its count; per tile of rows the totals of {rows present, batch segments, split rows} (svi_oth_rows_kernel's
// outputs from the runs instead of from a keep-mask)
__global__ __launch_bounds__(BLOCK) void svi_coo_rows_kernel(int64_t nrows, const int64_t *__restrict__ row_start,
                                                             int32_t *__restrict__ row_cnt, int cap,
                                                             long long *__restrict__ tiles, int64_t n,
                                                             int64_t *__restrict__ entries) {
    if (blockIdx.x == 0 && threadIdx.x == 0) entries[0] = n;      // (where the layout kernel reads the number of nonzeros)
    const int64_t per = (nrows + TILES - 1) / TILES;
    const int64_t r0 = (int64_t)blockIdx.x * per, r1 = min(nrows, r0 + per);
    long long v[3] = {0, 0, 0};
    for (int64_t r = r0 + threadIdx.x; r < r1; r += BLOCK) {
        const int end = row_cnt[r];
        const int c = end > 0 ? (int)(end - row_start[r]) : 0;
        row_cnt[r] = c;
        if (c > 0) {
            const int ns = (c + cap - 1) / cap;
            v[0]++;
            v[1] += ns;
            v[2] += ns > 1;
        }
    }
    long long e[3], tot[3];
    block_exclusive_scan<3>(v, e, tot);
    if (threadIdx.x == 0)
        for (int c = 0; c < 3; c++) tiles[(size_t)blockIdx.x * 3 + c] = tot[c];
}

inline int last_error() { return (int)hipGetLastError(); }

}  // namespace

extern "C" {

int64_t hpf_hip_svi_prep_scratch_words(void) { return (int64_t)TILES * 8; }   // int64 words of `tiles`
int64_t hpf_hip_svi_batch_sizeof(void) { return (int64_t)sizeof(hpf_svi_batch); }

int hpf_hip_svi_batch_prepare(const hpf_svi_batch *b, void *stream) {
    if (!b || !b->own_segs || !b->own_row_seg_ptr || !b->own_indptr || !b->oth_idx || !b->oth_y || !b->ids || !b->flag_own || !b->flag_oth || !b->acc_own || !b->b_segs || !b->b_multi ||
        !b->o_idx || !b->o_y || !b->o_segs || !b->o_multi || !b->sizes || !b->mask || !b->chunk_pre || !b->flag_bits || !b->tile_cnt || !b->tile_off ||
        !b->row_cnt || !b->row_start || !b->tiles || !b->oth_indptr || b->oth_nnz < 0 || b->own_nseg < 0 || b->nids < 0 || b->nprev < 0 || (b->nprev > 0 && !b->prev_ids) ||
        b->seg_cap <= 0 || b->seg_cap > HPF_SEG_LEN_MASK || b->ld <= 0 || b->own_nrows <= 0 || b->oth_nrows <= 0)
        return HPF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    auto grid_for = [](int64_t n, int per) { int64_t g = (n + per - 1) / per; return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); };
    // sizes[0..6] are this batch's; sizes[7], the overflow flag, is STICKY: the driver reads it once after all epochs, and
    // a truncated segment list of an earlier batch must not be forgotten (the workspace zeroes it at allocation)
    hipError_t e = hipMemsetAsync(b->sizes, 0, 7 * sizeof(int64_t), st);
    if (e != hipSuccess) return (int)e;
    if (b->nprev > 0)
        hipLaunchKernelGGL(svi_mark_kernel, dim3(grid_for(b->nprev, BLOCK)), dim3(BLOCK), 0, st, b->prev_ids, b->nprev,
                           b->ids, b->nids, b->flag_own, b->own_indptr, b->own_row_seg_ptr, b->acc_own, b->ld, 0);
    if (b->nids > 0)
        hipLaunchKernelGGL(svi_mark_kernel, dim3(grid_for(b->nids, BLOCK)), dim3(BLOCK), 0, st, b->prev_ids, b->nprev,
                           b->ids, b->nids, b->flag_own, b->own_indptr, b->own_row_seg_ptr, b->acc_own, b->ld, 1);
    long long *tiles_own = (long long *)b->tiles, *tiles_oth = (long long *)b->tiles + TILES * 4;   // [2 | 3 | 1 per tile]
    hipLaunchKernelGGL(svi_own_count_kernel<false>, dim3(TILES), dim3(BLOCK), 0, st, b->own_segs, b->own_nseg, b->flag_own,
                       tiles_own);
    hipLaunchKernelGGL(svi_own_write_kernel<false>, dim3(TILES), dim3(BLOCK), 0, st, b->own_segs, b->own_nseg,
                       b->own_row_seg_ptr, b->flag_own, (const long long *)tiles_own, b->b_segs, b->b_segs_cap, b->b_multi,
                       b->multi_cap, b->sizes);
    const int64_t nnz = b->oth_nnz;
    const int64_t ntiles_e = (nnz + 1023) / 1024;
    unsigned long long *mask = (unsigned long long *)b->mask;
    long long *groups = (long long *)b->tiles + TILES * 4 + TILES * 3;
    uint16_t *chunk_pre = (uint16_t *)b->chunk_pre;
    const int tgrid = grid_for(ntiles_e, WPB);
    if (nnz > 0) {
        const int64_t nwords = (b->own_nrows + 31) / 32;
        hipLaunchKernelGGL(svi_flag_bits_kernel, dim3(grid_for(nwords, BLOCK)), dim3(BLOCK), 0, st,
                           (const uint8_t *)b->flag_own, b->own_nrows, b->flag_bits);
        const size_t lds = (size_t)nwords * sizeof(uint32_t);
        static int cus = 0;
        static bool big_lds = false;
        if (cus == 0) {
            int dev = 0;
            hipDeviceProp_t prop;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
                cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
            else
                cus = 256;
            big_lds = hipFuncSetAttribute(reinterpret_cast<const void *>(svi_oth_mask_kernel<true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512) == hipSuccess;
        }
        if (lds <= 64 * 1024 - 512 || (big_lds && lds <= 160 * 1024 - 512))
            hipLaunchKernelGGL((svi_oth_mask_kernel<true>), dim3(cus), dim3(1024), lds, st, b->oth_idx, nnz,
                               (const uint32_t *)b->flag_bits, nwords, mask, chunk_pre, b->tile_cnt);
        else    // a side too large for the LDS: the bitset is still 8x smaller than the byte table in the caches
            hipLaunchKernelGGL((svi_oth_mask_kernel<false>), dim3(grid_for(ntiles_e, 16)), dim3(1024), 0, st, b->oth_idx,
                               nnz, (const uint32_t *)b->flag_bits, nwords, mask, chunk_pre, b->tile_cnt);
    }
    hipLaunchKernelGGL(svi_tile_sums_kernel, dim3(TILES), dim3(BLOCK), 0, st, (const int32_t *)b->tile_cnt, ntiles_e,
                       groups);
    hipLaunchKernelGGL(svi_tile_offsets_kernel, dim3(TILES), dim3(BLOCK), 0, st, (const int32_t *)b->tile_cnt, ntiles_e,
                       (const long long *)groups, b->tile_off);
    hipLaunchKernelGGL(svi_oth_rows_kernel, dim3(TILES), dim3(BLOCK), 0, st, b->oth_indptr, b->oth_nrows, nnz,
                       (const unsigned long long *)mask, (const uint16_t *)chunk_pre, (const int64_t *)b->tile_off,
                       b->seg_cap, b->row_start, b->row_cnt, tiles_oth);
    hipLaunchKernelGGL(svi_oth_layout_kernel, dim3(TILES), dim3(BLOCK), 0, st, b->oth_nrows,
                       (const int64_t *)b->row_start, (const int32_t *)b->row_cnt, b->seg_cap,
                       (const long long *)tiles_oth, (const int64_t *)b->tile_off, ntiles_e, b->flag_oth, b->o_segs,
                       b->o_segs_cap, b->o_multi, b->multi_cap, b->o_cap, b->sizes);
    if (nnz > 0)
        hipLaunchKernelGGL(svi_oth_write_kernel, dim3(tgrid), dim3(BLOCK), 0, st, b->oth_idx, b->oth_y, nnz,
                           (const unsigned long long *)mask, (const uint16_t *)chunk_pre, (const int64_t *)b->tile_off,
                           b->o_idx, b->o_y, b->o_cap);
    return last_error();
}

int64_t hpf_hip_svi_epoch_sizeof(void) { return (int64_t)sizeof(hpf_svi_epoch); }
int64_t hpf_hip_svi_epoch_scratch_words(int nb) { return (int64_t)(nb < 1 ? 1 : nb) * ETILES * 5 + TILES; }

int hpf_hip_svi_epoch_prepare(const hpf_svi_epoch *b, void *stream) {
    // (a side without any nonzero has no segment list and no nonzero arrays: null pointers with zero counts are fine)
    if (!b || (!b->own_segs && b->own_nseg != 0) || !b->own_row_seg_ptr || !b->own_indptr || (!b->oth_segs && b->oth_nseg != 0) ||
        !b->oth_row_seg_ptr || ((!b->oth_idx || !b->oth_y) && b->oth_nnz != 0) || !b->order || !b->acc_own || !b->batch_of || !b->flag_own || !b->flag_oth || !b->b_segs || !b->b_multi ||
        !b->e_idx || !b->e_y || !b->o_segs || !b->o_multi || !b->sizes || !b->key || !b->seg_cnt || !b->seg_pos || !b->tiles ||
        b->own_nrows <= 0 || b->oth_nrows <= 0 || b->own_nseg < 0 || b->oth_nseg < 0 || b->oth_nnz < 0 || b->per <= 0 ||
        b->nb < 1 || b->nb > 255 || (int64_t)b->nb * b->per < b->own_nrows || b->seg_cap <= 0 ||
        b->seg_cap > HPF_SEG_LEN_MASK || b->ld <= 0)
        return HPF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    auto grid_for = [](int64_t n, int per) { int64_t g = (n + per - 1) / per; return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g)); };
    const int nb = b->nb;
    hipError_t e = hipMemsetAsync(b->flag_own, 0, (size_t)nb * b->own_nrows, st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(svi_epoch_mark_kernel, dim3(grid_for(b->own_nrows, BLOCK)), dim3(BLOCK), 0, st, b->order, b->own_nrows,
                       b->per, b->batch_of, b->flag_own, b->own_indptr, b->own_row_seg_ptr, b->acc_own, b->ld);
    long long *tiles_own = (long long *)b->tiles, *tiles_oth = tiles_own + (size_t)nb * ETILES * 2,
              *groups = tiles_oth + (size_t)nb * ETILES * 3;
    hipLaunchKernelGGL(svi_own_count_kernel<true>, dim3(ETILES, nb), dim3(BLOCK), 0, st, b->own_segs, b->own_nseg,
                       (const uint8_t *)b->batch_of, tiles_own);
    hipLaunchKernelGGL(svi_own_write_kernel<true>, dim3(ETILES, nb), dim3(BLOCK), 0, st, b->own_segs, b->own_nseg,
                       b->own_row_seg_ptr, (const uint8_t *)b->batch_of, (const long long *)tiles_own, b->b_segs,
                       b->b_segs_cap, b->b_multi, b->multi_cap, b->sizes);
    const int64_t nseg = b->oth_nseg, n_pairs = (int64_t)nb * nseg;
    const int sgrid = grid_for(nseg, WPB);
    if (nseg > 0) {
        if (nb <= 64)
            hipLaunchKernelGGL((svi_epoch_key_kernel<1>), dim3(sgrid), dim3(BLOCK), 0, st, b->oth_segs, nseg, b->oth_idx,
                               (const uint8_t *)b->batch_of, b->key, b->seg_cnt, nb);
        else
            hipLaunchKernelGGL((svi_epoch_key_kernel<4>), dim3(sgrid), dim3(BLOCK), 0, st, b->oth_segs, nseg, b->oth_idx,
                               (const uint8_t *)b->batch_of, b->key, b->seg_cnt, nb);
    }
    hipLaunchKernelGGL(svi_tile_sums_kernel, dim3(TILES), dim3(BLOCK), 0, st, (const int32_t *)b->seg_cnt, n_pairs, groups);
    hipLaunchKernelGGL(svi_tile_offsets_kernel, dim3(TILES), dim3(BLOCK), 0, st, (const int32_t *)b->seg_cnt, n_pairs,
                       (const long long *)groups, b->seg_pos);
    if (nseg > 0) {
        if (nb <= 64)
            hipLaunchKernelGGL((svi_epoch_scatter_kernel<1>), dim3(sgrid), dim3(BLOCK), 0, st, b->oth_segs, nseg, b->oth_idx,
                               b->oth_y, (const uint8_t *)b->key, (const int64_t *)b->seg_pos, nb, b->e_idx, b->e_y);
        else
            hipLaunchKernelGGL((svi_epoch_scatter_kernel<4>), dim3(sgrid), dim3(BLOCK), 0, st, b->oth_segs, nseg, b->oth_idx,
                               b->oth_y, (const uint8_t *)b->key, (const int64_t *)b->seg_pos, nb, b->e_idx, b->e_y);
    }
    hipLaunchKernelGGL(svi_epoch_oth_count_kernel, dim3(ETILES, nb), dim3(BLOCK), 0, st, b->oth_row_seg_ptr, b->oth_nrows,
                       nseg, (const int64_t *)b->seg_pos, b->seg_cap, tiles_oth);
    hipLaunchKernelGGL(svi_epoch_oth_layout_kernel, dim3(ETILES, nb), dim3(BLOCK), 0, st, b->oth_row_seg_ptr, b->oth_nrows,
                       nseg, (const int64_t *)b->seg_pos, b->seg_cap, (const long long *)tiles_oth, b->flag_oth, b->o_segs,
                       b->o_segs_cap, b->o_multi, b->multi_cap, b->sizes);
    return last_error();
}

int hpf_hip_svi_coo_narrow(const int64_t *ids, int64_t n, int64_t limit, int32_t *out, int64_t *err, void *stream) {
    if (n == 0) return 0;
    if (!ids || !out || !err || n < 0 || limit <= 0 || limit > 0x7fffffffll) return HPF_EINVAL;
    int64_t g = (n + BLOCK - 1) / BLOCK;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(svi_coo_narrow_kernel, dim3((unsigned)g), dim3(BLOCK), 0, (hipStream_t)stream, ids, n, limit, out, err);
    return last_error();
}

int64_t hpf_hip_svi_coo_sizeof(void) { return (int64_t)sizeof(hpf_svi_coo); }

int hpf_hip_svi_coo_prepare(const hpf_svi_coo *b, void *stream) {
    if (!b || b->n < 0 || b->n > 0x7fffffffll || (b->n > 0 && !b->key) || b->nrows <= 0 || b->seg_cap <= 0 ||
        b->seg_cap > HPF_SEG_LEN_MASK || !b->flag || !b->row_start || !b->row_cnt || !b->segs || !b->multi || !b->sizes ||
        !b->tiles || b->segs_cap <= 0 || b->multi_cap <= 0)
        return HPF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    auto grid_for = [](int64_t n, int per) { int64_t g = (n + per - 1) / per; return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); };
    hipError_t e = hipMemsetAsync(b->sizes, 0, 7 * sizeof(int64_t), st);      // (sizes[7], the error word, is sticky)
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(b->row_cnt, 0, (size_t)b->nrows * sizeof(int32_t), st);
    if (e != hipSuccess) return (int)e;
    long long *tiles = (long long *)b->tiles;
    int64_t *entries = b->tiles + (size_t)TILES * 3;          // one word: the number of nonzeros (the layout kernel's tile_off[0])
    if (b->n > 0)
        hipLaunchKernelGGL(svi_coo_heads_kernel, dim3(grid_for(b->n, BLOCK)), dim3(BLOCK), 0, st, b->key, b->n, b->row_start,
                           b->row_cnt);
    hipLaunchKernelGGL(svi_coo_rows_kernel, dim3(TILES), dim3(BLOCK), 0, st, b->nrows, (const int64_t *)b->row_start,
                       b->row_cnt, b->seg_cap, tiles, b->n, entries);
    hipLaunchKernelGGL(svi_oth_layout_kernel, dim3(TILES), dim3(BLOCK), 0, st, b->nrows, (const int64_t *)b->row_start,
                       (const int32_t *)b->row_cnt, b->seg_cap, (const long long *)tiles, (const int64_t *)entries, (int64_t)0,
                       b->flag, b->segs, b->segs_cap, b->multi, b->multi_cap, b->n > 0 ? b->n : 1, b->sizes);
    return last_error();
}

int hpf_hip_segsum_desc_f32(const float *part, const int64_t *desc, const int64_t *ndesc_dev, int64_t ndesc_max,
                            float *acc, int ld, void *stream) {
    if (ndesc_max == 0) return 0;
    if (!part || !desc || !ndesc_dev || !acc || ndesc_max < 0 || ld < 32) return HPF_EINVAL;
    int64_t g = (ndesc_max * ld + BLOCK - 1) / BLOCK;
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(segsum_desc_kernel, dim3((unsigned)g), dim3(BLOCK), 0, (hipStream_t)stream, part, desc, ndesc_dev,
                       ndesc_max, acc, ld);
    return last_error();
}

}  // extern "C"
