// hpf_mt19937.hip -- the MT19937 stream of initialize_parameters (cython_loops.pxi:127-138, "PXI") on the device.
//
// The reference draws its four initial tables with numpy's Generator(MT19937(seed)).random(dtype=float32): one 32-bit
// output per value.  Bit-identical draws need the generator's state words x[n+624] = x[n+397] ^ A(x[n], x[n+1]) in
// stream order -- a recurrence that only parallelises over 227 words, so ONE workgroup can walk it at 0.27 us per
// 624-word block: 60 ms for C3's 138M words, 255 ms for C5's 552M (profiles/r02_svi_c5_rocprofv3.txt).
//
// Here the stream is cut into chunks of 624 * 2^p words and every chunk gets its own workgroup.  The state a chunk starts
// from is obtained by JUMP-AHEAD: MT19937's state transition T is linear over GF(2) with the primitive characteristic
// polynomial phi (degree 19937, 135 terms), so T^n = g(T) with g(x) = x^n mod phi, and because every bit of every stream
// word z_j (j counted from a regenerated state) is a linear functional of the state,
//     z_{n+m} = XOR over the set bits i of g of z_{i+m}         (m = 0 .. 623: the state n words ahead)
// -- a 19937-term XOR over a window of 19937 + 624 consecutive stream words, embarrassingly parallel.  Only the
// polynomials x^(624 * 2^q) mod phi are needed (successive squarings of the monomial x^624, host side, 2 ms once per
// process): a binary tree over the chunk indices reaches every chunk start in log2(#chunks) levels.
//
// Kernels: mt19937_prepare_kernel (head of the current state, first regeneration), mt19937_jump_kernel (one tree level),
// mt19937_words_kernel (the recurrence, one workgroup per chunk; alone it is the round-1/2 serial path, still used for
// short draws).  Tempering and the float conversion stay in hpf_hip.hip (uniform_rows_kernel).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "hpf_hip.h"

namespace {

constexpr int MT_N = 624;
constexpr int MT_DEG = 19937;
constexpr int MT_WIN = MT_DEG + MT_N;             // stream words a jump reads (indices 0 .. 19936 + 623)
constexpr int MT_STATE_STRIDE = 640;              // scratch words per chunk state: 624 key words, pos, padding
constexpr int MT_MAX_Q = 44;                      // polynomials x^(624 * 2^q), q < MT_MAX_Q (2^53 words: beyond any use)
constexpr long long MT_PAR_MIN = 1ll << 22;       // shorter draws: the single-workgroup walk is faster than the jump tree
constexpr int MT_META = 16;                       // scratch header words ([0] = words taken from the caller's state)

// exponents of the characteristic polynomial of MT19937's state transition, below its degree (phi = x^19937 + ...):
// a constant of the generator like its tempering masks.  tests/test_host_logic.py re-derives the jump from it against
// numpy's own stepping; a wrong entry cannot go unnoticed.
const int kPhiLow[134] = {
    0,     1189,  1416,  1585,  1643,  1870,  2493,  2773,  3000,  3227,  3454,  3681,  3908,  4135,  4362,  4753,
    5661,  6337,  6569,  7129,  7477,  7525,  7583,  7752,  7979,  8206,  9505,  9901,  9969,  10128, 10693, 10761,
    10920, 11089, 11147, 11157, 11215, 11321, 11374, 11384, 11485, 11611, 11712, 11717, 11838, 11881, 11944, 11997,
    12277, 12335, 12393, 12504, 12509, 12620, 12673, 12731, 12736, 12789, 12905, 12958, 12963, 13137, 13185, 13190,
    13243, 13301, 13412, 13528, 13533, 13639, 13697, 13760, 13813, 13866, 14093, 14151, 14209, 14320, 14325, 14436,
    14547, 14552, 14605, 14721, 14774, 14779, 14953, 15001, 15006, 15059, 15117, 15228, 15344, 15349, 15455, 15513,
    15576, 15629, 15682, 15909, 15967, 16025, 16136, 16141, 16252, 16363, 16368, 16421, 16537, 16590, 16595, 16817,
    16822, 16875, 16933, 17044, 17160, 17271, 17329, 17445, 17498, 17725, 17783, 17841, 17952, 18068, 18179, 18237,
    18406, 18633, 18691, 18860, 19087, 19314};

// g_q = x^(624 * 2^q) mod phi as 624 uint32 words (bit i of word i/32 = coefficient of x^i), q = 0 .. MT_MAX_Q-1
std::vector<uint32_t> g_polys;
std::once_flag g_polys_once;

inline uint64_t spread_bits(uint64_t x) {   // bit b of the low 32 -> bit 2b (squaring in GF(2)[x])
    x = (x | (x << 16)) & 0x0000ffff0000ffffull;
    x = (x | (x << 8)) & 0x00ff00ff00ff00ffull;
    x = (x | (x << 4)) & 0x0f0f0f0f0f0f0f0full;
    x = (x | (x << 2)) & 0x3333333333333333ull;
    x = (x | (x << 1)) & 0x5555555555555555ull;
    return x;
}

void build_polys() {
    constexpr int PW = 312;                      // 64-bit words of a reduced polynomial (degree < 19968)
    std::vector<uint64_t> cur(PW, 0), t(2 * PW + 2);
    cur[MT_N >> 6] = 1ull << (MT_N & 63);        // x^624
    g_polys.assign((size_t)MT_MAX_Q * MT_N, 0);
    for (int q = 0; q < MT_MAX_Q; q++) {
        memcpy(&g_polys[(size_t)q * MT_N], cur.data(), PW * sizeof(uint64_t));   // (little endian: uint64 = 2 uint32)
        std::fill(t.begin(), t.end(), 0);
        for (int i = 0; i < PW; i++) {
            t[2 * i] = spread_bits(cur[i] & 0xffffffffull);
            t[2 * i + 1] = spread_bits(cur[i] >> 32);
        }
        // reduce mod phi, 64 coefficients at a time from the top: x^p = sum_e x^(p - 19937 + e) over phi's low terms; the
        // highest of them is 19314 = 19937 - 623, so a chunk's image lies entirely below the chunk
        for (int w = 2 * PW - 1; w >= PW; --w) {
            const uint64_t c = t[w];
            if (!c) continue;
            t[w] = 0;
            for (int e : kPhiLow) {
                const int base = 64 * w - MT_DEG + e;
                t[base >> 6] ^= c << (base & 63);
                if (base & 63) t[(base >> 6) + 1] ^= c >> (64 - (base & 63));
            }
        }
        const int sh = MT_DEG & 63;              // coefficients 19937 .. 19967 of the last kept word
        const uint64_t c = t[PW - 1] >> sh;
        if (c) {
            t[PW - 1] &= (1ull << sh) - 1;
            for (int e : kPhiLow) {
                t[e >> 6] ^= c << (e & 63);
                if (e & 63) t[(e >> 6) + 1] ^= c >> (64 - (e & 63));
            }
        }
        memcpy(cur.data(), t.data(), PW * sizeof(uint64_t));
    }
}

__device__ __forceinline__ uint32_t mt_twist(uint32_t hi, uint32_t lo) {
    const uint32_t y = (hi & 0x80000000u) | (lo & 0x7fffffffu);
    return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

// The recurrence, one workgroup per chunk.  Chunk c starts from the state states[c * state_stride ..] (624 key words +
// numpy's `pos`), produces words [c * chunk_words, min((c+1) * chunk_words, total)) of raw and -- the chunk holding the
// stream's last word only -- leaves the stream's new state in final_state.  total_dev (optional): the word count is
// total - *total_dev (the words a prepare kernel already took from the caller's state).  Thread t of 227 produces words
// t, 227+t and 454+t of every 624-word state -- each depends on the OLD state and on the word the SAME thread produced
// just before (x[i-227]) -- and stores them, untempered; the state is double-buffered in LDS, one barrier per 624 words.
__global__ __launch_bounds__(256) void mt19937_words_kernel(const uint32_t *states, long long state_stride,
                                                            uint32_t *__restrict__ raw, long long chunk_words,
                                                            long long total, const uint32_t *__restrict__ total_dev,
                                                            uint32_t *final_state) {   // (may be `states` itself)
    __shared__ uint32_t buf[2][624];
    const int t = threadIdx.x;
    if (total_dev) {
        total -= (long long)total_dev[0];
        raw += total_dev[0];
    }
    const long long first = (long long)blockIdx.x * chunk_words;
    if (first >= total) return;
    const long long n = (total - first < chunk_words) ? total - first : chunk_words;
    const bool last_chunk = first + n >= total;
    const uint32_t *state = states + (long long)blockIdx.x * state_stride;
    raw += first;
    for (int i = t; i < 624; i += 256) buf[0][i] = state[i];
    const int pos = (int)state[624];
    __syncthreads();
    for (int i = t; i < 624; i += 256) {             // words left in the current state
        const long long j = (long long)i - pos;
        if (j >= 0 && j < n) raw[j] = buf[0][i];
    }
    const long long rest = n - (624 - pos);
    if (rest <= 0) {
        if (last_chunk && final_state) {
            for (int i = t; i < 624; i += 256) final_state[i] = buf[0][i];
            if (t == 0) final_state[624] = (uint32_t)(pos + n);
        }
        return;
    }
    const long long nblk = (rest + 623) / 624;
    uint32_t *dst = raw + (624 - pos) + t;           // this thread's word of the current block, first of three
    long long left = rest - t;                       // words of the stream from dst on
    const int tc = t < 170 ? t : 0;                  // (all ten LDS reads of a step are issued together, unconditionally)
    for (long long b = 0; b < nblk; ++b) {
        const uint32_t *old = buf[b & 1];
        uint32_t *nw = buf[(b & 1) ^ 1];
        if (t < 227) {
            const uint32_t a0 = old[t], a1 = old[t + 1], am = old[t + 397];
            const uint32_t b0 = old[227 + t], b1 = old[228 + t];
            const uint32_t c0 = old[454 + tc], c1 = old[455 + (t < 169 ? t : 0)];
            const uint32_t w0 = old[0], w1 = old[1], wm = old[397];
            asm volatile("" ::"v"(w0), "v"(w1), "v"(wm));      // (keeps these reads out of the t == 169 branch: one LDS latency)
            const uint32_t nA = am ^ mt_twist(a0, a1);
            const uint32_t nB = nA ^ mt_twist(b0, b1);
            // word 623 wraps around to the NEW word 0 (recomputed here instead of waiting for thread 0)
            const uint32_t nC = nB ^ mt_twist(c0, t < 169 ? c1 : (wm ^ mt_twist(w0, w1)));
            nw[t] = nA;
            nw[227 + t] = nB;
            if (t < 170) nw[454 + t] = nC;
            if (left > 0) dst[0] = nA;
            if (left > 227) dst[227] = nB;
            if (t < 170 && left > 454) dst[454] = nC;
            dst += 624;
            left -= 624;
        }
        __syncthreads();
    }
    if (last_chunk && final_state) {
        const uint32_t *fin = buf[nblk & 1];
        for (int i = t; i < 624; i += 256) final_state[i] = fin[i];
        if (t == 0) final_state[624] = (uint32_t)(rest - 624 * (nblk - 1));
    }
}

// Parallel path, step 1: the words still unread in the caller's state (numpy's key from `pos` on) go to raw[0 .. h);
// the NEXT state (one regeneration: all 624 words full, which the jump formula needs) becomes chunk state 0; meta[0] = h.
__global__ __launch_bounds__(256) void mt19937_prepare_kernel(const uint32_t *__restrict__ state, uint32_t *__restrict__ raw,
                                                              long long n, uint32_t *__restrict__ scratch) {
    __shared__ uint32_t lin[2 * 624];
    const int t = threadIdx.x;
    for (int i = t; i < 624; i += 256) lin[i] = state[i];
    const int pos = (int)state[624];
    __syncthreads();
    const int h = 624 - pos;
    for (int i = t; i < 624; i += 256) {
        const long long j = (long long)i - pos;
        if (j >= 0 && j < n) raw[j] = lin[i];
    }
    for (int base = 624; base < 2 * 624; base += 227) {      // x[j] = x[j-227] ^ A(x[j-624], x[j-623]), 227 at a time
        const int j = base + t;
        if (t < 227 && j < 2 * 624) lin[j] = lin[j - 227] ^ mt_twist(lin[j - 624], lin[j - 623]);
        __syncthreads();
    }
    uint32_t *s0 = scratch + MT_META;
    for (int i = t; i < 624; i += 256) s0[i] = lin[624 + i];
    if (t == 0) {
        s0[624] = 0;
        scratch[0] = (uint32_t)h;
    }
}

// One level of the jump tree: chunk state dst = src + dst_off is T^(624 * 2^q words) of chunk state src = b * src_step,
// i.e. word m of it = XOR over the set bits i of poly (= x^(624 * 2^q) mod phi) of z[i + m], z = the stream from state src
// on.  `split` workgroups share one jump: each regenerates the window (88 barrier steps) and folds its slice of the
// polynomial's words into the (pre-zeroed) destination with atomic XORs.
__global__ __launch_bounds__(256) void mt19937_jump_kernel(uint32_t *__restrict__ states, const uint32_t *__restrict__ poly,
                                                           int src_step, int dst_off, int nchunks, int split) {
    extern __shared__ uint32_t win[];                 // MT_WIN + 1 words
    const int t = threadIdx.x;
    const int job = blockIdx.x / split, part = blockIdx.x % split;
    const long long src = (long long)job * src_step, dst = src + dst_off;
    if (dst >= nchunks) return;
    const int w_per = (MT_N + split - 1) / split;     // polynomial words (32 coefficients each) per workgroup
    const int w0 = part * w_per, w1 = min(MT_N, w0 + w_per);
    if (w0 >= w1) return;
    const int need = min(MT_WIN, 32 * w1 + MT_N);     // window words this slice reads
    const uint32_t *s = states + src * MT_STATE_STRIDE;
    for (int i = t; i < 624; i += 256) win[i] = s[i];
    __syncthreads();
    for (int base = 624; base < need; base += 227) {
        const int j = base + t;
        if (t < 227 && j < need) win[j] = win[j - 227] ^ mt_twist(win[j - 624], win[j - 623]);
        __syncthreads();
    }
    uint32_t a0 = 0, a1 = 0, a2 = 0;
    const bool third = t + 512 < 624;
    for (int w = w0; w < w1; w++) {
        uint32_t bits = poly[w];                      // (uniform over the workgroup)
        while (bits) {
            const int i = 32 * w + __builtin_ctz(bits);
            bits &= bits - 1;
            if (i >= MT_DEG) break;
            a0 ^= win[i + t];
            a1 ^= win[i + t + 256];
            if (third) a2 ^= win[i + t + 512];
        }
    }
    uint32_t *d = states + dst * MT_STATE_STRIDE;
    if (split == 1) {
        d[t] = a0;
        d[t + 256] = a1;
        if (third) d[t + 512] = a2;
    } else {
        atomicXor(d + t, a0);
        atomicXor(d + t + 256, a1);
        if (third) atomicXor(d + t + 512, a2);
    }
}

struct MtPlan {
    int p;            // a chunk is 2^p blocks of 624 words
    long long chunks; // upper bound: the head taken from the caller's state only shortens the tail
};

MtPlan mt_plan(long long n) {
    MtPlan pl = {0, 0};
    if (n < MT_PAR_MIN) return pl;
    const long long blocks = (n + 623) / 624;
    // 512 .. 1024 chunks: enough workgroups to fill 256 CUs several times over while the jump tree stays 10 levels deep
    while (((blocks + (1ll << pl.p) - 1) >> pl.p) > 1024) pl.p++;
    pl.chunks = (blocks + (1ll << pl.p) - 1) >> pl.p;
    return pl;
}

inline int last_error() { return (int)hipGetLastError(); }

}  // namespace

extern "C" {

int64_t hpf_hip_mt19937_scratch_words(int64_t n) {
    const MtPlan pl = mt_plan(n);
    if (pl.chunks == 0) return 0;
    int levels = 0;
    while ((1ll << levels) < pl.chunks) levels++;
    return MT_META + pl.chunks * MT_STATE_STRIDE + (int64_t)levels * MT_N;
}

int hpf_hip_mt19937_jump_poly(int q, uint32_t *out) {
    if (q < 0 || q >= MT_MAX_Q || !out) return HPF_EINVAL;
    std::call_once(g_polys_once, build_polys);
    memcpy(out, &g_polys[(size_t)q * MT_N], MT_N * sizeof(uint32_t));
    return 0;
}

int hpf_hip_mt19937_words(uint32_t *state, uint32_t *raw, int64_t n, uint32_t *scratch, void *stream) {
    if (!state || n < 0 || (n > 0 && !raw)) return HPF_EINVAL;
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const MtPlan pl = mt_plan(n);
    if (pl.chunks == 0 || !scratch) {      // short draw (or no scratch given): one workgroup walks the recurrence
        hipLaunchKernelGGL(mt19937_words_kernel, dim3(1), dim3(256), 0, st, (const uint32_t *)state, 0ll, raw,
                           (long long)n, (long long)n, (const uint32_t *)nullptr, state);
        return last_error();
    }
    int levels = 0;
    while ((1ll << levels) < pl.chunks) levels++;
    if (pl.p + levels > MT_MAX_Q) return HPF_EUNSUPPORTED;
    std::call_once(g_polys_once, build_polys);
    uint32_t *states = scratch + MT_META;
    uint32_t *polys = states + pl.chunks * MT_STATE_STRIDE;
    hipError_t e = hipMemsetAsync(states, 0, (size_t)pl.chunks * MT_STATE_STRIDE * sizeof(uint32_t), st);
    if (e != hipSuccess) return (int)e;
    // level l of the tree (from the top) uses x^(624 * 2^(p + levels - 1 - l)); the table is static host memory
    e = hipMemcpyAsync(polys, &g_polys[(size_t)pl.p * MT_N], (size_t)levels * MT_N * sizeof(uint32_t),
                       hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(mt19937_prepare_kernel, dim3(1), dim3(256), 0, st, (const uint32_t *)state, raw, (long long)n,
                       scratch);
    const size_t lds = (size_t)(MT_WIN + 1) * sizeof(uint32_t);
    static std::once_flag attr_once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(attr_once, [&] {
        attr_err = hipFuncSetAttribute(reinterpret_cast<const void *>(mt19937_jump_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    if (attr_err != hipSuccess) return (int)attr_err;
    for (int q = levels - 1; q >= 0; q--) {
        const long long step = 2ll << q, off = 1ll << q;
        const long long jobs = (pl.chunks - off + step - 1) / step;     // sources b * step with b * step + off < chunks
        if (jobs <= 0) continue;
        int split = (int)(512 / jobs);
        split = split < 1 ? 1 : (split > 32 ? 32 : split);
        hipLaunchKernelGGL(mt19937_jump_kernel, dim3((unsigned)(jobs * split)), dim3(256), lds, st, states,
                           (const uint32_t *)(polys + (size_t)q * MT_N), (int)step, (int)off, (int)pl.chunks, split);
    }
    hipLaunchKernelGGL(mt19937_words_kernel, dim3((unsigned)pl.chunks), dim3(256), 0, st, (const uint32_t *)states,
                       (long long)MT_STATE_STRIDE, raw, 624ll << pl.p, (long long)n, (const uint32_t *)scratch, state);
    return last_error();
}

}  // extern "C"
