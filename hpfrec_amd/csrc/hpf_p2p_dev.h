// hpf_p2p_dev.h -- device-side helpers of the direct (peer-mapped) exchange; shared by hpf_p2p.hip and hpf_hip.hip.
//
// Nothing in the reference corresponds to this (it is single-node OpenMP, cython_loops.pxi:4); it carries SURVEY.md
// section 8(e)'s exchange of the item statistics without a collective library: every rank maps every other rank's
// exchange memory (hipIpc*) and the kernels of the iteration read it directly, ordered by flag words.
//
// Memory model used (gfx950, coarse-grained hipMalloc data + one fine-grained control block per rank):
//   * DATA buffers are only ever PULLED: written by their owner with plain stores, read by peers after
//       owner:  kernel end (L2 write-back at the kernel boundary) -> a LATER kernel/prologue: system-scope release fence
//               -> flag store into every peer's control block
//       peer:   poll its OWN control block (system-scope relaxed loads) -> system-scope acquire fence (invalidates L1 and
//               the L2 lines of non-local memory) -> __syncthreads() -> plain loads of the owner's buffer.
//   * The CONTROL block is fine-grained (uncached at the device level), written remotely and read locally with
//     system-scope atomics only.  Flags are monotonic epochs (never reset); k-float vectors travel as 8-byte
//     {value, epoch} granules written by ONE atomic store each, so they need no fence and no separate flag.
//   * Every poll is bounded by a wall-clock budget: on expiry the kernel sets the control block's error word and goes on
//     (the results are garbage, the host reports HPF_ETIMEOUT at the next status check) -- a missing peer can cost a
//     fit, never the GPU.  Once the error word is set no later poll of this rank spins at all: a dead peer costs ONE budget,
//     not one per wait of the iterations already queued.
#ifndef HPF_P2P_DEV_H
#define HPF_P2P_DEV_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hpf_hip.h"

namespace hpf_p2p {

// control block layout, in 32-bit words: [0] error word, [1..15] reserved, then flags[kind][src], then the vector slots
// (8-byte granules) vec[which][parity][src][ld]
constexpr int HDR_WORDS = 16;
__host__ __device__ inline size_t flag_word(int kind, int src) {
    return (size_t)HDR_WORDS + (size_t)kind * HPF_P2P_MAX_RANKS + src;
}
__host__ __device__ inline size_t vec_base_bytes() {
    return ((size_t)HDR_WORDS + (size_t)HPF_P2P_NKINDS * HPF_P2P_MAX_RANKS) * 4;     // a multiple of 8
}
__host__ __device__ inline size_t vec_granule(int which, int parity, int src, int ld, int c) {
    return (((size_t)which * 2 + parity) * HPF_P2P_MAX_RANKS + src) * (size_t)ld + c;
}
__host__ __device__ inline size_t ctrl_bytes(int ld) {
    return vec_base_bytes() + (size_t)HPF_P2P_NVEC * 2 * HPF_P2P_MAX_RANKS * (size_t)ld * 8;
}

struct Peers {      // one rank's view of the job's control blocks (kernel argument, by value)
    int world, rank;
    uint32_t *ctrl[HPF_P2P_MAX_RANKS];   // ctrl[rank]: the local block; a dry run maps every entry to the local block
    long long timeout_ticks;             // poll budget in wall_clock64() ticks (100 MHz)
    int emulate;                         // 1: single-process emulation -- nothing is waited for
};

// (hpf_p2p.hip) the kernel-argument view of a connected region and its mapped data buffers; false: not connected
bool region_view(void *region, Peers *pp, void **data, int *world, int *rank, int *ld);

#if defined(__HIPCC__)
__device__ __forceinline__ uint32_t ld_sys(const uint32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void st_sys(uint32_t *p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned long long ld_sys64(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void st_sys64(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// this lane waits until flags[kind][src] of the LOCAL control block has reached `epoch` (wrap-safe); false on time-out
__device__ __forceinline__ bool wait_flag(const Peers &pp, int kind, int src, uint32_t epoch) {
    if (pp.emulate) return true;
    const uint32_t *f = pp.ctrl[pp.rank] + flag_word(kind, src);
    if ((int32_t)(ld_sys(f) - epoch) >= 0) return true;
    if (ld_sys(pp.ctrl[pp.rank]) != 0u) return false;        // (an earlier wait of this rank has timed out)
    const long long t0 = wall_clock64();
    while ((int32_t)(ld_sys(f) - epoch) < 0) {
        if (wall_clock64() - t0 > pp.timeout_ticks) {
            atomicOr(pp.ctrl[pp.rank], 1u << (kind & 15));        // error word: which kind of flag never came
            return false;
        }
        __builtin_amdgcn_s_sleep(8);
    }
    return true;
}

// Block-level consumer side: lanes src < world of wave 0 with bit src of `mask` set wait for flags[kind][src]; then the
// system-scope acquire and a barrier -- after it every thread of the block may read the producers' buffers with plain
// loads.  Call with ALL threads of the block.
__device__ __forceinline__ void block_acquire(const Peers &pp, int kind, uint32_t epoch, uint32_t mask) {
    if (threadIdx.x < (unsigned)pp.world && ((mask >> threadIdx.x) & 1u) && (int)threadIdx.x != pp.rank)
        (void)wait_flag(pp, kind, (int)threadIdx.x, epoch);
    if (threadIdx.x < 64) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    __syncthreads();
}

// the same for a flag this rank raises for ITSELF (a kernel on another stream of this process is the producer)
__device__ __forceinline__ void block_acquire_self(const Peers &pp, int kind, uint32_t epoch) {
    if (threadIdx.x == 0) {      // (also in a single-process emulation: the producer is a stream of this process)
        const uint32_t *f = pp.ctrl[pp.rank] + flag_word(kind, pp.rank);
        const long long t0 = wall_clock64();
        const bool failed = ld_sys(pp.ctrl[pp.rank]) != 0u;
        while ((int32_t)(ld_sys(f) - epoch) < 0) {
            if (failed) break;
            if (wall_clock64() - t0 > pp.timeout_ticks) {
                atomicOr(pp.ctrl[pp.rank], 1u << (kind & 15));
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    if (threadIdx.x < 64) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    __syncthreads();
}

// Producer side, by ONE wavefront whose block's (or whose predecessors' on the stream) stores are complete: system-scope
// release, then lane p stores the epoch into peer p's flags[kind][rank] (the local block included: a rank also
// "signals itself", so consumers need not special-case their own data)
__device__ __forceinline__ void wave_signal(const Peers &pp, int kind, uint32_t epoch) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (restated: the compiler may drop the wait after buffer_wbl2)
    const int lane = threadIdx.x & 63;
    if (lane < pp.world) st_sys(pp.ctrl[lane] + flag_word(kind, pp.rank), epoch);
}

// k-float vectors as {value, epoch} granules.  publish: this rank's value of column c goes to every peer's slot
// [which][parity][rank][c]; collect: the sum over the ranks, in rank order (so every rank computes the same float).
__device__ __forceinline__ void vec_publish(const Peers &pp, int which, uint32_t epoch, int ld, int c, float v) {
    const unsigned long long g = ((unsigned long long)epoch << 32) | (unsigned long long)__float_as_uint(v);
    for (int p = 0; p < pp.world; p++) {
        unsigned long long *slot = reinterpret_cast<unsigned long long *>(
                                       reinterpret_cast<char *>(pp.ctrl[pp.emulate ? pp.rank : p]) + vec_base_bytes()) +
                                   vec_granule(which, (int)(epoch & 1u), pp.rank, ld, c);
        st_sys64(slot, g);
        if (pp.emulate) break;
    }
}
__device__ __forceinline__ float vec_collect(const Peers &pp, int which, uint32_t epoch, int ld, int c) {
    const unsigned long long *base = reinterpret_cast<const unsigned long long *>(
        reinterpret_cast<const char *>(pp.ctrl[pp.rank]) + vec_base_bytes());
    float s = 0.f;
    const long long t0 = wall_clock64();
    const bool failed = ld_sys(pp.ctrl[pp.rank]) != 0u;
    for (int src = 0; src < pp.world; src++) {
        // (emulation: this rank alone -- the "sum" is the local value, like an all-reduce on a one-rank communicator)
        const unsigned long long *slot = base + vec_granule(which, (int)(epoch & 1u), pp.emulate ? pp.rank : src, ld, c);
        unsigned long long g = ld_sys64(slot);
        while ((uint32_t)(g >> 32) != epoch) {
            if (failed) break;
            if (wall_clock64() - t0 > pp.timeout_ticks) {
                atomicOr(pp.ctrl[pp.rank], 1u << (16 + which));
                break;
            }
            __builtin_amdgcn_s_sleep(2);
            g = ld_sys64(slot);
        }
        const float v = __uint_as_float((uint32_t)g);
        s = (src == 0) ? v : s + v;
        if (pp.emulate) break;
    }
    return s;
}
#endif  // __HIPCC__

}  // namespace hpf_p2p
#endif  // HPF_P2P_DEV_H
