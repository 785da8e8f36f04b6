// hpf_shard.hip -- one rank's whole user-sharded CAVI iteration issued from C (include/hpf_hip.h, last section).
//
// What this replaces: nothing in the reference (it is single-node OpenMP, cython_loops.pxi:227-259); it is the multi-GPU
// schedule of SURVEY.md section 8(e) / DESIGN.md section 6 ("scatter" exchange), previously issued call by call from
// Python (hpfrec_amd/cavi.py, FullBatchCavi._iterate_scatter -- kept as the fallback and as the gloo/CPU test path).
// Host code only: the kernels are reached through the public C ABI of hpf_hip.hip, the collectives through RCCL's C API
// resolved at run time from the librccl.so the process has already loaded (never linked: one RCCL instance must serve
// torch.distributed and this library).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types, enums and prototypes only (decltype below); no RCCL symbol is referenced
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <mutex>
#include <new>
#include <vector>

#include "hpf_hip.h"
#include "hpf_internal.h"

namespace {

struct RcclApi {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclReduceScatter) ReduceScatter = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    bool ok = false;
};

RcclApi g_rccl;
std::mutex g_rccl_mutex;

inline int rccl_rc(ncclResult_t r) { return r == ncclSuccess ? 0 : HPF_ERCCL_BASE - (int)r; }

#define HPF_TRY(expr)              \
    do {                           \
        const int rc__ = (expr);   \
        if (rc__ != 0) return rc__; \
    } while (0)
#define HIP_TRY(expr)                                  \
    do {                                               \
        const hipError_t e__ = (expr);                 \
        if (e__ != hipSuccess) return (int)e__;        \
    } while (0)

// ---- trace mode (hpf_shard_desc.dry_run == 2): what an iteration ISSUES, in order, without a device --------------------
// Every kernel launch, collective, event record / wait and copy of the schedule is appended to the plan's trace instead of
// being issued (events and streams are opaque handles then).  tests/test_host_logic.py reads the traces of all ranks of a
// job on a machine without a GPU and checks what a multi-rank run depends on: identical collective sequences on every
// rank, every wait after its record, each schedule's order of kernels.
struct Tracer {
    std::vector<hpf_shard_trace_rec> recs;
    void add(int kind, int id, const void *stream, int64_t arg) {
        hpf_shard_trace_rec r;
        r.kind = kind;
        r.id = id;
        r.stream = (int64_t)(intptr_t)stream;
        r.arg = arg;
        recs.push_back(r);
    }
};
thread_local Tracer *g_tr = nullptr;
struct TraceScope {       // the tracer of the plan being issued, for the wrappers below
    Tracer *prev;
    explicit TraceScope(Tracer *t) : prev(g_tr) { g_tr = t; }
    ~TraceScope() { g_tr = prev; }
};

template <typename T>
const void *last_arg(T v) { return (const void *)v; }
template <typename T, typename... R>
const void *last_arg(T, R... rest) { return last_arg(rest...); }

inline hipError_t tr_event_record(hipEvent_t e, hipStream_t st) {
    if (g_tr) {
        g_tr->add(HPF_TRACE_RECORD, 0, st, (int64_t)(intptr_t)e);
        return hipSuccess;
    }
    return hipEventRecord(e, st);
}
inline hipError_t tr_stream_wait(hipStream_t st, hipEvent_t e, unsigned flags) {
    if (g_tr) {
        g_tr->add(HPF_TRACE_WAIT, 0, st, (int64_t)(intptr_t)e);
        return hipSuccess;
    }
    return hipStreamWaitEvent(st, e, flags);
}
inline hipError_t tr_memcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t st) {
    if (g_tr) {
        g_tr->add(HPF_TRACE_COPY, 0, st, (int64_t)bytes);
        return hipSuccess;
    }
    return hipMemcpyAsync(dst, src, bytes, kind, st);
}
#define hipEventRecord tr_event_record
#define hipStreamWaitEvent tr_stream_wait
#define hipMemcpyAsync tr_memcpy
// the kernels of the schedule are reached through the library's own C entries; their stream is the last argument
#define TR_KERNEL(id, fn, ...) (g_tr ? (g_tr->add(HPF_TRACE_KERNEL, id, last_arg(__VA_ARGS__), 0), 0) : fn(__VA_ARGS__))
#define hpf_hip_sweep_f32(...) TR_KERNEL(HPF_TRACE_K_SWEEP, (hpf_hip_sweep_f32), __VA_ARGS__)
#define hpf_hip_segsum_f32(...) TR_KERNEL(HPF_TRACE_K_SEGSUM, (hpf_hip_segsum_f32), __VA_ARGS__)
#define hpf_hip_sweep_finalize_f32(...) TR_KERNEL(HPF_TRACE_K_SWEEP_FINALIZE, (hpf_hip_sweep_finalize_f32), __VA_ARGS__)
#define hpf_hip_row_finalize_f32(...) TR_KERNEL(HPF_TRACE_K_ROW_FINALIZE, (hpf_hip_row_finalize_f32), __VA_ARGS__)
#define hpf_hip_row_finalize_ranges_f32(...) TR_KERNEL(HPF_TRACE_K_ROW_FINALIZE_RANGES, (hpf_hip_row_finalize_ranges_f32), __VA_ARGS__)
#define hpf_hip_colsum_reduce_f32(...) TR_KERNEL(HPF_TRACE_K_COLSUM_REDUCE, (hpf_hip_colsum_reduce_f32), __VA_ARGS__)
#define hpf_hip_item_shape_rows_f32(...) TR_KERNEL(HPF_TRACE_K_ITEM_SHAPE, (hpf_hip_item_shape_rows_f32), __VA_ARGS__)
#define hpf_hip_item_apply_rows_f32(...) TR_KERNEL(HPF_TRACE_K_ITEM_APPLY, (hpf_hip_item_apply_rows_f32), __VA_ARGS__)

struct Plan {
    Tracer tracer;
    bool tracing;                     // dry_run == 2
    hpf_shard_desc d;
    hipStream_t xs;
    hipEvent_t sw_done[HPF_MAX_ROW_RANGES], ag_done[HPF_MAX_ROW_RANGES], csT_ready, start;
    hipEvent_t p2_done, csB_done;     // gather-early: the apply kernel has run (compute) / colsum(Beta) is summed (exchange)
    int nevents;
    int64_t lo[HPF_MAX_ROW_RANGES], hi[HPF_MAX_ROW_RANGES];   // the ranges' rows, in issue order
    int64_t total;                    // rows of this rank's slices, all ranges
    int64_t m[HPF_MAX_ROW_RANGES];    // rows of this rank's slice of range j (= (hi-lo)/world)
    int64_t t0[HPF_MAX_ROW_RANGES];   // first row of that slice in acc_own / e_own
    int nfin;                         // slices with at least one real (non-pad) row
    int64_t fin_rows[HPF_MAX_ROW_RANGES], fin_acc[HPF_MAX_ROW_RANGES], fin_row0[HPF_MAX_ROW_RANGES];
    bool fresh;                       // nothing of this plan in flight on the exchange stream
    float *tiny;                      // dry runs with a communicator: the element of the stand-in RCCL call (plan-owned)
    // direct schedule: the rank's connected exchange region as the kernels see it
    hpf_p2p::Peers pp;
    const hpf_p2p::Peers *peers_dev;
    void *peer_data[HPF_P2P_MAX_RANKS];
    uint32_t epoch;                   // of the iteration in flight (hpf_hip_p2p_region_next_epoch)
};

// dry runs with an assumed bus bandwidth: the stream is held for the time the links would take by ONE wavefront that
// sleeps on the wall clock (100 MHz on gfx950) -- it takes a SIMD slot and no memory bandwidth
__global__ void link_time_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

// the same with the footprint of a collective library's kernel: 256 threads, 128 VGPRs, 64 KB of LDS per workgroup -- it
// starts only where a CU has that much free, which sweeps that fill every wave slot do not leave
__global__ __launch_bounds__(256) void link_time_fat_kernel(long long ticks, int *sink) {
    extern __shared__ int fat_lds[];
    asm volatile("" ::: "v127");
    fat_lds[threadIdx.x] = (int)threadIdx.x;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (ticks < 0) sink[threadIdx.x] = fat_lds[255 - threadIdx.x];
}

// one collective of the schedule.  `slice`: element offset of this rank's part inside the world-sized buffer
int collective(Plan *p, int op, const float *send, float *recv, int64_t count, hipStream_t st, bool small = false) {
    const hpf_shard_desc &d = p->d;
    if (p->tracing) {
        p->tracer.add(HPF_TRACE_COLLECTIVE, op | (small ? 0x100 : 0), st, count);
        return 0;
    }
    if (d.dry_run) {
        // this rank alone: the one-rank form of the collective on this rank's slice ...
        const size_t bytes = (size_t)count * sizeof(float);
        if (op == HPF_COLL_REDUCE_SCATTER)
            HIP_TRY(hipMemcpyAsync(recv, send + (size_t)d.rank * count, bytes, hipMemcpyDeviceToDevice, st));
        else if (op == HPF_COLL_ALL_GATHER)
            HIP_TRY(hipMemcpyAsync(recv + (size_t)d.rank * count, send, bytes, hipMemcpyDeviceToDevice, st));
        if (d.dry_run_busbw_GBps > 0.f) {
            // bytes a rank moves: (world-1)/world of the whole buffer (all-reduce: twice that)
            const double whole = (double)count * sizeof(float) * (op == HPF_COLL_ALL_REDUCE ? 2.0 : (double)d.world);
            const double us = d.dry_run_latency_us + whole * (d.world - 1) / d.world / (d.dry_run_busbw_GBps * 1e3);
            if (d.dry_run_footprint_blocks > 0 && op != HPF_COLL_ALL_REDUCE)
                hipLaunchKernelGGL(link_time_fat_kernel, dim3(d.dry_run_footprint_blocks), dim3(256), 64 * 1024, st,
                                   (long long)(us * 100.0), (int *)p->tiny);
            else
                hipLaunchKernelGGL(link_time_kernel, dim3(1), dim3(64), 0, st, (long long)(us * 100.0));
        }
        // ... plus a real (one-rank, one-element: identity) RCCL call, so that a launch of RCCL's is paid
        if (d.comm) {
            if (!g_rccl.ok) return HPF_ENOLIB;
            return rccl_rc(g_rccl.AllReduce(p->tiny, p->tiny, 1, ncclFloat32, ncclSum, (ncclComm_t)d.comm, st));
        }
        return 0;
    }
    if (d.coll) return d.coll(d.coll_ctx, op, send, recv, count, (void *)st);
    if (d.world == 1 && !d.comm) {      // a single rank without a communicator: the identity forms
        if (op != HPF_COLL_ALL_REDUCE && recv != send)
            HIP_TRY(hipMemcpyAsync(recv, send, (size_t)count * sizeof(float), hipMemcpyDeviceToDevice, st));
        return 0;
    }
    if (!g_rccl.ok) return HPF_ENOLIB;
    ncclComm_t comm = (ncclComm_t)d.comm;
    switch (op) {
        case HPF_COLL_ALL_REDUCE:
            return rccl_rc(g_rccl.AllReduce(send, recv, (size_t)count, ncclFloat32, ncclSum, comm, st));
        case HPF_COLL_REDUCE_SCATTER:
            return rccl_rc(g_rccl.ReduceScatter(send, recv, (size_t)count, ncclFloat32, ncclSum, comm, st));
        case HPF_COLL_ALL_GATHER:
            return rccl_rc(g_rccl.AllGather(send, recv, (size_t)count, ncclFloat32, comm, st));
    }
    return HPF_EINVAL;
}

int all_gather_range(Plan *p, int j, hipStream_t st) {       // padded rows straight into the replicated E table
    const hpf_shard_desc &d = p->d;
    const hpf_shard_range &r = d.ranges[j];
    return collective(p, HPF_COLL_ALL_GATHER, d.e_own + (size_t)p->t0[j] * d.ld, d.eB + (size_t)r.lo * d.ld,
                      p->m[j] * d.ld, st);
}

int reduce_scatter_range(Plan *p, int j, hipStream_t st) {
    const hpf_shard_desc &d = p->d;
    return collective(p, HPF_COLL_REDUCE_SCATTER, d.acc_i + (size_t)d.ranges[j].lo * d.k,
                      d.acc_own + (size_t)p->t0[j] * d.k, p->m[j] * d.k, st);
}

// ---- direct schedule: the kernels that carry the exchange (hpf_internal.h), traced like the others ------------------------
// sig_kind >= 0: the launch raises flags[sig_kind][rank] = epoch in every peer on entry; < 0: no signal
inline hpf_direct::Signal raise(Plan *p, int sig_kind) {
    return sig_kind >= 0 ? hpf_direct::Signal{p->peers_dev, sig_kind, p->epoch} : hpf_direct::Signal{nullptr, 0, 0};
}

// an emulated link time for a pull of `bytes` (dry runs with dry_run_busbw_GBps > 0; one sleeping wavefront)
int pull_link_time(Plan *p, double bytes, hipStream_t st) {
    const hpf_shard_desc &d = p->d;
    if (!d.dry_run || p->tracing || !(d.dry_run_busbw_GBps > 0.f)) return 0;
    const double us = d.dry_run_latency_us + bytes / (d.dry_run_busbw_GBps * 1e3);
    hipLaunchKernelGGL(link_time_kernel, dim3(1), dim3(64), 0, st, (long long)(us * 100.0));
    return (int)hipGetLastError();
}

int direct_item_sweep(Plan *p, int j, const float *eT, int sig_kind, hipStream_t cs) {
    const hpf_shard_desc &d = p->d;
    const hpf_shard_range &r = d.ranges[j];
    if (g_tr) {
        g_tr->add(HPF_TRACE_KERNEL, HPF_TRACE_K_SWEEP, cs, sig_kind >= 0 ? 1 + sig_kind : 0);
        return 0;
    }
    return hpf_direct::sweep(d.i_segs + r.seg_lo, r.nseg, d.i_idx, d.i_y, d.eB, eT, d.part_i + (size_t)r.seg_lo * d.ld,
                             d.acc_i, d.k, d.k, d.ld, r.short_rows, d.item_sweep_grid, raise(p, sig_kind), cs);
}

// range j on the exchange stream: ONE wavefront waits until every rank -- this one included: its flag is raised on entry of
// the launch that follows the range's sweep on the compute stream, so no stream event is needed -- has range j complete
// in its exchange buffer; then this rank's slice is summed straight out of the N buffers (the reduce-scatter)
int direct_pull_reduce(Plan *p, int j, hipStream_t xs) {
    const hpf_shard_desc &d = p->d;
    const int kind = HPF_P2P_FLAG_SWEPT(j);
    if (g_tr) {
        g_tr->add(HPF_TRACE_KERNEL, HPF_TRACE_K_WAIT, xs, (int64_t)(1u << kind));
        g_tr->add(HPF_TRACE_KERNEL, HPF_TRACE_K_PULL_REDUCE, xs, j);
        return 0;
    }
    // a single-process emulation reads every rank's slice (the traffic) and counts its own (the value)
    const uint32_t mask = p->pp.emulate ? (1u << d.rank) : ((d.world >= 32) ? 0xFFFFFFFFu : ((1u << d.world) - 1u));
    HPF_TRY(hpf_p2p::wait_flags(d.p2p_region, 1u << kind, p->epoch, 0xFFFFFFFFu, kind, xs));
    const int64_t o0 = p->lo[j] + (int64_t)d.rank * p->m[j];
    HPF_TRY(hpf_p2p::pull_reduce(d.p2p_region, d.p2p_acc_offset + o0 * d.k * (int64_t)sizeof(float),
                                 d.acc_own + (size_t)p->t0[j] * d.k, p->m[j] * d.k, mask, d.direct_pull_grid, xs));
    return pull_link_time(p, (double)p->m[j] * d.k * 4.0 * (d.world - 1), xs);
}

int direct_colsum_allreduce(Plan *p, const float *part, int rows, float *out, int which, uint32_t then_wait_kinds,
                            int then_wait_self, hipStream_t st) {
    if (g_tr) {
        g_tr->add(HPF_TRACE_KERNEL, HPF_TRACE_K_COLSUM_ALLREDUCE, st,
                  which | ((int64_t)(then_wait_self + 1) << 8) | ((int64_t)then_wait_kinds << 16));
        return 0;
    }
    return hpf_direct::colsum_reduce_allreduce(part, rows, out, p->d.ld, p->peers_dev, which, p->epoch, then_wait_kinds,
                                               then_wait_self, st);
}

}  // namespace

extern "C" {

int hpf_hip_rccl_open(const char *librccl_path) {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.ok) return 0;
    const char *path = librccl_path ? librccl_path : "librccl.so";
    // the instance the process has loaded already (torch's), else load it
    void *h = dlopen(path, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    if (!h) h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
    if (!h) return HPF_ENOLIB;
    RcclApi a;
    a.handle = h;
#define HPF_SYM(field, name)                                        \
    a.field = reinterpret_cast<decltype(a.field)>(dlsym(h, name));  \
    if (!a.field) return HPF_ENOLIB;
    HPF_SYM(GetUniqueId, "ncclGetUniqueId")
    HPF_SYM(CommInitRank, "ncclCommInitRank")
    HPF_SYM(CommDestroy, "ncclCommDestroy")
    HPF_SYM(CommCount, "ncclCommCount")
    HPF_SYM(AllReduce, "ncclAllReduce")
    HPF_SYM(ReduceScatter, "ncclReduceScatter")
    HPF_SYM(AllGather, "ncclAllGather")
#undef HPF_SYM
    a.ok = true;
    g_rccl = a;
    return 0;
}

int hpf_hip_rccl_unique_id(uint8_t id[128]) {
    if (!id) return HPF_EINVAL;
    if (!g_rccl.ok) return HPF_ENOLIB;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId uid;
    const int rc = rccl_rc(g_rccl.GetUniqueId(&uid));
    if (rc == 0) memcpy(id, &uid, sizeof(uid));
    return rc;
}

int hpf_hip_rccl_comm_init(void **comm, int world, int rank, const uint8_t id[128]) {
    if (!comm || !id || world <= 0 || rank < 0 || rank >= world) return HPF_EINVAL;
    if (!g_rccl.ok) return HPF_ENOLIB;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclComm_t c = nullptr;
    const int rc = rccl_rc(g_rccl.CommInitRank(&c, world, uid, rank));
    if (rc == 0) *comm = (void *)c;
    return rc;
}

int hpf_hip_rccl_comm_count(void *comm, int *count) {
    if (!comm || !count) return HPF_EINVAL;
    if (!g_rccl.ok) return HPF_ENOLIB;
    return rccl_rc(g_rccl.CommCount((ncclComm_t)comm, count));
}

int hpf_hip_rccl_comm_destroy(void *comm) {
    if (!comm) return HPF_EINVAL;
    if (!g_rccl.ok) return HPF_ENOLIB;
    return rccl_rc(g_rccl.CommDestroy((ncclComm_t)comm));
}

int hpf_hip_rccl_all_reduce_f32(void *comm, float *buf, int64_t n, void *stream) {
    if (!comm || !buf || n < 0) return HPF_EINVAL;
    if (!g_rccl.ok) return HPF_ENOLIB;
    if (n == 0) return 0;
    return rccl_rc(g_rccl.AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, (ncclComm_t)comm, (hipStream_t)stream));
}

int hpf_hip_rccl_reduce_scatter_f32(void *comm, const float *send, float *recv, int64_t recv_n, void *stream) {
    if (!comm || !send || !recv || recv_n < 0) return HPF_EINVAL;
    if (!g_rccl.ok) return HPF_ENOLIB;
    if (recv_n == 0) return 0;
    return rccl_rc(g_rccl.ReduceScatter(send, recv, (size_t)recv_n, ncclFloat32, ncclSum, (ncclComm_t)comm,
                                        (hipStream_t)stream));
}

int hpf_hip_rccl_all_gather_f32(void *comm, const float *send, float *recv, int64_t send_n, void *stream) {
    if (!comm || !send || !recv || send_n < 0) return HPF_EINVAL;
    if (!g_rccl.ok) return HPF_ENOLIB;
    if (send_n == 0) return 0;
    return rccl_rc(g_rccl.AllGather(send, recv, (size_t)send_n, ncclFloat32, (ncclComm_t)comm, (hipStream_t)stream));
}

int hpf_hip_shard_desc_layout(int64_t out[4]) {
    if (!out) return HPF_EINVAL;
    out[0] = (int64_t)sizeof(hpf_shard_desc);
    out[1] = (int64_t)offsetof(hpf_shard_desc, ranges);
    out[2] = (int64_t)offsetof(hpf_shard_desc, acc_i);
    out[3] = (int64_t)offsetof(hpf_shard_desc, dry_run);
    return 0;
}

int hpf_hip_shard_plan_create(const hpf_shard_desc *desc, void **plan) {
    if (!desc || !plan) return HPF_EINVAL;
    const hpf_shard_desc &d = *desc;
    if (d.world <= 0 || d.rank < 0 || d.rank >= d.world || d.k <= 0 || d.ld != hpf_hip_ld_for_k(d.k) || d.nU < 0 ||
        d.nI <= 0 || d.nranges <= 0 || d.nranges > HPF_MAX_ROW_RANGES)
        return HPF_EINVAL;          // (nU == 0: a rank without a single user still takes part in every exchange step)
    // (a rank may hold no nonzeros at all -- fewer users than ranks, or one user with most of them: u_nseg == 0 and every
    //  row is finished by the row finalizer over u_multi_rows; zero-length arrays have null pointers then)
    if (d.u_nseg < 0 || !d.u_row_seg_ptr || (d.u_nmulti > 0 && !d.u_multi_rows) || !d.i_row_seg_ptr)
        return HPF_EINVAL;
    if (d.u_nseg > 0 && (!d.u_segs || !d.u_idx || !d.u_y || !d.i_segs || !d.i_idx || !d.i_y)) return HPF_EINVAL;
    if (!d.eB || !d.part_u || !d.part_i || !d.Lambda_shp ||
        !d.Beta || !d.t_rte || !d.t_rte_prev || !d.csT || !d.csB || !d.csB_used || !d.csT_part || !d.csB_part ||
        !d.acc_i || !d.acc_own || !d.e_own || !d.xstream)
        return HPF_EINVAL;
    if (d.nU > 0 && (!d.Gamma_shp || !d.Theta || !d.k_rte || !d.k_rte_prev)) return HPF_EINVAL;
    if (d.user_sweep_grid <= 0 || d.user_multi_grid <= 0 || d.user_sweep_grid + d.user_multi_grid > d.csT_part_rows ||
        d.csB_part_rows <= 0 || d.item_sweep_grid <= 0)
        return HPF_EINVAL;
    const bool tracing = d.dry_run == 2;
    hpf_p2p::Peers pp = {};
    void *peer_data[HPF_P2P_MAX_RANKS] = {};
    if (d.schedule == HPF_SCHEDULE_DIRECT) {
        if (d.e_own_ld != hpf_hip_gather_payload_ld(d.k) || !d.shp_own || (d.direct_prefetch && !d.ag_recv) ||
            d.world > HPF_P2P_MAX_RANKS || d.nranges > HPF_MAX_ROW_RANGES || d.csB_part_rows % d.world != 0 ||
            !d.acc_own || d.direct_pull_grid <= 0 || d.direct_gather_gx <= 0)
            return HPF_EINVAL;
        if (!tracing) {
            int w = 0, r = 0, ld = 0;
            if (!hpf_p2p::region_view(d.p2p_region, &pp, peer_data, &w, &r, &ld)) return HPF_EINVAL;
            if (w != d.world || r != d.rank || ld != d.ld || (d.p2p_acc_offset & 15) || (d.p2p_send_offset & 15) ||
                (const char *)d.acc_i != (const char *)peer_data[r] + d.p2p_acc_offset ||
                (const char *)d.e_own != (const char *)peer_data[r] + d.p2p_send_offset)
                return HPF_EINVAL;
            if (d.dry_run && !pp.emulate) return HPF_EINVAL;      // (a dry run stands alone: a region connected to itself)
        }
    } else if (d.schedule == HPF_SCHEDULE_GATHER_EARLY) {
        if (d.e_own_ld != hpf_hip_gather_payload_ld(d.k) || !d.ag_recv || !d.shp_own) return HPF_EINVAL;
    } else if (d.schedule != HPF_SCHEDULE_FINALIZE_THEN_GATHER) {
        return HPF_EINVAL;
    } else if (d.e_own_ld != d.ld) {
        return HPF_EINVAL;
    }
    if (d.schedule == HPF_SCHEDULE_DIRECT) {
        // no communicator, no callback: the exchange is in the kernels
    } else if (d.dry_run) {
        if (d.comm) {   // the stand-in call of a dry run must be an identity: a one-rank communicator
            int n = 0;
            HPF_TRY(hpf_hip_rccl_comm_count(d.comm, &n));
            if (n != 1) return HPF_EINVAL;
        }
    } else if (d.world > 1 && !d.comm && !d.coll) {
        return HPF_EINVAL;
    } else if (d.comm && !d.coll) {
        int n = 0;
        HPF_TRY(hpf_hip_rccl_comm_count(d.comm, &n));
        if (n != d.world) return HPF_EINVAL;
    }
    Plan *p = new (std::nothrow) Plan();
    if (!p) return (int)hipErrorOutOfMemory;
    p->d = d;
    p->xs = (hipStream_t)d.xstream;
    p->nevents = 0;
    p->nfin = 0;
    p->fresh = true;
    p->tiny = nullptr;
    p->tracing = d.dry_run == 2;
    p->pp = pp;
    p->peers_dev = (d.schedule == HPF_SCHEDULE_DIRECT && !p->tracing) ? hpf_p2p::region_peers_dev(d.p2p_region) : nullptr;
    for (int q = 0; q < HPF_P2P_MAX_RANKS; q++) p->peer_data[q] = peer_data[q];
    p->epoch = 0;
    if (!p->tracing && d.dry_run && d.comm && d.schedule != HPF_SCHEDULE_DIRECT) {
        const hipError_t e = hipMalloc((void **)&p->tiny, 256);
        if (e != hipSuccess) {
            delete p;
            return (int)e;
        }
        (void)hipMemset(p->tiny, 0, 256);
    }
    int64_t t = 0;
    for (int j = 0; j < d.nranges; j++) {
        const hpf_shard_range &r = d.ranges[j];
        if (r.lo < 0 || r.hi <= r.lo || (r.hi - r.lo) % d.world != 0 || r.nseg < 0 || r.seg_lo < 0 || r.nmulti < 0 ||
            (r.nmulti > 0 && !r.multi_rows)) {
            if (p->tiny) (void)hipFree(p->tiny);
            delete p;
            return HPF_EINVAL;
        }
        const int64_t m = (r.hi - r.lo) / d.world;
        const int64_t o0 = r.lo + (int64_t)d.rank * m;
        int64_t n_real = d.nI - o0;          // rows of the slice that are real items (the last range ends in pad rows)
        if (n_real > m) n_real = m;
        p->m[j] = m;
        p->t0[j] = t;
        p->lo[j] = r.lo;
        p->hi[j] = r.hi;
        if (n_real > 0) {
            p->fin_rows[p->nfin] = n_real;
            p->fin_acc[p->nfin] = t;
            p->fin_row0[p->nfin] = o0;
            p->nfin++;
        }
        t += m;
    }
    p->total = t;
    hipEvent_t *evs[2 * HPF_MAX_ROW_RANGES + 4];
    int ne = 0;
    for (int j = 0; j < d.nranges; j++) {
        evs[ne++] = &p->sw_done[j];
        evs[ne++] = &p->ag_done[j];
    }
    evs[ne++] = &p->csT_ready;
    evs[ne++] = &p->start;
    evs[ne++] = &p->p2_done;
    evs[ne++] = &p->csB_done;
    for (int i = 0; i < ne; i++) {
        if (p->tracing) {       // opaque handles: HPF_TRACE_EVENT_BASE + the event's index (sw_done j: 2j, ag_done j: 2j+1, ...)
            *evs[i] = (hipEvent_t)(intptr_t)(HPF_TRACE_EVENT_BASE + i);
            continue;
        }
        const hipError_t e = hipEventCreateWithFlags(evs[i], hipEventDisableTiming);
        if (e != hipSuccess) {
            for (int q = 0; q < i; q++) (void)hipEventDestroy(*evs[q]);
            if (p->tiny) (void)hipFree(p->tiny);
            delete p;
            return (int)e;
        }
    }
    p->nevents = ne;
    *plan = p;
    return 0;
}

int hpf_hip_shard_plan_destroy(void *plan) {
    if (!plan) return HPF_EINVAL;
    Plan *p = (Plan *)plan;
    if (p->tracing) {
        delete p;
        return 0;
    }
    for (int j = 0; j < p->d.nranges; j++) {
        (void)hipEventDestroy(p->sw_done[j]);
        (void)hipEventDestroy(p->ag_done[j]);
    }
    (void)hipEventDestroy(p->csT_ready);
    (void)hipEventDestroy(p->start);
    (void)hipEventDestroy(p->p2_done);
    (void)hipEventDestroy(p->csB_done);
    if (p->tiny) (void)hipFree(p->tiny);
    delete p;
    return 0;
}

int hpf_hip_shard_join(void *plan, void *stream) {
    if (!plan) return HPF_EINVAL;
    Plan *p = (Plan *)plan;
    TraceScope scope(p->tracing ? &p->tracer : nullptr);
    if (!p->fresh) {
        const hpf_shard_desc &d = p->d;
        hipStream_t st = (hipStream_t)stream;
        // the exchange stream is in order: the last thing on it is the last range's all-gather (+ unpack), or -- gather-
        // early -- the all-reduce of colsum(Beta)
        hipEvent_t last = (d.schedule == HPF_SCHEDULE_GATHER_EARLY) ? p->csB_done
                          : (d.schedule == HPF_SCHEDULE_DIRECT)     ? p->ag_done[0]
                                                                     : p->ag_done[d.nranges - 1];
        HIP_TRY(hipStreamWaitEvent(st, last, 0));
        p->fresh = true;
    }
    return 0;
}

static int iterate_gather_early(Plan *p, const float *eT, float *eT_next, int store, hipStream_t cs) {
    const hpf_shard_desc &d = p->d;
    hipStream_t xs = p->xs;
    const int k = d.k, ld = d.ld;
    const bool fresh = p->fresh;
    if (fresh) {
        HIP_TRY(hipEventRecord(p->start, cs));
        HIP_TRY(hipStreamWaitEvent(xs, p->start, 0));
    }
    p->fresh = false;
    // item pass (the E rows it reads were written by the apply kernel of the last iteration, on this very stream), the
    // reduce-scatter of each range on the exchange stream
    for (int j = 0; j < d.nranges; j++) {
        const hpf_shard_range &r = d.ranges[j];
        if (r.nseg > 0)
            HPF_TRY(hpf_hip_sweep_f32(d.i_segs + r.seg_lo, r.nseg, d.i_idx, d.i_y, d.eB, eT,
                                      d.part_i + (size_t)r.seg_lo * ld, d.acc_i, k, k, ld, r.short_rows,
                                      d.item_sweep_grid, nullptr, (void *)cs));
        if (r.nmulti > 0)
            HPF_TRY(hpf_hip_segsum_f32(d.part_i, d.i_row_seg_ptr, r.multi_rows, r.nmulti, d.acc_i, ld, k, 1, (void *)cs));
        HIP_TRY(hipEventRecord(p->sw_done[j], cs));
        HIP_TRY(hipStreamWaitEvent(xs, p->sw_done[j], 0));
        HPF_TRY(reduce_scatter_range(p, j, xs));
    }
    // exchange stream, under the user side: the shape / psi half of the finalizer for this rank's slices, then ONE
    // all-gather of the [numerators | base rate] rows of all ranges
    if (p->nfin > 0)
        HPF_TRY(hpf_hip_item_shape_rows_f32(d.acc_own, p->nfin, p->fin_rows, p->fin_acc, p->fin_row0, d.eB, d.shp_own,
                                            d.e_own, d.t_rte, d.t_rte_prev, d.c, d.t_shp, k, ld, d.csB_part_rows,
                                            (void *)xs));
    HPF_TRY(collective(p, HPF_COLL_ALL_GATHER, d.e_own, d.ag_recv, p->total * d.e_own_ld, xs));
    HIP_TRY(hipEventRecord(p->ag_done[0], xs));
    // user side on the compute stream; colsum(Beta) of the last iteration was summed on the exchange stream
    if (!fresh) HIP_TRY(hipStreamWaitEvent(cs, p->csB_done, 0));
    if (store) HIP_TRY(hipMemcpyAsync(d.csB_used, d.csB, (size_t)ld * sizeof(float), hipMemcpyDeviceToDevice, cs));
    float *shp = store ? d.Gamma_shp : nullptr, *fac = store ? d.Theta : nullptr;
    if (d.u_nseg > 0)
        HPF_TRY(hpf_hip_sweep_finalize_f32(d.u_segs, d.u_nseg, d.u_idx, d.u_y, eT, d.eB, d.part_u, eT_next, shp, nullptr, fac,
                                       d.k_rte, d.k_rte_prev, d.csB, d.csT_part, d.a, d.k_shp, d.add_k_rte, k, ld,
                                       d.user_sweep_grid, (void *)cs));
    if (d.u_nmulti > 0)      // (its column-sum partial rows stay zero otherwise: never written)
        HPF_TRY(hpf_hip_row_finalize_f32(d.part_u, d.u_row_seg_ptr, d.u_multi_rows, d.u_nmulti, eT, eT_next, shp, nullptr, fac,
                                     d.k_rte, d.k_rte_prev, d.csB, d.csT_part + (size_t)d.user_sweep_grid * ld, d.a,
                                     d.k_shp, d.add_k_rte, k, ld, ld, d.user_multi_grid, (void *)cs));
    HPF_TRY(hpf_hip_colsum_reduce_f32(d.csT_part, d.csT_part_rows, d.csT, ld, (void *)cs));
    HPF_TRY(collective(p, HPF_COLL_ALL_REDUCE, d.csT, d.csT, ld, cs));      // (on the compute stream: no hand-over)
    // the rates applied to ALL items locally, from the gathered rows
    HIP_TRY(hipStreamWaitEvent(cs, p->ag_done[0], 0));
    HPF_TRY(hpf_hip_item_apply_rows_f32(d.ag_recv, d.shp_own, d.eB, store ? d.Lambda_shp : nullptr,
                                        store ? d.Beta : nullptr, d.t_rte, d.csT, d.csB_part, d.add_t_rte, k, ld, d.rank,
                                        d.world, d.nI, d.nranges, p->lo, p->hi, d.csB_part_rows, (void *)cs));
    HIP_TRY(hipEventRecord(p->p2_done, cs));
    // colsum(Beta): this rank's partial, summed over ranks on the exchange stream (its reader is the next user side)
    HIP_TRY(hipStreamWaitEvent(xs, p->p2_done, 0));
    HPF_TRY(hpf_hip_colsum_reduce_f32(d.csB_part, d.csB_part_rows, d.csB, ld, (void *)xs));
    HPF_TRY(collective(p, HPF_COLL_ALL_REDUCE, d.csB, d.csB, ld, xs));
    HIP_TRY(hipEventRecord(p->csB_done, xs));
    return 0;
}

// The direct schedule (include/hpf_hip.h, HPF_SCHEDULE_DIRECT).  Per iteration, epoch e:
//   compute stream:  sweep(range 0) . [split rows] . sweep(range 1; on entry: SWEPT(0) = e) . [split rows] . ... .
//                    user sweep + finalize (on entry: SWEPT(last) = e) . split user rows .
//                    colsum(Theta) over ALL ranks (this small launch then waits for GATHERED / the owners' SHAPED) .
//                    apply . colsum(Beta) over all ranks
//   exchange stream: per range j: a one-wave wait for every rank's SWEPT(j) = e (this rank's too: NO stream event ties the
//                    two streams in steady state), the slice pulled out of the N buffers and summed;  the shape half of
//                    all slices;  ONE pull of every owner's finished rows (on entry: SHAPED = e; last block: GATHERED = e)
//                    -- or, without prefetch, a launch that only raises SHAPED.
// Why no buffer is overwritten while a peer still reads it: a rank's sweep of epoch e+1 follows its apply of e, which has
// seen every owner's SHAPED(e), raised after that owner's pulls of e; a rank's pulls and shape half of e+1 wait for every
// rank's SWEPT(e+1), raised after that rank's apply of e (which read the finished rows of e, directly or through its pull).
static int iterate_direct(Plan *p, const float *eT, float *eT_next, int store, hipStream_t cs) {
    const hpf_shard_desc &d = p->d;
    hipStream_t xs = p->xs;
    const int k = d.k, ld = d.ld;
    if (p->tracing) {
        p->epoch++;
    } else {
        HPF_TRY(hpf_hip_p2p_region_next_epoch(d.p2p_region, &p->epoch));
    }
    if (p->fresh) {      // (first iteration after a join: the exchange stream orders itself after whatever the caller queued)
        HIP_TRY(hipEventRecord(p->start, cs));
        HIP_TRY(hipStreamWaitEvent(xs, p->start, 0));
    }
    p->fresh = false;
    for (int j = 0; j < d.nranges; j++) {
        const hpf_shard_range &r = d.ranges[j];
        // (a range without local nonzeros still launches when it carries the previous range's flag)
        if (r.nseg > 0 || j > 0) HPF_TRY(direct_item_sweep(p, j, eT, j > 0 ? HPF_P2P_FLAG_SWEPT(j - 1) : -1, cs));
        if (r.nmulti > 0)
            HPF_TRY(hpf_hip_segsum_f32(d.part_i, d.i_row_seg_ptr, r.multi_rows, r.nmulti, d.acc_i, ld, k, 1, (void *)cs));
        HPF_TRY(direct_pull_reduce(p, j, xs));
    }
    // exchange stream, under the user side: the shape half of this rank's slices; SHAPED; every owner's rows -> ag_recv
    if (p->nfin > 0)
        HPF_TRY(hpf_hip_item_shape_rows_f32(d.acc_own, p->nfin, p->fin_rows, p->fin_acc, p->fin_row0, d.eB, d.shp_own,
                                            d.e_own, d.t_rte, d.t_rte_prev, d.c, d.t_shp, k, ld, d.csB_part_rows,
                                            (void *)xs));
    const double gather_bytes = (double)p->total * d.e_own_ld * 4.0 * (d.world - 1);
    if (d.direct_prefetch) {
        if (g_tr) {
            g_tr->add(HPF_TRACE_KERNEL, HPF_TRACE_K_GATHER_PULL, xs, HPF_P2P_FLAG_SHAPED(0) | (HPF_P2P_FLAG_GATHERED << 8));
        } else {
            HPF_TRY(hpf_p2p::gather_pull(d.p2p_region, d.p2p_send_offset, d.ag_recv, p->total * d.e_own_ld,
                                         HPF_P2P_FLAG_SHAPED(0), HPF_P2P_FLAG_GATHERED, p->epoch, d.direct_gather_gx, 0, xs));
            HPF_TRY(pull_link_time(p, gather_bytes, xs));
        }
    } else if (g_tr) {
        g_tr->add(HPF_TRACE_KERNEL, HPF_TRACE_K_SIGNAL, xs, HPF_P2P_FLAG_SHAPED(0));
    } else {
        HPF_TRY(hpf_hip_p2p_signal(d.p2p_region, HPF_P2P_FLAG_SHAPED(0), p->epoch, (void *)xs));
    }
    HIP_TRY(hipEventRecord(p->ag_done[0], xs));          // (what hpf_hip_shard_join waits for)
    // user side; the launch that follows the last item sweep raises its flag
    const int last_flag = HPF_P2P_FLAG_SWEPT(d.nranges - 1);
    float *shp = store ? d.Gamma_shp : nullptr, *fac = store ? d.Theta : nullptr;
    if (d.u_nseg > 0) {
        if (g_tr) {
            g_tr->add(HPF_TRACE_KERNEL, HPF_TRACE_K_SWEEP_FINALIZE, cs, 1 + last_flag);
        } else {
            HPF_TRY(hpf_direct::sweep_finalize(d.u_segs, d.u_nseg, d.u_idx, d.u_y, eT, d.eB, d.part_u, eT_next, shp, nullptr,
                                               fac, d.k_rte, d.k_rte_prev, d.csB, d.csT_part, d.a, d.k_shp, d.add_k_rte, k,
                                               ld, d.user_sweep_grid, store ? d.csB_used : nullptr,
                                               raise(p, last_flag), cs));
        }
    } else {
        if (g_tr) {
            g_tr->add(HPF_TRACE_KERNEL, HPF_TRACE_K_SIGNAL, cs, last_flag);
        } else {
            HPF_TRY(hpf_hip_p2p_signal(d.p2p_region, last_flag, p->epoch, (void *)cs));
            if (store) HIP_TRY(hipMemcpyAsync(d.csB_used, d.csB, (size_t)ld * sizeof(float), hipMemcpyDeviceToDevice, cs));
        }
    }
    if (d.u_nmulti > 0)      // (its column-sum partial rows stay zero otherwise: never written)
        HPF_TRY(hpf_hip_row_finalize_f32(d.part_u, d.u_row_seg_ptr, d.u_multi_rows, d.u_nmulti, eT, eT_next, shp, nullptr,
                                         fac, d.k_rte, d.k_rte_prev, d.csB, d.csT_part + (size_t)d.user_sweep_grid * ld,
                                         d.a, d.k_shp, d.add_k_rte, k, ld, ld, d.user_multi_grid, (void *)cs));
    // colsum(Theta) over all ranks; the same small launch waits for what the apply's large grid needs
    const int apply_kind = d.direct_prefetch ? HPF_P2P_FLAG_GATHERED : HPF_P2P_FLAG_SHAPED(0);
    HPF_TRY(direct_colsum_allreduce(p, d.csT_part, d.csT_part_rows, d.csT, HPF_P2P_VEC_CST,
                                    d.direct_prefetch ? 0u : (1u << apply_kind), apply_kind, cs));
    // the rates applied to ALL items, from the gathered rows (or straight from the owners' buffers)
    if (g_tr) {      // (arg: 1 + the flag kind its own workgroups wait for; 0: none -- the colsum launch has waited)
        g_tr->add(HPF_TRACE_KERNEL, HPF_TRACE_K_ITEM_APPLY, cs, d.direct_prefetch ? 0 : 1 + apply_kind);
    } else {
        const float *blocks[HPF_P2P_MAX_RANKS];
        for (int q = 0; q < d.world; q++)
            blocks[q] = (d.direct_prefetch && q != d.rank)       // (own rows: where the shape half left them)
                            ? d.ag_recv + (size_t)q * p->total * d.e_own_ld
                            : reinterpret_cast<const float *>(reinterpret_cast<const char *>(p->peer_data[q]) +
                                                              d.p2p_send_offset);
        // prefetch: everything the apply reads is LOCAL memory, complete before the colsum launch ahead of it on this stream
        // saw GATHERED -- the launch boundary is the acquire, no workgroup polls or fences (a system-scope acquire in each
        // of its 2048 workgroups cost the kernel 8 of its 55 us at C3 x 8)
        const int in_kernel_wait = d.direct_prefetch ? -1 : apply_kind;
        HPF_TRY(hpf_direct::item_apply_blocks(blocks, d.world, in_kernel_wait, d.direct_prefetch ? 1 : 0, p->epoch, p->pp,
                                              d.shp_own, d.eB, store ? d.Lambda_shp : nullptr, store ? d.Beta : nullptr,
                                              d.t_rte, d.csT, d.csB_part, d.add_t_rte, k, ld, d.rank, d.world, d.nI,
                                              d.nranges, p->lo, p->hi, d.csB_part_rows, cs));
        if (!d.direct_prefetch) HPF_TRY(pull_link_time(p, gather_bytes, cs));
    }
    HPF_TRY(direct_colsum_allreduce(p, d.csB_part, d.csB_part_rows, d.csB, HPF_P2P_VEC_CSB, 0u, -1, cs));
    return 0;
}

int hpf_hip_shard_status(void *plan) {
    if (!plan) return HPF_EINVAL;
    Plan *p = (Plan *)plan;
    if (p->tracing || p->d.schedule != HPF_SCHEDULE_DIRECT) return 0;
    uint32_t err = 0;
    return hpf_hip_p2p_region_status(p->d.p2p_region, &err);
}

int hpf_hip_shard_iterate(void *plan, const float *eT, float *eT_next, int store, void *compute_stream) {
    if (!plan) return HPF_EINVAL;
    Plan *p = (Plan *)plan;
    if (p->d.nU > 0 && (!eT || !eT_next || eT == eT_next)) return HPF_EINVAL;
    TraceScope scope(p->tracing ? &p->tracer : nullptr);
    if (p->d.schedule == HPF_SCHEDULE_DIRECT) return iterate_direct(p, eT, eT_next, store, (hipStream_t)compute_stream);
    if (p->d.schedule == HPF_SCHEDULE_GATHER_EARLY)
        return iterate_gather_early(p, eT, eT_next, store, (hipStream_t)compute_stream);
    const hpf_shard_desc &d = p->d;
    hipStream_t cs = (hipStream_t)compute_stream, xs = p->xs;
    const int k = d.k, ld = d.ld;
    const bool fresh = p->fresh;
    if (fresh) {   // first iteration after a join: the exchange stream orders itself after whatever the caller queued
        HIP_TRY(hipEventRecord(p->start, cs));
        HIP_TRY(hipStreamWaitEvent(xs, p->start, 0));
    }
    p->fresh = false;
    // item pass, range by range: sweep this rank's CSC slice into the packed exchange buffer (split / empty rows via
    // part[] + segsum), then reduce-scatter the range on the exchange stream
    for (int j = 0; j < d.nranges; j++) {
        const hpf_shard_range &r = d.ranges[j];
        if (!fresh) HIP_TRY(hipStreamWaitEvent(cs, p->ag_done[j], 0));   // the range's E rows of the last iteration
        if (r.nseg > 0)
            HPF_TRY(hpf_hip_sweep_f32(d.i_segs + r.seg_lo, r.nseg, d.i_idx, d.i_y, d.eB, eT,
                                      d.part_i + (size_t)r.seg_lo * ld, d.acc_i, k, k, ld, r.short_rows,
                                      d.item_sweep_grid, nullptr, (void *)cs));
        if (r.nmulti > 0)
            HPF_TRY(hpf_hip_segsum_f32(d.part_i, d.i_row_seg_ptr, r.multi_rows, r.nmulti, d.acc_i, ld, k, 1, (void *)cs));
        HIP_TRY(hipEventRecord(p->sw_done[j], cs));
        HIP_TRY(hipStreamWaitEvent(xs, p->sw_done[j], 0));
        HPF_TRY(reduce_scatter_range(p, j, xs));
    }
    // user side under the exchange: fused sweep + finalizer, the split / empty rows, colsum(Theta) of this rank
    if (store) HIP_TRY(hipMemcpyAsync(d.csB_used, d.csB, (size_t)ld * sizeof(float), hipMemcpyDeviceToDevice, cs));
    float *shp = store ? d.Gamma_shp : nullptr, *fac = store ? d.Theta : nullptr;
    if (d.u_nseg > 0)
        HPF_TRY(hpf_hip_sweep_finalize_f32(d.u_segs, d.u_nseg, d.u_idx, d.u_y, eT, d.eB, d.part_u, eT_next, shp, nullptr, fac,
                                       d.k_rte, d.k_rte_prev, d.csB, d.csT_part, d.a, d.k_shp, d.add_k_rte, k, ld,
                                       d.user_sweep_grid, (void *)cs));
    if (d.u_nmulti > 0)      // (its column-sum partial rows stay zero otherwise: never written)
        HPF_TRY(hpf_hip_row_finalize_f32(d.part_u, d.u_row_seg_ptr, d.u_multi_rows, d.u_nmulti, eT, eT_next, shp, nullptr, fac,
                                     d.k_rte, d.k_rte_prev, d.csB, d.csT_part + (size_t)d.user_sweep_grid * ld, d.a,
                                     d.k_shp, d.add_k_rte, k, ld, ld, d.user_multi_grid, (void *)cs));
    HPF_TRY(hpf_hip_colsum_reduce_f32(d.csT_part, d.csT_part_rows, d.csT, ld, (void *)cs));
    HIP_TRY(hipEventRecord(p->csT_ready, cs));
    HIP_TRY(hipStreamWaitEvent(xs, p->csT_ready, 0));
    // one in-order chain on the exchange stream: colsum(Theta) summed over ranks, the finalizer of this rank's slices
    // of all ranges, the all-gathers of the new E rows, colsum(Beta) summed over ranks ahead of the last all-gather
    HPF_TRY(collective(p, HPF_COLL_ALL_REDUCE, d.csT, d.csT, ld, xs));
    if (p->nfin > 0)
        HPF_TRY(hpf_hip_row_finalize_ranges_f32(d.acc_own, p->nfin, p->fin_rows, p->fin_acc, p->fin_row0, d.eB, d.e_own,
                                                store ? d.Lambda_shp : nullptr, nullptr, store ? d.Beta : nullptr,
                                                d.t_rte, d.t_rte_prev, d.csT, d.csB_part, d.c, d.t_shp, d.add_t_rte, k,
                                                ld, k, d.e_own_ld, d.csB_part_rows, (void *)xs));
    for (int j = 0; j < d.nranges; j++) {
        if (j == d.nranges - 1) {
            HPF_TRY(hpf_hip_colsum_reduce_f32(d.csB_part, d.csB_part_rows, d.csB, ld, (void *)xs));
            HPF_TRY(collective(p, HPF_COLL_ALL_REDUCE, d.csB, d.csB, ld, xs));
        }
        HPF_TRY(all_gather_range(p, j, xs));
        HIP_TRY(hipEventRecord(p->ag_done[j], xs));
    }
    return 0;
}

int hpf_hip_shard_exchange_only(void *plan, int op, int range, void *stream) {
    if (!plan) return HPF_EINVAL;
    Plan *p = (Plan *)plan;
    TraceScope scope(p->tracing ? &p->tracer : nullptr);
    const hpf_shard_desc &d = p->d;
    hipStream_t st = (hipStream_t)stream;
    if (d.schedule == HPF_SCHEDULE_DIRECT) {
        // the pulls on their own, every rank in step: a fresh epoch, this rank's flag, then the kernel that carries the
        // exchange (scratch results: the state is not meaningful afterwards)
        if (range >= d.nranges) return HPF_EINVAL;
        if (p->tracing) {
            p->epoch++;
        } else {
            HPF_TRY(hpf_hip_p2p_region_next_epoch(d.p2p_region, &p->epoch));
        }
        if (range < 0) return direct_colsum_allreduce(p, d.csT_part, d.csT_part_rows, d.csT, HPF_P2P_VEC_CST, 0u, -1, st);
        if (op == HPF_COLL_REDUCE_SCATTER) {
            if (g_tr) {
                g_tr->add(HPF_TRACE_KERNEL, HPF_TRACE_K_SIGNAL, st, HPF_P2P_FLAG_SWEPT(range));
            } else {
                HPF_TRY(hpf_hip_p2p_signal(d.p2p_region, HPF_P2P_FLAG_SWEPT(range), p->epoch, (void *)st));
            }
            return direct_pull_reduce(p, range, st);
        }
        if (op == HPF_COLL_ALL_GATHER) {
            if (range != 0) return 0;         // (one pull for all ranges: counted with range 0)
            if (g_tr) {
                g_tr->add(HPF_TRACE_KERNEL, HPF_TRACE_K_GATHER_PULL, st, HPF_P2P_FLAG_SHAPED(0) | (HPF_P2P_FLAG_GATHERED << 8));
                return 0;
            }
            if (!d.ag_recv) return HPF_EINVAL;
            return hpf_p2p::gather_pull(d.p2p_region, d.p2p_send_offset, d.ag_recv, p->total * d.e_own_ld,
                                        HPF_P2P_FLAG_SHAPED(0), HPF_P2P_FLAG_GATHERED, p->epoch, d.direct_gather_gx, 1, st);
        }
        return HPF_EINVAL;
    }
    if (range < 0) {
        if (op != HPF_COLL_ALL_REDUCE) return HPF_EINVAL;
        // scratch between iterations: the reduce-scatter outputs are rewritten before the finalizer reads them
        return collective(p, HPF_COLL_ALL_REDUCE, d.acc_own, d.acc_own, d.ld, st, true);
    }
    if (range >= d.nranges) return HPF_EINVAL;
    if (op == HPF_COLL_REDUCE_SCATTER) return reduce_scatter_range(p, range, st);
    if (op == HPF_COLL_ALL_GATHER) {
        if (d.schedule == HPF_SCHEDULE_GATHER_EARLY)     // one collective for all ranges: counted with range 0
            return range == 0 ? collective(p, HPF_COLL_ALL_GATHER, d.e_own, d.ag_recv, p->total * d.e_own_ld, st) : 0;
        return all_gather_range(p, range, st);   // (idempotent: e_own still holds the rows)
    }
    return HPF_EINVAL;
}

int hpf_hip_shard_trace(void *plan, hpf_shard_trace_rec *out, int64_t cap, int64_t *n) {
    if (!plan || !n || cap < 0 || (cap > 0 && !out)) return HPF_EINVAL;
    Plan *p = (Plan *)plan;
    if (!p->tracing) return HPF_EINVAL;
    const int64_t have = (int64_t)p->tracer.recs.size();
    *n = have;
    if (cap < have) return 0;         // (size query: nothing is consumed)
    for (int64_t i = 0; i < have; i++) out[i] = p->tracer.recs[(size_t)i];
    p->tracer.recs.clear();
    return 0;
}

}  // extern "C"
