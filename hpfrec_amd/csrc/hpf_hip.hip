// hpf_hip.hip -- gfx950 (MI355X, CDNA4) kernels + C ABI for the HPF full-batch CAVI sweep.
//
// What is replaced (reference: /root/reference/hpfrec/cython_loops.pxi = "PXI"):
//   sweep_kernel        <- update_phi PXI:551-591 fused with update_G_n_L_sh PXI:613-621; MODE 1 also
//                          runs the row finalizer for whole-row segments (epilogue); MODE 2 / 3: the two sides of a
//                          stochastic step (PXI:292-325, 344-377, 438-473) fused into the sweeps that form their phi-sums
//   row_finalize_kernel <- numpy rate/shape statements of fit_hpf PXI:236-259 + the psi/log/exp
//                          hoisted out of update_phi (PXI:588: they only depend on the row)
//   llk_sweep_kernel, pair_llk_kernel <- llk_plus_rmse PXI:627-658, sum_prediction PXI:816-825
//   pair_dot_kernel     <- predict_multiple PXI:803-810;  score_rows_kernel <- HPF.topN's GEMV
//   svi_*_kernel        <- the numpy statements of an SVI batch / partial_fit PXI:300-325,352-377,443-473
//   colsum_sequential_kernel <- Theta.sum(axis=0) / Beta.sum(axis=0) in numpy's own order (HPF_COLSUM_ORDER=reference)
//
// Design: the reference evaluates, per nonzero and factor,
//   exp(psi(Gs_uk) - log(Gr_uk) + psi(Ls_ik) - log(Lr_ik)) = eT_uk * eB_ik
// with eT/eB depending on one row only.  We keep eT (users) and eB (items) as padded fp32
// tables and never materialise phi: per nonzero s = <eT_u, eB_i>, w = y/s and
//   Gamma_shp_u = a + eT_u (*) sum_i w eB_i        (CSR pass, rows = users)
//   Lambda_shp_i = c + eB_i (*) sum_u w eT_u       (CSC pass, rows = items)
// Both passes are the same kernel: HBM/L2-bound gathers of 4*ld-byte rows, no atomics,
// bit-reproducible.  Wave64 layout: a row of ld floats is held by LPR = ld/4 lanes as one
// float4 each (VPL float4 per lane when ld > 256), so a wavefront processes 64/LPR nonzeros
// per step; the k-reduction is 4 DPP adds inside a 16-lane DPP row.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "hpf_hip.h"
#include "hpf_internal.h"
#include "hpf_p2p_dev.h"

namespace {

constexpr int WAVE = 64;
constexpr int BLOCK = 256;
constexpr int WPB = BLOCK / WAVE;

// tuning knobs (compile-time; defaults are the measured best)
#ifndef HPF_U
#define HPF_U 8  // gathers in flight per wavefront (U=4: -3%, U=2: -11%, U=16: -12% at C3)
#endif
#ifndef HPF_NT
#define HPF_NT 0  // 1: non-temporal hints on the streamed operands (idx, y, part)
#endif
#ifndef HPF_SWEEP_WAVES_PER_EU
#define HPF_SWEEP_WAVES_PER_EU 1
#endif
#ifndef HPF_FUSED_MAX_WAVES
// occupancy cap (waves per SIMD) of the sweep with a fused row finalizer: 4 x 8 gathers in flight per SIMD is the
// measured optimum at C3; a 5th wave (which the register count would allow) costs 1.4 % through L2 thrash
#define HPF_FUSED_MAX_WAVES 4
#endif

template <typename T>
__device__ __forceinline__ void stream_store(T *p, T v) {
#if HPF_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

template <typename T>
__device__ __forceinline__ T stream_load(const T *p) {
#if HPF_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}

// one 16-byte load (global_load_dwordx4).  Written as a vector-typed load on purpose: `cond ? p[i] : zero4` on HIP's float4
// struct is lowered to four dword loads per lane (svi_side_kernel ran its whole-table passes that way until round 5).
typedef float hpf_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4(const float4 *p) {
    const hpf_v4f t = *reinterpret_cast<const hpf_v4f *>(p);
    return make_float4(t.x, t.y, t.z, t.w);
}

// ----------------------------------------------------------------------------------------
// cross-lane helpers
// ----------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// all-reduce (sum) over aligned groups of W lanes; every lane of the group gets the total
template <int W>
__device__ __forceinline__ float group_sum(float v) {
    v += dpp_f<0xB1>(v);                         // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);                         // quad_perm [2,3,0,1]
    if constexpr (W >= 8) v += dpp_f<0x141>(v);  // row_half_mirror
    if constexpr (W >= 16) v += dpp_f<0x140>(v); // row_mirror
    if constexpr (W >= 32) v += __shfl_xor(v, 16);
    if constexpr (W >= 64) v += __shfl_xor(v, 32);
    return v;
}

__device__ __forceinline__ float wave_sum(float v) { return group_sum<64>(v); }

// Row scale of the E tables.  Any positive per-row scale cancels in w*e; it only has to keep
// fp32 in range.  Scale by the power of two that brings the row's largest entry into [1,2): exact (no rounding
// besides the one fp64->fp32 conversion), and it costs an integer max over the high words (positive doubles order
// like their bit patterns) instead of an fp64 max tree plus an fp64 divide.  `hi` = this lane's largest high word
// (0 for lanes without a valid column); returns 2^-floor(log2(row max)) (2^1023 if the whole row underflowed).
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}

__device__ __forceinline__ double row_pow2_scale(int hi) {
    hi = max(hi, dpp_i<0xB1>(hi));
    hi = max(hi, dpp_i<0x4E>(hi));
    hi = max(hi, dpp_i<0x141>(hi));
    hi = max(hi, dpp_i<0x140>(hi));
    hi = max(hi, __shfl_xor(hi, 16));
    hi = max(hi, __shfl_xor(hi, 32));
    return __hiloint2double((2046 - (hi >> 20)) << 20, 0);
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int m = 1; m < WAVE; m <<= 1) v += __shfl_xor(v, m);
    return v;
}

__device__ __forceinline__ float dot4(const float4 a, const float4 b) {
    return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)));
}

// ----------------------------------------------------------------------------------------
// exp(psi(x)) / r for x > 0, r > 0, in double, rounded once by the caller.
//   psi(x) = psi(x+6) - sum_{i<6} 1/(x+i)            (recurrence, applied for every x: no divergence)
//   psi(s) = log s - 1/(2s) - sum_n B_2n/(2n s^2n)   (asymptotic, s >= 6: truncation < 2e-13)
// so exp(psi(x))/r = (s/r) * exp(-(1/(2s) + series + recurrence sum)) with no log at all.
// Same series as the Cephes psi the reference calls through scipy (PXI:5,588).
// Cost matters (it runs in the sweep's epilogue and, replicated, in the multi-GPU item finalizer):
// one shared reciprocal for 1/s, the recurrence sum and 1/rte, Newton steps instead of an IEEE divide,
// and a short exp (fdlibm-style ln2 split + degree-9 Taylor, rel. error < 1e-11) instead of ocml's.
// ----------------------------------------------------------------------------------------
// The blends of a stochastic step, a*x + b*y, with each product and the sum rounded on its own -- numpy's float32 statements
// (PXI:316-325, 368-377) -- and NOT contracted into an fma: a contraction is the compiler's choice per kernel, and the same
// statement is formed by several kernels (the whole-table passes, the row-list kernels, the sweep's fused epilogue) whose
// results must agree bit for bit.  (`#pragma clang fp contract(off)`: HIP's __fmul_rn / __fadd_rn are plain operators, which
// -ffp-contract=fast still fuses -- the whole-table pass did, the sweep's epilogue did not: one ulp apart in 3 rows of 4.)
__device__ __forceinline__ float blend2(float a, float x, float b, float y) {
#pragma clang fp contract(off)
    const float p = a * x;
    const float q = b * y;
    return p + q;
}
// shp = w_new*(prior + e*acc) [+ w_old*shp]: `fresh` = fmaf(e, acc, prior) at every site
__device__ __forceinline__ float blend_shape(float w_new, float fresh, float w_old, float old) {
#pragma clang fp contract(off)
    if (w_old == 0.f) return w_new * fresh;
    return blend2(w_new, fresh, w_old, old);
}
// step*(base + c) + step_prev*old (rates: base = top/rs, c = a column sum; row scalars: base = add, c = sum_k fac)
__device__ __forceinline__ float blend_rate(float step, float base, float c, float step_prev, float old) {
#pragma clang fp contract(off)
    const float t = base + c;
    return blend2(step, t, step_prev, old);
}

// a / b for TWO independent pairs, correctly rounded, for operands and quotients in the normal range: the compiler's IEEE
// expansion of a float32 division (v_div_scale x2, v_rcp, the Newton / residual FMAs, v_div_fmas, v_div_fixup) minus its range
// handling -- the scale factors are 1 and the fix-up returns the quotient unchanged unless an operand or the quotient is
// denormal, infinite or NaN -- with the seven FMA-type steps of the two divisions issued as packed instructions
// (v_pk_fma_f32 / v_pk_mul_f32: two lanes' worth per issue).  The same roundings, so the same bits as `a / b`
// (tests/test_hip_parity.py::test_lazy_batch_side_kernel_is_the_general_one_bit_for_bit holds the two against each other);
// a streaming pass over a [rows][256] table does 200 divisions per 832-byte row and was bound by their issue rate, not by
// memory.  Shapes are >= the prior (0.3) and rates >= a column sum: normal range by construction.
typedef float hpf_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ hpf_v2f div2_normal(hpf_v2f a, hpf_v2f b) {
    hpf_v2f r;
    r.x = __builtin_amdgcn_rcpf(b.x);
    r.y = __builtin_amdgcn_rcpf(b.y);
    const hpf_v2f one = {1.f, 1.f};
    hpf_v2f e = __builtin_elementwise_fma(-b, r, one);
    r = __builtin_elementwise_fma(e, r, r);
    hpf_v2f q;
    {
#pragma clang fp contract(off)
        q = a * r;
    }
    e = __builtin_elementwise_fma(-b, q, a);
    q = __builtin_elementwise_fma(e, r, q);
    e = __builtin_elementwise_fma(-b, q, a);
    q = __builtin_elementwise_fma(e, r, q);
    return q;
}

// fc[t] = valid ? sh[t] / rt[t] : 0 for a lane's NC columns, in pairs through div2_normal where NC is even
template <int NC>
__device__ __forceinline__ void div_columns(const float (&sh)[NC], const float (&rt)[NC], const bool (&valid)[NC],
                                            float (&fc)[NC]) {
    if constexpr (NC % 2 == 0) {
#pragma unroll
        for (int t = 0; t < NC; t += 2) {
            const hpf_v2f q = div2_normal(hpf_v2f{sh[t], sh[t + 1]}, hpf_v2f{rt[t], rt[t + 1]});
            fc[t] = valid[t] ? q.x : 0.f;
            fc[t + 1] = valid[t + 1] ? q.y : 0.f;
        }
    } else {
#pragma unroll
        for (int t = 0; t < NC; t++) fc[t] = valid[t] ? sh[t] / rt[t] : 0.f;
    }
}

__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);  // v_rcp_f64: ~24 good bits
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

__device__ __forceinline__ double exp_neg(double v) {  // exp(-v) for v >= 0
    const double t = rint(v * 1.4426950408889634);
    double f = fma(t, -6.93147180369123816490e-01, v);  // t*ln2_hi is exact for |t| < 2^11
    f = fma(t, -1.90821492927058770002e-10, f);
    const double g = -f;                                 // |g| <= 0.3466
    double p = 1.0 / 362880.0;
    p = fma(p, g, 1.0 / 40320.0);
    p = fma(p, g, 1.0 / 5040.0);
    p = fma(p, g, 1.0 / 720.0);
    p = fma(p, g, 1.0 / 120.0);
    p = fma(p, g, 1.0 / 24.0);
    p = fma(p, g, 1.0 / 6.0);
    p = fma(p, g, 0.5);
    p = fma(p, g, 1.0);
    p = fma(p, g, 1.0);
    return ldexp(p, -(int)t);
}

__device__ __forceinline__ double expect_ratio(float shp, float rte) {
    const double x = (double)shp;
    const double p01 = x * (x + 1.0), p23 = (x + 2.0) * (x + 3.0), p45 = (x + 4.0) * (x + 5.0);
    const double n01 = 2.0 * x + 1.0, n23 = 2.0 * x + 5.0, n45 = 2.0 * x + 9.0;
    const double den = p01 * p23 * p45;                                   // prod_{i<6} (x+i)
    const double num = fma(n01, p23 * p45, p01 * fma(n23, p45, n45 * p23));  // den * sum_{i<6} 1/(x+i)
    const double s = x + 6.0;
    const double rd = (double)rte;
    const double ds = den * s;
    const double R = fast_rcp(ds * rd);  // ONE reciprocal serves 1/s, the recurrence sum and 1/rte
    const double Rr = R * rd;            // 1/(den*s)
    const double r = Rr * den;           // 1/s
    const double w = num * (Rr * s);     // sum_{i<6} 1/(x+i)
    const double z = r * r;
    double poly = 8.33333333333333333333E-2;
    poly = fma(poly, z, -2.10927960927960927961E-2);
    poly = fma(poly, z, 7.57575757575757575758E-3);
    poly = fma(poly, z, -4.16666666666666666667E-3);
    poly = fma(poly, z, 3.96825396825396825397E-3);
    poly = fma(poly, z, -8.33333333333333333333E-3);
    poly = fma(poly, z, 8.33333333333333333333E-2);
    const double v = fma(poly, z, 0.5 * r) + w;
    return (s * exp_neg(v)) * (R * ds);
}

// ----------------------------------------------------------------------------------------
// sweep: one wavefront per segment
// ----------------------------------------------------------------------------------------
struct FinalizeArgs {  // the row-finalize operands when it is fused into the sweep (MODE 1)
    const float *cs_other;
    float *cs_partial, *e_new, *shp, *rte, *fac, *rs;
    float prior_shp, top_shp, add_rte;
    int k;
    float *rs_prev;   // optional: receives the row's OLD scalar rate (rte = top/rs_prev + cs_other is rank-1,
                      // so callers may keep this instead of the [rows][ld] rte table)
    float *acc_rows;  // MODE 0: packed [rows][acc_ld] accumulator rows of whole-row segments (or null)
    int acc_ld;
    const int64_t *nseg_dev;  // optional: the live segment count on the device (<= nseg; stochastic batches)
    // direct exchange (hpf_p2p_dev.h): on entry, block 0 tells every peer that what the PREVIOUS launches of this stream
    // wrote into this rank's exchange buffer is complete -- flags[sig_kind][rank] = sig_epoch (null: nothing)
    const hpf_p2p::Peers *sig_peers;
    int sig_kind;
    uint32_t sig_epoch;
    float *cs_other_copy;     // optional: block 0 keeps a copy of cs_other (the column sums this launch used)
    int nq4;                  // SKIP: float4s of a gathered row that hold columns < k, rounded up to whole 64-byte sectors
    float w_new, w_old, step, step_prev;   // MODE 2 / 3: the blend weights of a stochastic step (shapes; rates and row scalars)
    // MODE 3: the rate a batch row's E row is formed with in the PROLOGUE -- factored (rate_rs != null):
    // rate_top / rate_rs[row] + rate_cs[c], else the stored table rte_in[row][c] (may be the table `rte` points at)
    const float *rate_rs, *rate_cs, *rte_in;
    float rate_top;
};

// MODE: 0 = plain sweep; 1 = row finalize fused as EPILOGUE of whole-row segments; 2 = the OTHER side of a stochastic step
// fused the same way (svi_side_kernel's rate_mode 1 statements for the rows the batch touches: shapes and rates blended
// towards the step's estimate, means, the row scalar, the new E row): the rows a batch touches on its other side are short
// (8-12 nonzeros at BASELINE config C5), so writing their phi-sums to memory, reading them back next to the E row the sweep
// already holds, and only then updating the row was as much traffic as the gathers themselves.
// MODE 3 = the BATCH side of a stochastic step, both ends of the row's life in the wavefront that sweeps it.  PROLOGUE: the
// row's E row exp(psi(shp))/rte is formed from its CURRENT shape and rate (expect_kernel's statements: the psi/log/exp hoisted
// out of update_phi_csr, PXI:683-692) in registers and stored for the other side's sweep of the same step -- the loads of a
// wave's NEXT segment are requested one segment ahead, the fp64 work runs under the other waves' gathers (as a launch of its
// own it ran at the latency of one dependent load chain per row: 1.85 TB/s, 7 % of a C5 epoch).  EPILOGUE (a row present in
// one segment): shape = prior + E (*) phi-sum (PXI:304-314 / 356-366), rate = top/rs + cs_other (PXI:300 / 352), mean, the
// row's share of the column sums, its scalar rate (PXI:324 / 377) -- svi_side_kernel's rate_mode 0 statements through the
// same helpers -- so the phi-sum never goes through memory.  Split rows write part[] and are finished by the whole-table pass.
// SKIP: the zero padding of the gathered rows is not fetched -- lanes whose float4 lies in a 64-byte sector past column k
// issue no load (k = 200 in ld = 256: 13 of a row's 16 sectors; k = 100 in ld = 128: 7 of 8).  Only instantiated for those k:
// with k = 50 in ld = 64 every sector holds columns and the kernel is the unmasked one.
template <int LPR, int VPL, int MODE, int UU = HPF_U, bool SKIP = false>
__global__ __launch_bounds__(BLOCK)
__attribute__((amdgpu_waves_per_eu(HPF_SWEEP_WAVES_PER_EU, (MODE != 0) ? HPF_FUSED_MAX_WAVES : 8))) void sweep_kernel(const hpf_segment *__restrict__ segs, int64_t nseg,
                                                      const int32_t *__restrict__ idx,
                                                      const float *__restrict__ y,
                                                      const float *tab_self,  // may alias fa.e_new
                                                      const float *__restrict__ tab_other,
                                                      float *__restrict__ part, const FinalizeArgs fa) {
    constexpr int LD = 4 * LPR * VPL;
    constexpr int NG = WAVE / LPR;  // nonzeros per step
    constexpr int U = UU;           // gathers in flight per wavefront (WAVE/NG = LPR >= 8 is a multiple)
    constexpr int NQ = 4 * VPL;     // factors held per lane during the sweep
    constexpr bool FUSE = (MODE != 0);
    const int lane = threadIdx.x & (WAVE - 1);
    const int g = lane / LPR;
    const int j = lane % LPR;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * WPB;
    if (fa.nseg_dev) nseg = min(nseg, fa.nseg_dev[0]);   // (a batch whose size only the device knows)
    if (fa.sig_peers && blockIdx.x == 0 && wid == 0) hpf_p2p::wave_signal(*fa.sig_peers, fa.sig_kind, fa.sig_epoch);

    // FUSE: after the cross-group fold every group holds the whole accumulator row, so the
    // finalize work is dealt out over ALL 64 lanes: lane (g,j) owns the NC factors q = g + t*NG of
    // its float4s (q = 4v + e  ->  column (v*LPR + j)*4 + e); with NG = 8 > NQ only groups 0-3 work.
    constexpr int NC = (NQ >= NG) ? NQ / NG : 1;
    float csl[NC], csacc[NC];
    int colq[NC];
    if constexpr (FUSE) {
#pragma unroll
        for (int t = 0; t < NC; t++) {
            const int q = g + t * NG;
            colq[t] = (q < NQ) ? ((q >> 2) * LPR + j) * 4 + (q & 3) : LD;  // LD = "no column"
            csl[t] = (colq[t] < fa.k) ? fa.cs_other[colq[t]] : 0.f;
            csacc[t] = 0.f;
            if (fa.cs_other_copy && blockIdx.x == 0 && wid == 0 && colq[t] < LD) fa.cs_other_copy[colq[t]] = csl[t];
        }
    }

    // the closed-form updates of one row with its factors dealt over all 64 lanes (lane (g,j) owns columns
    // colq[t]): a = accumulator entries, eo = the row's old E entries; writes the row's tables, returns the new
    // E entries (row scaled by a power of two: max in [1,2)) in en
    auto finish_row = [&](const float (&a)[NC], const float (&eo)[NC], float (&en)[NC], int row) {
        const float rs_old = fa.rs[row];
        const float base_rte = fa.top_shp / rs_old;
        float sh[NC], rt[NC], fc[NC];
        double ev[NC];
        float fsum = 0.f;
        int ehi = 0;
#pragma unroll
        for (int t = 0; t < NC; t++) {
            const bool valid = colq[t] < fa.k;
            sh[t] = fmaf(eo[t], a[t], fa.prior_shp);
            rt[t] = base_rte + csl[t];
            fc[t] = valid ? sh[t] / rt[t] : 0.f;
            ev[t] = valid ? expect_ratio(sh[t], rt[t]) : 0.0;
            fsum += fc[t];
            ehi = max(ehi, __double2hiint(ev[t]));
            csacc[t] += fc[t];
        }
        fsum = wave_sum(fsum);
        const double inv = row_pow2_scale(ehi);
#pragma unroll
        for (int t = 0; t < NC; t++) {
            const bool valid = colq[t] < fa.k;
            en[t] = valid ? (float)(ev[t] * inv) : 0.f;
            if (colq[t] < LD) {
                const size_t o = (size_t)row * LD + colq[t];
                fa.e_new[o] = en[t];
                if (fa.shp) stream_store(fa.shp + o, valid ? sh[t] : 0.f);
                if (fa.rte) stream_store(fa.rte + o, valid ? rt[t] : 0.f);
                if (fa.fac) stream_store(fa.fac + o, fc[t]);
            }
        }
        if (lane == 0) {
            fa.rs[row] = fa.add_rte + fsum;
            if (fa.rs_prev) fa.rs_prev[row] = rs_old;
        }
    };

    // MODE 2: the same for a row of the OTHER side of a stochastic step (svi_side_kernel, rate_mode 1, rs_mode 1, a flagged
    // row -- the same statements through the same helpers): so / ro = the row's old shape and rate entries
    auto finish_row_svi = [&](const float (&a)[NC], const float (&eo)[NC], const float (&so)[NC], const float (&ro)[NC],
                              int row) {
        const float rs_old = fa.rs[row];
        const float base_rte = fa.top_shp / rs_old;
        float sh[NC], rt[NC], fc[NC], en[NC];
        bool vld[NC];
        float fsum = 0.f;
#pragma unroll
        for (int t = 0; t < NC; t++) {
            const bool valid = colq[t] < fa.k;
            const float fresh = fmaf(eo[t], a[t], fa.prior_shp);
            sh[t] = valid ? blend_shape(fa.w_new, fresh, fa.w_old, so[t]) : 0.f;
            rt[t] = valid ? blend_rate(fa.step, base_rte, csl[t], fa.step_prev, ro[t]) : 0.f;
            vld[t] = valid;
            en[t] = 0.f;
        }
        div_columns<NC>(sh, rt, vld, fc);
#pragma unroll
        for (int t = 0; t < NC; t++) {
            fsum += fc[t];
            csacc[t] += fc[t];
        }
        if (fa.e_new) {
            double ev[NC];
            int ehi = 0;
#pragma unroll
            for (int t = 0; t < NC; t++) {
                ev[t] = (colq[t] < fa.k) ? expect_ratio(sh[t], rt[t]) : 0.0;
                ehi = max(ehi, __double2hiint(ev[t]));
            }
            const double inv = row_pow2_scale(ehi);
#pragma unroll
            for (int t = 0; t < NC; t++) en[t] = (colq[t] < fa.k) ? (float)(ev[t] * inv) : 0.f;
        }
        if constexpr (NG == 1) {
            // lane j owns whole float4s (columns (v*LPR + j)*4 .. +3 = entries 4v .. 4v+3): 16-byte stores
#pragma unroll
            for (int v = 0; v < VPL; v++) {
                if (v * LPR + j < fa.nq4 || !SKIP) {
                    const size_t o4 = (size_t)row * (LD / 4) + v * LPR + j;
                    reinterpret_cast<float4 *>(fa.shp)[o4] = make_float4(sh[4 * v], sh[4 * v + 1], sh[4 * v + 2], sh[4 * v + 3]);
                    reinterpret_cast<float4 *>(fa.rte)[o4] = make_float4(rt[4 * v], rt[4 * v + 1], rt[4 * v + 2], rt[4 * v + 3]);
                    if (fa.fac)
                        reinterpret_cast<float4 *>(fa.fac)[o4] = make_float4(fc[4 * v], fc[4 * v + 1], fc[4 * v + 2], fc[4 * v + 3]);
                    if (fa.e_new)
                        reinterpret_cast<float4 *>(fa.e_new)[o4] = make_float4(en[4 * v], en[4 * v + 1], en[4 * v + 2], en[4 * v + 3]);
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < NC; t++) {
                if (colq[t] < LD) {
                    const size_t o = (size_t)row * LD + colq[t];
                    fa.shp[o] = sh[t];
                    fa.rte[o] = rt[t];
                    if (fa.fac) fa.fac[o] = fc[t];
                    if (fa.e_new) fa.e_new[o] = en[t];
                }
            }
        }
        fsum = wave_sum(fsum);
        if (lane == 0) fa.rs[row] = blend_rate(fa.step, fa.add_rte, fsum, fa.step_prev, rs_old);
    };

    // MODE 3: a row of the BATCH side of a stochastic step (svi_side_kernel, rate_mode 0, a flagged row -- the same statements
    // through the same helpers): a = the row's phi-sum entries, eo = its E entries (formed in the prologue below)
    auto finish_row_batch = [&](const float (&a)[NC], const float (&eo)[NC], int row) {
        const float rs_old = fa.rs[row];
        const float base_rte = fa.top_shp / rs_old;
        float sh[NC], rt[NC], fc[NC];
        bool vld[NC];
        float fsum = 0.f;
#pragma unroll
        for (int t = 0; t < NC; t++) {
            const bool valid = colq[t] < fa.k;
            const float fresh = fmaf(eo[t], a[t], fa.prior_shp);
            sh[t] = valid ? blend_shape(fa.w_new, fresh, fa.w_old, 0.f) : 0.f;
            rt[t] = valid ? base_rte + csl[t] : 0.f;
            vld[t] = valid;
        }
        div_columns<NC>(sh, rt, vld, fc);
#pragma unroll
        for (int t = 0; t < NC; t++) {
            fsum += fc[t];
            csacc[t] += fc[t];
        }
        if constexpr (NG == 1) {
#pragma unroll
            for (int v = 0; v < VPL; v++) {
                if (v * LPR + j < fa.nq4 || !SKIP) {
                    const size_t o4 = (size_t)row * (LD / 4) + v * LPR + j;
                    reinterpret_cast<float4 *>(fa.shp)[o4] = make_float4(sh[4 * v], sh[4 * v + 1], sh[4 * v + 2], sh[4 * v + 3]);
                    if (fa.rte)
                        reinterpret_cast<float4 *>(fa.rte)[o4] = make_float4(rt[4 * v], rt[4 * v + 1], rt[4 * v + 2], rt[4 * v + 3]);
                    if (fa.fac)
                        reinterpret_cast<float4 *>(fa.fac)[o4] = make_float4(fc[4 * v], fc[4 * v + 1], fc[4 * v + 2], fc[4 * v + 3]);
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < NC; t++) {
                if (colq[t] < LD) {
                    const size_t o = (size_t)row * LD + colq[t];
                    fa.shp[o] = sh[t];
                    if (fa.rte) fa.rte[o] = rt[t];
                    if (fa.fac) fa.fac[o] = fc[t];
                }
            }
        }
        fsum = wave_sum(fsum);
        if (lane == 0) {
            fa.rs[row] = blend_rate(fa.step, fa.add_rte, fsum, fa.step_prev, rs_old);
            if (fa.rs_prev) fa.rs_prev[row] = rs_old;
        }
    };

    // MODE 3 prologue operands, requested ONE SEGMENT AHEAD: the shape entries (and, when the rate is a stored table, the rate
    // entries) of the row in this lane's dealt columns colq[t], and the scalar a factored rate is formed with
    struct RowRequest {
        float sh[NC], rt[NC], rsr;
    };
    float cs_rate[(MODE == 3) ? NC : 1];
    if constexpr (MODE == 3) {
#pragma unroll
        for (int t = 0; t < NC; t++) cs_rate[t] = (fa.rate_rs && colq[t] < fa.k) ? fa.rate_cs[colq[t]] : 0.f;
    }
    auto request_row = [&](int row, RowRequest &rq) {
        rq.rsr = fa.rate_rs ? fa.rate_rs[row] : 1.f;
        if constexpr (NG == 1) {
            const float4 *sp4 = reinterpret_cast<const float4 *>(fa.shp + (size_t)row * LD);
            const float4 *rp4 = reinterpret_cast<const float4 *>(fa.rte_in + (size_t)row * LD);
#pragma unroll
            for (int v = 0; v < VPL; v++) {
                float4 sq = make_float4(1.f, 1.f, 1.f, 1.f), rq4 = make_float4(1.f, 1.f, 1.f, 1.f);
                if (!SKIP || v * LPR + j < fa.nq4) {
                    sq = ld4(sp4 + v * LPR + j);
                    if (!fa.rate_rs) rq4 = ld4(rp4 + v * LPR + j);
                }
                rq.sh[4 * v] = sq.x, rq.sh[4 * v + 1] = sq.y, rq.sh[4 * v + 2] = sq.z, rq.sh[4 * v + 3] = sq.w;
                rq.rt[4 * v] = rq4.x, rq.rt[4 * v + 1] = rq4.y, rq.rt[4 * v + 2] = rq4.z, rq.rt[4 * v + 3] = rq4.w;
            }
        } else {
#pragma unroll
            for (int t = 0; t < NC; t++) {
                const bool valid = colq[t] < fa.k;
                rq.sh[t] = valid ? fa.shp[(size_t)row * LD + colq[t]] : 1.f;
                rq.rt[t] = (valid && !fa.rate_rs) ? fa.rte_in[(size_t)row * LD + colq[t]] : 1.f;
            }
        }
    };
    // MODE 2 / 3 (the rows of a stochastic batch: 8-50 nonzeros, one or a few rounds of gathers each): a wave's segments are
    // PIPELINED -- the descriptor of the segment after next is requested while the current segment is swept (wave-uniform:
    // scalar registers) and, MODE 2 (the other side's rows: 8-12 nonzeros at C5), so are the first 64 ids / counts of the next
    // segment: a row costs the latency of its gathers, not of the chain descriptor -> ids -> gathers.  (MODE 3 requests the
    // next row's shape instead; with the ids as well it would hold 131 registers: three waves per SIMD instead of four.)
    // (the same for the plain sweep's short-row launch -- the item-side ranges of an 8-rank sharded iteration, ~16 nonzeros per
    //  row -- changed nothing: 0.580 vs 0.581 ms per rank-0-of-8 iteration, profiles/r06_shard_probe_pipelined.txt)
    constexpr bool PIPE = (MODE == 2 || MODE == 3);
    RowRequest rq_next;
    hpf_segment sg_cur, sg_next;
    sg_cur.begin = 0, sg_cur.len = 0, sg_cur.row = 0;
    sg_next = sg_cur;
    int c_pre = 0;
    float y_pre = 0.f;
    auto request_chunk0 = [&](const hpf_segment &d) {
        const int l0 = min(WAVE, d.len & HPF_SEG_LEN_MASK);
        c_pre = 0;
        y_pre = 0.f;
        if (lane < l0) {
            c_pre = stream_load(idx + d.begin + lane);
            y_pre = stream_load(y + d.begin + lane);
        }
    };
    if constexpr (PIPE) {
        const int64_t sg0 = (int64_t)blockIdx.x * WPB + wid;
        if (sg0 < nseg) {
            sg_cur = segs[sg0];
            if constexpr (MODE == 3) request_row(sg_cur.row, rq_next);
            if constexpr (MODE == 2) request_chunk0(sg_cur);
            if (sg0 + nwaves < nseg) sg_next = segs[sg0 + nwaves];
        }
    }

    for (int64_t sg = (int64_t)blockIdx.x * WPB + wid; sg < nseg; sg += nwaves) {
        const hpf_segment sgm = PIPE ? sg_cur : segs[sg];
        const int len = sgm.len & HPF_SEG_LEN_MASK;
        const float4 *selfp = reinterpret_cast<const float4 *>(tab_self + (size_t)sgm.row * LD);
        float4 rv[VPL], acc[VPL];
        float en3[(MODE == 3) ? NC : 1];
        const int c_first = c_pre;
        const float y_first = y_pre;
        if constexpr (MODE == 2) {
            if (sg + nwaves < nseg) {
                sg_cur = sg_next;
                request_chunk0(sg_cur);
                if (sg + 2 * nwaves < nseg) sg_next = segs[sg + 2 * nwaves];
            }
        }
        if constexpr (MODE == 3) {
            // PROLOGUE: this row's E row from the operands requested one segment ago; then the next segment's are requested
            const RowRequest rq = rq_next;
            if (sg + nwaves < nseg) {
                sg_cur = sg_next;
                request_row(sg_cur.row, rq_next);
                if (sg + 2 * nwaves < nseg) sg_next = segs[sg + 2 * nwaves];
            }
            const float base_old = fa.rate_rs ? fa.rate_top / rq.rsr : 0.f;
            double ev[NC];
            int ehi = 0;
#pragma unroll
            for (int t = 0; t < NC; t++) {
                const bool valid = colq[t] < fa.k;
                const float rt = valid ? (fa.rate_rs ? base_old + cs_rate[t] : rq.rt[t]) : 1.f;
                ev[t] = valid ? expect_ratio(valid ? rq.sh[t] : 1.f, rt) : 0.0;
                ehi = max(ehi, __double2hiint(ev[t]));
            }
            const double inv = row_pow2_scale(ehi);
#pragma unroll
            for (int t = 0; t < NC; t++) en3[t] = (colq[t] < fa.k) ? (float)(ev[t] * inv) : 0.f;
            // the row of the E table: the other side's sweep of this step gathers it (every segment of a split row writes
            // the same values)
            if constexpr (NG == 1) {
#pragma unroll
                for (int v = 0; v < VPL; v++) {
                    rv[v] = make_float4(en3[4 * v], en3[4 * v + 1], en3[4 * v + 2], en3[4 * v + 3]);
                    if (!SKIP || v * LPR + j < fa.nq4)
                        reinterpret_cast<float4 *>(fa.e_new)[(size_t)sgm.row * (LD / 4) + v * LPR + j] = rv[v];
                }
            } else {
#pragma unroll
                for (int t = 0; t < NC; t++)
                    if (colq[t] < LD) fa.e_new[(size_t)sgm.row * LD + colq[t]] = en3[t];
                // every lane group needs the whole row as float4s: component q of lane (g', j)'s float4 is held, in the
                // dealt layout, by lane (q % NG, j) in slot q / NG
                float comp[NQ];
#pragma unroll
                for (int q = 0; q < NQ; q++) comp[q] = __shfl(en3[(q / NG < NC) ? q / NG : 0], (q % NG) * LPR + j);
#pragma unroll
                for (int v = 0; v < VPL; v++) rv[v] = make_float4(comp[4 * v], comp[4 * v + 1], comp[4 * v + 2], comp[4 * v + 3]);
            }
#pragma unroll
            for (int v = 0; v < VPL; v++) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
#pragma unroll
            for (int v = 0; v < VPL; v++) {
                rv[v] = selfp[v * LPR + j];
                acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        // MODE 2: the row's old shapes and rates are requested with its E row, ahead of the gathers (in-order returns: by
        // the time the last gather has landed they are there)
        float4 sold[MODE == 2 ? VPL : 1], rold[MODE == 2 ? VPL : 1];
        if constexpr (MODE == 2) {
            if ((sgm.len & HPF_SEG_WHOLE_ROW) != 0) {
                const float4 *sp4 = reinterpret_cast<const float4 *>(fa.shp + (size_t)sgm.row * LD);
                const float4 *rp4 = reinterpret_cast<const float4 *>(fa.rte + (size_t)sgm.row * LD);
#pragma unroll
                for (int v = 0; v < VPL; v++) {
                    const bool in = !SKIP || v * LPR + j < fa.nq4;
                    sold[v] = in ? sp4[v * LPR + j] : make_float4(0.f, 0.f, 0.f, 0.f);
                    rold[v] = in ? rp4[v * LPR + j] : make_float4(1.f, 1.f, 1.f, 1.f);
                }
            }
        }
        const int32_t *ip = idx + sgm.begin;
        const float *yp = y + sgm.begin;
        const bool whole_row = (sgm.len & HPF_SEG_WHOLE_ROW) != 0;

        for (int base = 0; base < len; base += WAVE) {
            const int n = min(WAVE, len - base);
            int myc = 0;
            float myy = 0.f;
            if (MODE == 2 && base == 0) {       // (requested one segment ago)
                myc = c_first;
                myy = y_first;
            } else if (lane < n) {
                myc = stream_load(ip + base + lane);
                myy = stream_load(yp + base + lane);
            }
            int nsteps = (n + NG - 1) / NG;
            nsteps = (nsteps + U - 1) & ~(U - 1);
            for (int t0 = 0; t0 < nsteps; t0 += U) {
                float4 o[U][VPL];
                float yy[U];
                int cc[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int src = (t0 + u) * NG + g;
                    cc[u] = __shfl(myc, src);
                    yy[u] = __shfl(myy, src);
                    const float4 *op = reinterpret_cast<const float4 *>(tab_other + (size_t)cc[u] * LD);
#pragma unroll
                    for (int v = 0; v < VPL; v++) {
                        if constexpr (SKIP)
                            o[u][v] = (v * LPR + j < fa.nq4) ? op[v * LPR + j] : make_float4(0.f, 0.f, 0.f, 0.f);
                        else
                            o[u][v] = op[v * LPR + j];
                    }
                }
#pragma unroll
                for (int u = 0; u < U; u++) {
                    float p = dot4(rv[0], o[u][0]);   // (not 0.f + ...: an add the compiler must keep for -0)
#pragma unroll
                    for (int v = 1; v < VPL; v++) p += dot4(rv[v], o[u][v]);
                    const float s = group_sum<LPR>(p);
                    const float w = (yy[u] > 0.f) ? yy[u] * __builtin_amdgcn_rcpf(s) : 0.f;
#pragma unroll
                    for (int v = 0; v < VPL; v++) {
                        acc[v].x = fmaf(w, o[u][v].x, acc[v].x);
                        acc[v].y = fmaf(w, o[u][v].y, acc[v].y);
                        acc[v].z = fmaf(w, o[u][v].z, acc[v].z);
                        acc[v].w = fmaf(w, o[u][v].w, acc[v].w);
                    }
                }
            }
        }
        // fold the NG sub-wave groups (fixed order: deterministic)
#pragma unroll
        for (int v = 0; v < VPL; v++) {
#pragma unroll
            for (int m = LPR; m < WAVE; m <<= 1) {
                acc[v].x += __shfl_xor(acc[v].x, m);
                acc[v].y += __shfl_xor(acc[v].y, m);
                acc[v].z += __shfl_xor(acc[v].z, m);
                acc[v].w += __shfl_xor(acc[v].w, m);
            }
        }
        if (MODE != 1 && MODE != 3 && whole_row && fa.acc_rows) {
            // the row's complete accumulator goes straight into the packed exchange buffer
            if (g == 0) {
                float *ar = fa.acc_rows + (size_t)sgm.row * fa.acc_ld;
#pragma unroll
                for (int v = 0; v < VPL; v++) {
                    const int c = (v * LPR + j) * 4;
                    if (c + 0 < fa.acc_ld) ar[c + 0] = acc[v].x;
                    if (c + 1 < fa.acc_ld) ar[c + 1] = acc[v].y;
                    if (c + 2 < fa.acc_ld) ar[c + 2] = acc[v].z;
                    if (c + 3 < fa.acc_ld) ar[c + 3] = acc[v].w;
                }
            }
        } else if (MODE == 0 || !whole_row) {
            if (g == 0) {
                float4 *pp = reinterpret_cast<float4 *>(part + (size_t)sg * LD);
#pragma unroll
                for (int v = 0; v < VPL; v++) pp[v * LPR + j] = acc[v];
            }
        } else if constexpr (MODE == 1) {
            // this segment is its row's only one: finish the row here (same math as
            // row_finalize_kernel), overlapping the fp64 work with other waves' gathers
            float a[NC], eo[NC], en[NC];
#pragma unroll
            for (int t = 0; t < NC; t++) {
                const int q = g + t * NG;
                float av_ = 0.f, ov_ = 0.f;
#pragma unroll
                for (int qq = 0; qq < NQ; qq++) {  // pick float4 component q (q is lane-dependent)
                    const int v = qq >> 2, e = qq & 3;
                    const float av = (e == 0) ? acc[v].x : (e == 1) ? acc[v].y : (e == 2) ? acc[v].z : acc[v].w;
                    const float ov = (e == 0) ? rv[v].x : (e == 1) ? rv[v].y : (e == 2) ? rv[v].z : rv[v].w;
                    av_ = (qq == q) ? av : av_;
                    ov_ = (qq == q) ? ov : ov_;
                }
                a[t] = av_;
                eo[t] = ov_;
            }
            finish_row(a, eo, en, sgm.row);
        } else if constexpr (MODE == 2) {
            float a[NC], eo[NC], so[NC], ro[NC];
#pragma unroll
            for (int t = 0; t < NC; t++) {
                const int q = g + t * NG;
                float av_ = 0.f, ov_ = 0.f, sv_ = 0.f, rv_ = 1.f;
#pragma unroll
                for (int qq = 0; qq < NQ; qq++) {  // pick float4 component q (q is lane-dependent unless NG == 1)
                    const int v = qq >> 2, e = qq & 3;
                    const float av = (e == 0) ? acc[v].x : (e == 1) ? acc[v].y : (e == 2) ? acc[v].z : acc[v].w;
                    const float ov = (e == 0) ? rv[v].x : (e == 1) ? rv[v].y : (e == 2) ? rv[v].z : rv[v].w;
                    const float sv = (e == 0) ? sold[v].x : (e == 1) ? sold[v].y : (e == 2) ? sold[v].z : sold[v].w;
                    const float rr = (e == 0) ? rold[v].x : (e == 1) ? rold[v].y : (e == 2) ? rold[v].z : rold[v].w;
                    av_ = (qq == q) ? av : av_;
                    ov_ = (qq == q) ? ov : ov_;
                    sv_ = (qq == q) ? sv : sv_;
                    rv_ = (qq == q) ? rr : rv_;
                }
                a[t] = av_;
                eo[t] = ov_;
                so[t] = sv_;
                ro[t] = rv_;
            }
            finish_row_svi(a, eo, so, ro, sgm.row);
        } else if constexpr (MODE == 3) {
            float a[NC];
#pragma unroll
            for (int t = 0; t < NC; t++) {
                const int q = g + t * NG;
                float av_ = 0.f;
#pragma unroll
                for (int qq = 0; qq < NQ; qq++) {  // pick float4 component q (q is lane-dependent unless NG == 1)
                    const int v = qq >> 2, e = qq & 3;
                    const float av = (e == 0) ? acc[v].x : (e == 1) ? acc[v].y : (e == 2) ? acc[v].z : acc[v].w;
                    av_ = (qq == q) ? av : av_;
                }
                a[t] = av_;
            }
            finish_row_batch(a, en3, sgm.row);
        }
    }

    if constexpr (FUSE) {
        // per-block column sums of fac over the rows finished here
        __shared__ float red[WPB][LD];
#pragma unroll
        for (int t = 0; t < NC; t++) {
            if (colq[t] < LD) red[wid][colq[t]] = csacc[t];
        }
        __syncthreads();
        for (int c = threadIdx.x; c < LD; c += BLOCK) {
            float t = red[0][c];
#pragma unroll
            for (int w2 = 1; w2 < WPB; w2++) t += red[w2][c];
            fa.cs_partial[(size_t)blockIdx.x * LD + c] = t;
        }
    }
}

// DENSE: every row has exactly one accumulator row (row_seg_ptr == row_list == nullptr; the replicated item
// finalizer of the sharded path, over all-reduced statistics): the loads of the NEXT row are issued before the
// fp64 work of the current one, and the generic segment loops are compiled out.
// DENSE only: the launch covers virtual rows v = 0..nrows-1; v in range i (v_begin[i] <= v < v_begin[i+1]) stands for
// accumulator row t = t_begin[i] + (v - v_begin[i]) of `part` -- e_new is indexed by t as well: it is the all-gather
// send buffer -- and table row r = row_begin[i] + (v - v_begin[i]) of e_old/shp/rte/fac/rs.  The slices one rank owns
// of several item ranges are finished by ONE launch.
struct RowRanges {
    int n;                                   // 0: identity (accumulator row t <-> table row t, e_new row t)
    int64_t v_begin[HPF_MAX_ROW_RANGES];     // first virtual index of range i (prefix sums of the range lengths)
    int64_t t_begin[HPF_MAX_ROW_RANGES];     // its first accumulator row (and e_new row)
    int64_t row_begin[HPF_MAX_ROW_RANGES];   // its first table row
};

template <int LD, bool DENSE>
__global__ __launch_bounds__(BLOCK) void row_finalize_kernel(
    const float *__restrict__ part, const int64_t *__restrict__ row_seg_ptr, const int64_t *__restrict__ row_list,
    int64_t nrows, const float *e_old, float *e_new, float *__restrict__ shp, float *__restrict__ rte,
    float *__restrict__ fac, float *rs, float *__restrict__ rs_prev, const float *__restrict__ cs_other,
    float *__restrict__ cs_partial, float prior_shp, float top_shp, float add_rte, int k, int part_ld,
    const RowRanges rr, int e_new_ld) {
    // e_new_ld: row stride of e_new (LD, or k <= e_new_ld < LD: the packed all-gather send buffer of the sharded path)
    constexpr int CPL = (LD + WAVE - 1) / WAVE;  // factors per lane
    __shared__ float red[WPB][LD];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * WPB;

    float csl[CPL], csacc[CPL];
#pragma unroll
    for (int q = 0; q < CPL; q++) {
        const int c = lane + WAVE * q;
        csl[q] = (c < k) ? cs_other[c] : 0.f;
        csacc[q] = 0.f;
    }

    // the closed-form updates of row r from its accumulator entries a[], old E entries eo[] and old scalar rate
    auto finish = [&](int64_t r, int64_t re, const float (&a)[CPL], const float (&eo)[CPL], float rs_old) {
        const float base_rte = top_shp / rs_old;
        float sh[CPL], rt[CPL], fc[CPL];
        double ev[CPL];
        float fsum = 0.f;
        int ehi = 0;
#pragma unroll
        for (int q = 0; q < CPL; q++) {
            const bool valid = lane + WAVE * q < k;
            sh[q] = fmaf(eo[q], a[q], prior_shp);
            rt[q] = base_rte + csl[q];
            fc[q] = valid ? sh[q] / rt[q] : 0.f;
            ev[q] = valid ? expect_ratio(sh[q], rt[q]) : 0.0;
            fsum += fc[q];
            ehi = max(ehi, __double2hiint(ev[q]));
            csacc[q] += fc[q];
        }
        fsum = wave_sum(fsum);
        const double inv = row_pow2_scale(ehi);
#pragma unroll
        for (int q = 0; q < CPL; q++) {
            const int c = lane + WAVE * q;
            if (c < LD) {
                const size_t o = (size_t)r * LD + c;
                const bool valid = c < k;
                if (c < e_new_ld) e_new[(size_t)re * e_new_ld + c] = valid ? (float)(ev[q] * inv) : 0.f;
                if (shp) shp[o] = valid ? sh[q] : 0.f;
                if (rte) rte[o] = valid ? rt[q] : 0.f;
                if (fac) fac[o] = fc[q];
            }
        }
        if (lane == 0) {
            rs[r] = add_rte + fsum;
            if (rs_prev) rs_prev[r] = rs_old;
        }
    };

    if constexpr (DENSE) {
        auto locate = [&](int64_t v, int64_t &t, int64_t &r) {
            t = v;
            r = v;
            if (rr.n > 0) {
                t = rr.t_begin[0] + v;
                r = rr.row_begin[0] + v;
#pragma unroll
                for (int i = 1; i < HPF_MAX_ROW_RANGES; i++)
                    if (i < rr.n && v >= rr.v_begin[i]) {
                        t = rr.t_begin[i] + (v - rr.v_begin[i]);
                        r = rr.row_begin[i] + (v - rr.v_begin[i]);
                    }
            }
        };
        auto fetch = [&](int64_t t, int64_t r, float (&a)[CPL], float (&eo)[CPL], float &rs_old) {
#pragma unroll
            for (int q = 0; q < CPL; q++) {
                const int c = lane + WAVE * q;
                a[q] = (c < part_ld) ? part[(size_t)t * part_ld + c] : 0.f;
                eo[q] = (c < LD) ? e_old[(size_t)r * LD + c] : 0.f;
            }
            rs_old = rs[r];
        };
        int64_t v = (int64_t)blockIdx.x * WPB + wid;
        float a_n[CPL], eo_n[CPL], rs_n = 1.f;
        int64_t t_n = 0, r_n = 0;
        if (v < nrows) {
            locate(v, t_n, r_n);
            fetch(t_n, r_n, a_n, eo_n, rs_n);
        }
        for (; v < nrows; v += nwaves) {
            float a[CPL], eo[CPL];
            const float rs_old = rs_n;
            const int64_t r = r_n, t = t_n;
#pragma unroll
            for (int q = 0; q < CPL; q++) {
                a[q] = a_n[q];
                eo[q] = eo_n[q];
            }
            if (v + nwaves < nrows) {
                locate(v + nwaves, t_n, r_n);
                fetch(t_n, r_n, a_n, eo_n, rs_n);
            }
            finish(r, (rr.n > 0) ? t : r, a, eo, rs_old);
        }
    } else {
        for (int64_t t = (int64_t)blockIdx.x * WPB + wid; t < nrows; t += nwaves) {
            const int64_t r = row_list ? row_list[t] : t;
            int64_t s0 = r, s1 = r + 1;
            if (row_seg_ptr) {
                s0 = row_seg_ptr[r];
                s1 = row_seg_ptr[r + 1];
            }
            float a[CPL], eo[CPL];
#pragma unroll
            for (int q = 0; q < CPL; q++) {
                const int c = lane + WAVE * q;
                a[q] = 0.f;
                if (c < part_ld) {
                    // popular rows have up to ~1e3 segments: 8 independent loads in flight, fixed fold order
                    int64_t sg = s0;
                    for (; sg + 8 <= s1; sg += 8) {
                        float p[8];
#pragma unroll
                        for (int u = 0; u < 8; u++) p[u] = part[(size_t)(sg + u) * part_ld + c];
                        a[q] += ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
                    }
                    for (; sg < s1; sg++) a[q] += part[(size_t)sg * part_ld + c];
                }
                eo[q] = (c < LD) ? e_old[(size_t)r * LD + c] : 0.f;
            }
            finish(r, r, a, eo, rs[r]);
        }
    }

    // per-block column sums of fac (fixed order -> reproducible for a fixed grid)
#pragma unroll
    for (int q = 0; q < CPL; q++) {
        const int c = lane + WAVE * q;
        if (c < LD) red[wid][c] = csacc[q];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < LD; c += BLOCK) {
        float t = red[0][c];
#pragma unroll
        for (int w2 = 1; w2 < WPB; w2++) t += red[w2][c];
        cs_partial[(size_t)blockIdx.x * LD + c] = t;
    }
}

// ----------------------------------------------------------------------------------------------------------------
// The item finalizer of the sharded path in TWO parts ("gather-early" exchange).  The finalizer's
// expensive half -- shp = c + e (*) acc and exp(psi(shp)) -- needs only the reduce-scattered statistics; only the rate
// rte = t_shp/tau + colsum(Theta) needs the user side of the same iteration.  So the owner of a slice computes
//   part 1 (item_shape_kernel):  shp (kept in place of acc), num = exp(psi(shp)) row-scaled, base = t_shp/tau_old
// right after the reduce-scatter, and [num | base] rows are ALL-GATHERED WHILE THE USER SWEEP RUNS; after colsum(Theta)
// has been summed over the ranks every rank runs
//   part 2 (item_apply_kernel):  E = num / (base + colsum(Theta)), row-scaled, for ALL items into its replicated table
//                                (identical arithmetic on identical inputs: replicas stay bit-identical); the owner of a
//                                row also finishes Beta = shp/rte, tau = add + sum_k Beta and its colsum(Beta) partials.
// Against the one-part finalizer E carries one more float32 rounding (num is rounded before the division).
// ----------------------------------------------------------------------------------------------------------------
template <int LD>
__global__ __launch_bounds__(BLOCK) void item_shape_kernel(const float *__restrict__ acc, const float *__restrict__ e_old,
                                                           float *__restrict__ shp_out, float *__restrict__ send,
                                                           int sld, const float *__restrict__ rs,
                                                           float *__restrict__ rs_prev, float prior_shp, float top_shp,
                                                           int k, const RowRanges rr, int64_t nrows) {
    // send rows: k numerators, the row's base rate at column k, zero up to the stride sld (a multiple of 4 floats, so
    // that part 2 reads them as float4); shp_out rows: the padded table layout [.][LD].
    // SR rows per wave step, all their loads issued first: this kernel runs on the exchange stream BESIDE the sweeps, which
    // leave it one wave slot per SIMD and a saturated memory system -- with one row in flight per wave it took 80-100 us
    // there against 19 us on an idle GPU (profiles/r03_timeline_links_gather_carried_300GBps.txt)
    constexpr int CPL = (LD + WAVE - 1) / WAVE;
    constexpr int SR = (CPL == 1) ? 4 : (CPL <= 4 ? 2 : 1);
    const int lane = threadIdx.x & (WAVE - 1);
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * WPB;
    for (int64_t v0 = ((int64_t)blockIdx.x * WPB + wid) * SR; v0 < nrows; v0 += nwaves * SR) {
        int64_t tt[SR], rr_[SR];
        float a[SR][CPL], eo[SR][CPL], rs_old[SR];
        bool live[SR];
#pragma unroll
        for (int i = 0; i < SR; i++) {
            const int64_t v = v0 + i;
            live[i] = v < nrows;
            int64_t t = rr.t_begin[0] + v, r = rr.row_begin[0] + v;
#pragma unroll
            for (int q = 1; q < HPF_MAX_ROW_RANGES; q++)
                if (q < rr.n && v >= rr.v_begin[q]) {
                    t = rr.t_begin[q] + (v - rr.v_begin[q]);
                    r = rr.row_begin[q] + (v - rr.v_begin[q]);
                }
            tt[i] = t;
            rr_[i] = r;
#pragma unroll
            for (int q = 0; q < CPL; q++) {
                const int c = lane + WAVE * q;
                const bool valid = live[i] && c < k;
                a[i][q] = valid ? acc[(size_t)t * k + c] : 0.f;
                eo[i][q] = valid ? e_old[(size_t)r * LD + c] : 0.f;
            }
            rs_old[i] = live[i] ? rs[r] : 1.f;
        }
#pragma unroll
        for (int i = 0; i < SR; i++) {
            if (!live[i]) continue;          // (wave-uniform)
            const int64_t t = tt[i], r = rr_[i];
            double ev[CPL];
            float sh[CPL];
            int ehi = 0;
#pragma unroll
            for (int q = 0; q < CPL; q++) {
                const int c = lane + WAVE * q;
                const bool valid = c < k;
                sh[q] = fmaf(eo[i][q], a[i][q], prior_shp);
                ev[q] = valid ? expect_ratio(sh[q], 1.0f) : 0.0;       // exp(psi(shp))
                ehi = max(ehi, __double2hiint(ev[q]));
            }
            const double inv = row_pow2_scale(ehi);
#pragma unroll
            for (int q = 0; q < CPL; q++) {
                const int c = lane + WAVE * q;
                if (c < LD) shp_out[(size_t)t * LD + c] = (c < k) ? sh[q] : 0.f;      // the shape row, for part 2
                if (c < k) send[(size_t)t * sld + c] = (float)(ev[q] * inv);
            }
            // the base rate and the zero tail of the payload row (sld - k <= 4 columns; with k == LD they lie past the
            // table row)
            if (lane < sld - k) send[(size_t)t * sld + k + lane] = (lane == 0) ? top_shp / rs_old[i] : 0.f;
            if (lane == 0 && rs_prev) rs_prev[r] = rs_old[i];
        }
    }
}

struct ApplyRanges {   // the item ranges of the exchange in issue order = the order of the slices inside a rank's block
    int n;
    int64_t lo[HPF_MAX_ROW_RANGES], m[HPF_MAX_ROW_RANGES], t0[HPF_MAX_ROW_RANGES];   // first row, slice rows, slice offset
    int64_t total;     // rows per rank in the gathered buffer (= sum of m)
};

template <int W>
__device__ __forceinline__ float group_max(float v) {          // all lanes of an aligned W-lane group get the maximum
    v = fmaxf(v, dpp_f<0xB1>(v));
    v = fmaxf(v, dpp_f<0x4E>(v));
    if constexpr (W >= 8) v = fmaxf(v, dpp_f<0x141>(v));
    if constexpr (W >= 16) v = fmaxf(v, dpp_f<0x140>(v));
    if constexpr (W >= 32) v = fmaxf(v, __shfl_xor(v, 16));
    if constexpr (W >= 64) v = fmaxf(v, __shfl_xor(v, 32));
    return v;
}

// grid (gx, world): blockIdx.y = the rank whose block of the gathered buffer is read, so a row's owner costs nothing.
// A row is held by LPR lanes as one float4 each (VPL of them when ld > 256), 64/LPR rows per wave step, R steps in
// flight -- the layout of the sweep; with one row per wavefront this kernel spent 340 instructions per row and was
// bound by instruction issue at a third of its memory rate.  All loads, then all arithmetic, then all stores.
// Direct exchange: the rows of owner o are read from own.block[o] -- that rank's send buffer as mapped here (the
// all-gather happens inside this kernel's loads), or a local copy of it -- once flags[own.wait_kind][o] (own.local_flag:
// [this rank]) has reached own.epoch; own.n == 0: the gathered buffer `recv`.
struct OwnerBlocks {
    int n, wait_kind, local_flag;                 // wait_kind < 0: nothing to wait for
    uint32_t epoch;
    const float *block[HPF_P2P_MAX_RANKS];
    hpf_p2p::Peers peers;
};

template <int LPR, int VPL>
__global__ __launch_bounds__(BLOCK) void item_apply_kernel(const float *__restrict__ recv, int sld,
                                                           const float *__restrict__ shp_own, float *__restrict__ e_tab,
                                                           float *__restrict__ shp, float *__restrict__ fac,
                                                           float *__restrict__ rs, const float *__restrict__ cs_other,
                                                           float *__restrict__ cs_partial, float add_rte, int k,
                                                           int rank, int64_t nrows, const ApplyRanges ar,
                                                           const OwnerBlocks own) {
    constexpr int LD = 4 * LPR * VPL;
    constexpr int NG = WAVE / LPR;
    constexpr int R = (VPL == 1) ? 4 : (VPL == 2 ? 2 : 1);
    __shared__ float red[WPB][LD];
    const int lane = threadIdx.x & (WAVE - 1);
    const int g = lane / LPR, j = lane % LPR;
    const int wid = threadIdx.x >> 6;
    const int owner = blockIdx.y;
    const bool mine = owner == rank;
    if (own.n > 0 && own.wait_kind >= 0) {
        if (own.local_flag || mine)       // (a rank raises its flags in its own control block too)
            hpf_p2p::block_acquire_self(own.peers, own.wait_kind, own.epoch);
        else
            hpf_p2p::block_acquire(own.peers, own.wait_kind, own.epoch, 1u << owner);
    }
    const float *block = (own.n > 0) ? own.block[owner] : recv + (size_t)owner * ar.total * sld;
    const int sq = min(sld >> 2, LPR * VPL);                   // float4s of a gathered row that hold numerators
    float4 csl[VPL], csacc[VPL];
    bool valid[VPL][4];
#pragma unroll
    for (int v = 0; v < VPL; v++) {
        const int c = (v * LPR + j) * 4;
        csl[v] = make_float4(c + 0 < k ? cs_other[c + 0] : 0.f, c + 1 < k ? cs_other[c + 1] : 0.f,
                             c + 2 < k ? cs_other[c + 2] : 0.f, c + 3 < k ? cs_other[c + 3] : 0.f);
        csacc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int e = 0; e < 4; e++) valid[v][e] = c + e < k;
    }
    const int64_t ngroups = (int64_t)gridDim.x * WPB * NG;
    const int64_t gid = ((int64_t)blockIdx.x * WPB + wid) * NG + g;
    const int64_t iters = (ar.total + ngroups * R - 1) / (ngroups * R);        // wave-uniform trip count
    for (int64_t it = 0; it < iters; it++) {
        float4 nv[R][VPL], sv[R][VPL];
        float base[R];
        int64_t row[R];
        bool live[R];
#pragma unroll
        for (int i = 0; i < R; i++) {
            const int64_t pos = (it * R + i) * ngroups + gid;
            int64_t lo = ar.lo[0], m = ar.m[0], t0 = ar.t0[0];
#pragma unroll
            for (int q = 1; q < HPF_MAX_ROW_RANGES; q++)
                if (q < ar.n && pos >= ar.t0[q]) {
                    lo = ar.lo[q];
                    m = ar.m[q];
                    t0 = ar.t0[q];
                }
            row[i] = lo + (int64_t)owner * m + (pos - t0);
            live[i] = pos < ar.total && row[i] < nrows;            // (slices end in pad rows past the last item)
            const float4 *src = reinterpret_cast<const float4 *>(block + (size_t)(live[i] ? pos : 0) * sld);
            const float4 *sp = reinterpret_cast<const float4 *>(shp_own + (size_t)(live[i] ? pos : 0) * LD);
            base[i] = live[i] ? block[(size_t)pos * sld + k] : 1.f;     // (one address per lane group: a broadcast)
#pragma unroll
            for (int v = 0; v < VPL; v++) {
                const int q4 = v * LPR + j;
                nv[i][v] = (live[i] && q4 < sq) ? src[q4] : make_float4(0.f, 0.f, 0.f, 0.f);
                sv[i][v] = (mine && live[i]) ? sp[q4] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        float4 en[R][VPL], fc[R][VPL];
        float fs[R];
#pragma unroll
        for (int i = 0; i < R; i++) {
            float emax = 0.f, fsum = 0.f;
#pragma unroll
            for (int v = 0; v < VPL; v++) {
                const float n4[4] = {nv[i][v].x, nv[i][v].y, nv[i][v].z, nv[i][v].w};
                const float s4[4] = {sv[i][v].x, sv[i][v].y, sv[i][v].z, sv[i][v].w};
                const float c4[4] = {csl[v].x, csl[v].y, csl[v].z, csl[v].w};
                float e4[4], f4[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float rt = base[i] + c4[e];
                    // E only has to be the same on every rank and accurate to a few ulps: reciprocal + multiply (the
                    // means keep the correctly rounded division of the one-part finalizer)
                    e4[e] = valid[v][e] ? n4[e] * __builtin_amdgcn_rcpf(rt) : 0.f;
                    f4[e] = (mine && valid[v][e] && live[i]) ? s4[e] / rt : 0.f;
                    emax = fmaxf(emax, e4[e]);
                    fsum += f4[e];
                }
                en[i][v] = make_float4(e4[0], e4[1], e4[2], e4[3]);
                fc[i][v] = make_float4(f4[0], f4[1], f4[2], f4[3]);
                csacc[v].x += f4[0];
                csacc[v].y += f4[1];
                csacc[v].z += f4[2];
                csacc[v].w += f4[3];
            }
            // row scale: the power of two that puts the largest entry into [1, 2) (exact)
            emax = group_max<LPR>(emax);
            const float scale = __int_as_float((254 - (__float_as_int(emax) >> 23)) << 23);
#pragma unroll
            for (int v = 0; v < VPL; v++) {
                en[i][v].x *= scale;
                en[i][v].y *= scale;
                en[i][v].z *= scale;
                en[i][v].w *= scale;
            }
            fs[i] = group_sum<LPR>(fsum);
        }
#pragma unroll
        for (int i = 0; i < R; i++) {
            if (live[i]) {
                const size_t o4 = (size_t)row[i] * (LD / 4);
#pragma unroll
                for (int v = 0; v < VPL; v++) {
                    reinterpret_cast<float4 *>(e_tab)[o4 + v * LPR + j] = en[i][v];
                    if (mine) {
                        if (fac) reinterpret_cast<float4 *>(fac)[o4 + v * LPR + j] = fc[i][v];
                        if (shp) reinterpret_cast<float4 *>(shp)[o4 + v * LPR + j] = sv[i][v];
                    }
                }
                if (mine && j == 0) rs[row[i]] = add_rte + fs[i];
            }
        }
    }
    // per-block column sums of the means: fold the wave's groups, then the block's waves
#pragma unroll
    for (int v = 0; v < VPL; v++) {
#pragma unroll
        for (int m = LPR; m < WAVE; m <<= 1) {
            csacc[v].x += __shfl_xor(csacc[v].x, m);
            csacc[v].y += __shfl_xor(csacc[v].y, m);
            csacc[v].z += __shfl_xor(csacc[v].z, m);
            csacc[v].w += __shfl_xor(csacc[v].w, m);
        }
        if (g == 0) {
            const int c = (v * LPR + j) * 4;
            red[wid][c + 0] = csacc[v].x;
            red[wid][c + 1] = csacc[v].y;
            red[wid][c + 2] = csacc[v].z;
            red[wid][c + 3] = csacc[v].w;
        }
    }
    __syncthreads();
    const size_t blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    for (int c = threadIdx.x; c < LD; c += BLOCK) {
        float t = red[0][c];
#pragma unroll
        for (int w2 = 1; w2 < WPB; w2++) t += red[w2][c];
        cs_partial[blk * LD + c] = t;
    }
}

template <int LD>
__global__ __launch_bounds__(BLOCK) void colsum_kernel(const float *__restrict__ tab, int64_t nrows,
                                                       float *__restrict__ cs_partial) {
    constexpr int CPL = (LD + WAVE - 1) / WAVE;
    __shared__ float red[WPB][LD];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * WPB;
    float csacc[CPL];
#pragma unroll
    for (int q = 0; q < CPL; q++) csacc[q] = 0.f;
    for (int64_t r = (int64_t)blockIdx.x * WPB + wid; r < nrows; r += nwaves) {
#pragma unroll
        for (int q = 0; q < CPL; q++) {
            const int c = lane + WAVE * q;
            if (c < LD) csacc[q] += tab[(size_t)r * LD + c];
        }
    }
#pragma unroll
    for (int q = 0; q < CPL; q++) {
        const int c = lane + WAVE * q;
        if (c < LD) red[wid][c] = csacc[q];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < LD; c += BLOCK) {
        float t = red[0][c];
#pragma unroll
        for (int w2 = 1; w2 < WPB; w2++) t += red[w2][c];
        cs_partial[(size_t)blockIdx.x * LD + c] = t;
    }
}

// tab.sum(axis=0) in numpy's OWN order (PXI:236,255: Beta.sum(axis=0), Theta.sum(axis=0) of float32 arrays): numpy adds the
// rows one after the other in float32 -- out[c] = fl(... fl(fl(a[0][c] + a[1][c]) + a[2][c]) ...), verified against a
// row-by-row loop -- so one lane per column walking the rows in sequence reproduces the reference's sums BIT FOR BIT
// (HPF_COLSUM_ORDER=reference: the mode in which the HIP path holds 1e-4 against the reference itself at every size and
// horizon, tests/test_hip_parity.py::test_large_vs_golden; the default sums per-block partials in double, fixed order).
// A chain of nrows dependent adds cannot be spread over the chip, but it can be FED: fed straight from global memory by the
// lane that adds (round 5) it ran at the latency of its 64 loads in flight -- 84 cycles per row, 7 ms per 2e5-row table.
// Here a workgroup owns SEQ_COLS = 16 columns (ld/16 workgroups, each on its own CU): all 8 waves fetch the next tile of
// SEQ_ROWS rows x 64 bytes into registers while wave 0 walks the current tile out of LDS, stored TRANSPOSED (row index
// fastest) so that one ds_read_b128 hands a lane four consecutive rows of its column.  Measured: 4.25 ns per row whatever
// the width (1M x 64: 4.25 ms, 380k x 64: 1.6 ms, 200k x 64: 0.85 ms -- 8x the round-5 kernel; profiles/r06_colsum_sequential.txt)
// against a floor of 3.3 ns: a DEPENDENT v_add_f32 issues every 8 cycles on this chip, and a sequential float32 sum is
// nothing but dependent adds.  Rows past the end are stored as +0 (x + 0 = x exactly).
constexpr int SEQ_COLS = 16, SEQ_ROWS = 512, SEQ_PITCH = SEQ_ROWS + 4;      // (pitch 516: conflict-free transposed stores)
constexpr int SEQ_THREADS = 512, SEQ_G = 16;       // SEQ_G groups of 4 rows are read while the previous SEQ_G are added
__global__ __launch_bounds__(SEQ_THREADS) void colsum_sequential_kernel(const float *__restrict__ tab, int64_t nrows, int ld,
                                                                        float *__restrict__ cs_out) {
    __shared__ __attribute__((aligned(16))) float tile[2][SEQ_COLS][SEQ_PITCH];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid >> 6;
    const int c0 = blockIdx.x * SEQ_COLS;
    // thread t fetches the float4s (row = t / 4 + 128 h, columns c0 + 4*(t % 4) ..), h = 0..3, of a tile
    constexpr int NH = SEQ_ROWS * 4 / SEQ_THREADS, RSTEP = SEQ_THREADS / 4;
    const int rq = tid >> 2, cq = tid & 3;
    const float *src = tab + c0 + 4 * cq;
    const int64_t ntiles = (nrows + SEQ_ROWS - 1) / SEQ_ROWS;
    float4 pre[NH];
    auto fetch = [&](int64_t t) {
#pragma unroll
        for (int h = 0; h < NH; h++) {
            const int64_t r = t * SEQ_ROWS + rq + RSTEP * h;
            pre[h] = (r < nrows) ? ld4(reinterpret_cast<const float4 *>(src + (size_t)r * ld)) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int h = 0; h < NH; h++) {
            const int r = rq + RSTEP * h;
            tile[buf][4 * cq + 0][r] = pre[h].x;
            tile[buf][4 * cq + 1][r] = pre[h].y;
            tile[buf][4 * cq + 2][r] = pre[h].z;
            tile[buf][4 * cq + 3][r] = pre[h].w;
        }
    };
    float s = 0.f;                     // (0 + a[0] is a[0]: the same chain as starting from the first row)
    if (ntiles > 0) {
        fetch(0);
        stage(0);
    }
    __syncthreads();
    for (int64_t t = 0; t < ntiles; t++) {
        const int buf = (int)(t & 1);
        if (t + 1 < ntiles) fetch(t + 1);
        if (wid == 0 && lane < SEQ_COLS) {
            const float4 *col = reinterpret_cast<const float4 *>(&tile[buf][lane][0]);
            // A dependent v_add_f32 issues every 8 cycles (4 to issue + 4 until its result can feed the next): the chain's
            // own pace is 8 cycles per row, and each add leaves a 4-cycle issue slot free.  Two register sets alternate:
            // while one is added, the reads of the other are issued INTO those slots -- one ds_read_b128 after every four
            // adds, pinned with scheduling-group barriers (left to itself the compiler reads 14 groups, adds them, and only
            // then reads again: the reads' issue slots and the LDS latency on top of the chain, 9.8 cycles per row)
            float4 A[SEQ_G], B[SEQ_G];
#pragma unroll
            for (int g = 0; g < SEQ_G; g++) A[g] = col[g];
#pragma unroll 1
            for (int r4 = 0; r4 < SEQ_ROWS / 4; r4 += 2 * SEQ_G) {
#pragma unroll
                for (int g = 0; g < SEQ_G; g++) {
                    B[g] = col[r4 + SEQ_G + g];
                    s = __fadd_rn(s, A[g].x);
                    s = __fadd_rn(s, A[g].y);
                    s = __fadd_rn(s, A[g].z);
                    s = __fadd_rn(s, A[g].w);
                }
#pragma unroll
                for (int g = 0; g < SEQ_G; g++) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // one DS read
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);      // four VALU
                }
                __builtin_amdgcn_sched_barrier(0);
                const int nx = (r4 + 2 * SEQ_G < SEQ_ROWS / 4) ? r4 + 2 * SEQ_G : 0;      // (last round: a harmless re-read)
#pragma unroll
                for (int g = 0; g < SEQ_G; g++) {
                    A[g] = col[nx + g];
                    s = __fadd_rn(s, B[g].x);
                    s = __fadd_rn(s, B[g].y);
                    s = __fadd_rn(s, B[g].z);
                    s = __fadd_rn(s, B[g].w);
                }
#pragma unroll
                for (int g = 0; g < SEQ_G; g++) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (t + 1 < ntiles) stage(buf ^ 1);
        __syncthreads();
    }
    if (wid == 0 && lane < SEQ_COLS) cs_out[c0 + lane] = s;
}


// cs_out[c] = sum_b cs_partial[b][c] in double, fixed order.  ONE workgroup per FOUR columns: a thread sums its rows
// (b = t, t + 1024, ...) of the block's float4 column group, the 1024 chains are folded by a butterfly inside each wave and
// in order across the 16 waves.  A single workgroup per 64 columns read the sweep's 4096 partial rows (1 MB) at the ~40 GB/s
// one CU can pull -- 26 us, twice per iteration on the critical path, however wide or many its loads; 16 workgroups read
// 64 KB each (the row's other columns come out of L2 for the others).
// peers != null (direct exchange): the column sums of ALL ranks -- each rank publishes its value of a column as an
// {value, epoch} granule in every peer's control block and sums the N granules it receives in rank order, so every rank
// ends with the same floats (the k-float all-reduce of the iteration without a collective library)
__global__ __launch_bounds__(1024) void colsum_reduce_kernel(const float *__restrict__ cs_partial, int nblk,
                                                             float *__restrict__ cs_out, int ld,
                                                             const hpf_p2p::Peers *__restrict__ peers, int which,
                                                             uint32_t epoch, uint32_t then_wait_kinds, int then_wait_self) {
    __shared__ double red[16][4];
    const int c0 = blockIdx.x * 4;          // (ld is a multiple of 4: grid = ld / 4)
    const float *col = cs_partial + c0;
    double s4[4] = {0.0, 0.0, 0.0, 0.0};
    int b = threadIdx.x;
    for (; b + 1024 * 3 < nblk; b += 1024 * 4) {
        float4 p[4];
#pragma unroll
        for (int u = 0; u < 4; u++) p[u] = *reinterpret_cast<const float4 *>(col + (size_t)(b + 1024 * u) * ld);
#pragma unroll
        for (int u = 0; u < 4; u++) {
            s4[0] += (double)p[u].x;
            s4[1] += (double)p[u].y;
            s4[2] += (double)p[u].z;
            s4[3] += (double)p[u].w;
        }
    }
    for (; b < nblk; b += 1024) {
        const float4 p = *reinterpret_cast<const float4 *>(col + (size_t)b * ld);
        s4[0] += (double)p.x;
        s4[1] += (double)p.y;
        s4[2] += (double)p.z;
        s4[3] += (double)p.w;
    }
#pragma unroll
    for (int e = 0; e < 4; e++) {
#pragma unroll
        for (int m = 1; m < WAVE; m <<= 1) s4[e] += __shfl_xor(s4[e], m);
    }
    if ((threadIdx.x & (WAVE - 1)) == 0) {
#pragma unroll
        for (int e = 0; e < 4; e++) red[threadIdx.x >> 6][e] = s4[e];
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int c = c0 + threadIdx.x;
        double t = red[0][threadIdx.x];
#pragma unroll
        for (int q = 1; q < 16; q++) t += red[q][threadIdx.x];
        float out = (float)t;
        if (peers) {
            const hpf_p2p::Peers pp = *peers;
            hpf_p2p::vec_publish(pp, which, epoch, ld, c, out);
            out = hpf_p2p::vec_collect(pp, which, epoch, ld, c);
        }
        cs_out[c] = out;
    }
    // direct exchange: this small launch also does the waiting for the large-grid launch behind it on the stream (flag
    // kinds of the bit mask from every peer, then_wait_self from this rank itself) -- a grid of polling workgroups would
    // sit on the wave slots the flags' producers need
    if (peers && blockIdx.x == 0 && (then_wait_kinds || then_wait_self >= 0)) {      // (ONE workgroup waits)
        const hpf_p2p::Peers pp = *peers;
        for (int kind = 0; kind < HPF_P2P_NKINDS; kind++)
            if ((then_wait_kinds >> kind) & 1u) hpf_p2p::block_acquire(pp, kind, epoch, 0xFFFFFFFFu);
        if (then_wait_self >= 0) hpf_p2p::block_acquire_self(pp, then_wait_self, epoch);
    }
}

struct FactoredRate {     // rte[r][c] = top / rs[r] + cs[c] (rank-1), when the [rows][ld] table is not kept; rs == null: table
    const float *rs, *cs;
    float top;
};

// One row of E in two phases, so that a wave can have the NEXT row's loads in flight while it works on this one: with
// one row at a time the flagged-row launches of an SVI step (a few per cent of a table, k = 200) ran at the latency of
// one dependent load chain per row -- 0.29 ms for 376k rows, a quarter of their memory rate.
template <int LD>
struct ExpectIn {
    static constexpr int CPL = (LD + WAVE - 1) / WAVE;
    float sh[CPL], rt[CPL];
};

template <int LD>
__device__ __forceinline__ void expect_load(const float *__restrict__ shp, const float *__restrict__ rte, int64_t r, int k,
                                            int lane, const FactoredRate fr, float rs_r, ExpectIn<LD> &in) {
    // rs_r: fr.rs[r] (factored rate), fetched by the caller -- coalesced for 64 rows at a time where it can
    constexpr int CPL = ExpectIn<LD>::CPL;
    const float base = fr.rs ? fr.top / rs_r : 0.f;
#pragma unroll
    for (int q = 0; q < CPL; q++) {
        const int c = lane + WAVE * q;
        in.sh[q] = (c < k) ? shp[(size_t)r * LD + c] : 1.f;
        if (fr.rs)
            in.rt[q] = (c < k) ? base + fr.cs[c] : 1.f;
        else
            in.rt[q] = (c < k) ? rte[(size_t)r * LD + c] : 1.f;
    }
}

template <int LD>
__device__ __forceinline__ void expect_finish(float *__restrict__ e, int64_t r, int k, int lane, const ExpectIn<LD> &in,
                                              float *__restrict__ rte_out = nullptr) {
    constexpr int CPL = ExpectIn<LD>::CPL;
    if (rte_out) {       // (the factored rate just formed, kept as the table row: pads zero, as every table holds them)
#pragma unroll
        for (int q = 0; q < CPL; q++) {
            const int c = lane + WAVE * q;
            if (c < LD) rte_out[(size_t)r * LD + c] = (c < k) ? in.rt[q] : 0.f;
        }
    }
    double ev[CPL];
    int ehi = 0;
#pragma unroll
    for (int q = 0; q < CPL; q++) {
        const int c = lane + WAVE * q;
        ev[q] = (c < k) ? expect_ratio(in.sh[q], in.rt[q]) : 0.0;
        ehi = max(ehi, __double2hiint(ev[q]));
    }
    const double inv = row_pow2_scale(ehi);
#pragma unroll
    for (int q = 0; q < CPL; q++) {
        const int c = lane + WAVE * q;
        if (c < LD) e[(size_t)r * LD + c] = (c < k) ? (float)(ev[q] * inv) : 0.f;
    }
}

template <int LD>
__global__ __launch_bounds__(BLOCK) void expect_kernel(const float *__restrict__ shp, const float *__restrict__ rte,
                                                       float *__restrict__ e, const int64_t *__restrict__ row_list,
                                                       const uint8_t *__restrict__ flag, int64_t nrows, int k,
                                                       const FactoredRate fr, float *__restrict__ rte_out) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * WPB;
    ExpectIn<LD> cur, nxt;
    if (flag && !row_list) {
        // the flagged rows of a stochastic step, a few per cent of the table: a wave reads 64 flags at once and visits the
        // set ones (a flag test per row would walk the whole table at one dependent load per row)
        for (int64_t g = ((int64_t)blockIdx.x * WPB + wid) * WAVE; g < nrows; g += nwaves * WAVE) {
            unsigned long long m = __ballot(g + lane < nrows && flag[g + lane] != 0);
            if (!m) continue;
            const float rs_l = (fr.rs && g + lane < nrows) ? fr.rs[g + lane] : 1.f;   // the 64 rows' scalars, one load
            int b = __builtin_ctzll(m);
            m &= m - 1;
            expect_load<LD>(shp, rte, g + b, k, lane, fr, __shfl(rs_l, b), cur);
            while (true) {
                const bool more = m != 0;
                int b2 = 0;
                if (more) {
                    b2 = __builtin_ctzll(m);
                    m &= m - 1;
                    expect_load<LD>(shp, rte, g + b2, k, lane, fr, __shfl(rs_l, b2), nxt);
                }
                expect_finish<LD>(e, g + b, k, lane, cur, rte_out);
                if (!more) break;
                cur = nxt;
                b = b2;
            }
        }
        return;
    }
    // listed rows (optionally filtered by a flag), or all rows: the same one-row look-ahead
    int64_t t = (int64_t)blockIdx.x * WPB + wid;
    auto next_row = [&](int64_t &tt) -> int64_t {       // the next row this wave works on, or -1
        for (; tt < nrows; tt += nwaves) {
            const int64_t r = row_list ? row_list[tt] : tt;
            if (flag && !flag[r]) continue;
            tt += nwaves;
            return r;
        }
        return -1;
    };
    int64_t r = next_row(t);
    if (r < 0) return;
    expect_load<LD>(shp, rte, r, k, lane, fr, fr.rs ? fr.rs[r] : 1.f, cur);
    while (true) {
        const int64_t r2 = next_row(t);
        if (r2 >= 0) expect_load<LD>(shp, rte, r2, k, lane, fr, fr.rs ? fr.rs[r2] : 1.f, nxt);
        expect_finish<LD>(e, r, k, lane, cur, rte_out);
        if (r2 < 0) break;
        cur = nxt;
        r = r2;
    }
}

__global__ __launch_bounds__(BLOCK) void segsum_kernel(const float *__restrict__ part,
                                                       const int64_t *__restrict__ row_seg_ptr,
                                                       const int64_t *__restrict__ row_list, int64_t nrows,
                                                       float *__restrict__ acc, int ld, int acc_ld, int acc_by_row) {
    // acc rows have stride acc_ld <= ld (acc_ld = k packs the all-reduce payload: pads are zero anyway)
    const int64_t total = nrows * (int64_t)acc_ld;
    for (int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x; t < total; t += (int64_t)gridDim.x * BLOCK) {
        const int64_t rr = t / acc_ld;
        const int c = (int)(t - rr * acc_ld);
        const int64_t r = row_list ? row_list[rr] : rr;
        float a = 0.f;
        int64_t sg = row_seg_ptr[r];
        const int64_t s1 = row_seg_ptr[r + 1];
        for (; sg + 8 <= s1; sg += 8) {  // same fold order as row_finalize_kernel
            float p[8];
#pragma unroll
            for (int u = 0; u < 8; u++) p[u] = part[(size_t)(sg + u) * ld + c];
            a += ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
        }
        for (; sg < s1; sg++) a += part[(size_t)sg * ld + c];
        acc[acc_by_row ? (size_t)r * acc_ld + c : (size_t)t] = a;
    }
}

// ----------------------------------------------------------------------------------------
// fold-in of ONE user with the item parameters fixed (calc_user_factors, PXI:476-520): the whole local coordinate ascent
// -- E row of the user, phi-sums over the user's items, rate / shape / mean updates, the stopping rule (PXI:505-517) --
// as ONE workgroup looping on the device, instead of three launches and a host-side convergence check per round (the
// round trips made a 40-item fold-in take 1.8 ms).  Wave w sweeps the nonzeros w, w+4, ... eight at a time (lane <->
// column, 64*CPL columns); every wave keeps the k-vector state redundantly, so only the accumulators cross waves.
// ----------------------------------------------------------------------------------------
template <int LD>
__global__ __launch_bounds__(BLOCK) void fold_in_kernel(const int32_t *__restrict__ idx, const float *__restrict__ y,
                                                        int64_t n, const float *__restrict__ eB,
                                                        const float *__restrict__ cs_other, float *__restrict__ shp,
                                                        float *__restrict__ rte, float *__restrict__ fac,
                                                        float *__restrict__ e_last, int32_t *__restrict__ rounds,
                                                        float prior, float top, float add, float rs, float stop_thr,
                                                        int maxiter, int k) {
    constexpr int CPL = (LD + WAVE - 1) / WAVE;
    constexpr int U = 8;
    __shared__ float red[WPB][LD];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float gs[CPL], gr[CPL], th_prev[CPL], th[CPL], et[CPL], cso[CPL];
#pragma unroll
    for (int q = 0; q < CPL; q++) {
        const int c = lane + WAVE * q;
        const bool in = c < k;
        gs[q] = in ? shp[c] : 1.f;
        gr[q] = in ? rte[c] : 1.f;
        th_prev[q] = in ? fac[c] : 0.f;
        th[q] = th_prev[q];
        cso[q] = in ? cs_other[c] : 0.f;
        et[q] = 0.f;
    }
    int it = 0;
    while (it < maxiter) {
        // E row from the current shape / rate (PXI:505: phi is computed from the Gamma BEFORE its update)
        double ev[CPL];
        int ehi = 0;
#pragma unroll
        for (int q = 0; q < CPL; q++) {
            ev[q] = (lane + WAVE * q < k) ? expect_ratio(gs[q], gr[q]) : 0.0;
            ehi = max(ehi, __double2hiint(ev[q]));
        }
        const double inv = row_pow2_scale(ehi);
#pragma unroll
        for (int q = 0; q < CPL; q++) et[q] = (float)(ev[q] * inv);
        // phi-sums over the user's items: acc_c = sum_n y_n * eB[i_n][c] / <et, eB[i_n]>
        float acc[CPL];
#pragma unroll
        for (int q = 0; q < CPL; q++) acc[q] = 0.f;
        for (int64_t base = (int64_t)wid * U; base < n; base += (int64_t)WPB * U) {
            float o[U][CPL], yy[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int64_t t = base + u;
                const bool live = t < n;
                const int32_t r = live ? idx[t] : 0;
                yy[u] = live ? y[t] : 0.f;
#pragma unroll
                for (int q = 0; q < CPL; q++) {
                    const int c = lane + WAVE * q;
                    o[u][q] = (c < LD) ? eB[(size_t)r * LD + c] : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                float p = et[0] * o[u][0];
#pragma unroll
                for (int q = 1; q < CPL; q++) p = fmaf(et[q], o[u][q], p);
                const float s = wave_sum(p);
                const float w = (yy[u] > 0.f) ? yy[u] * __builtin_amdgcn_rcpf(s) : 0.f;
#pragma unroll
                for (int q = 0; q < CPL; q++) acc[q] = fmaf(w, o[u][q], acc[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < CPL; q++) {
            const int c = lane + WAVE * q;
            if (c < LD) red[wid][c] = acc[q];
        }
        __syncthreads();
        float ssum = 0.f, d2 = 0.f;
#pragma unroll
        for (int q = 0; q < CPL; q++) {
            const int c = lane + WAVE * q;
            const bool in = c < k;
            float a = 0.f;
            if (c < LD) {
                a = red[0][c];
#pragma unroll
                for (int w2 = 1; w2 < WPB; w2++) a += red[w2][c];        // fixed order: deterministic
            }
            gr[q] = in ? top / rs + cso[q] : 1.f;                           // PXI:507
            gs[q] = in ? prior + et[q] * a : 1.f;                          // PXI:508: a + phi.sum(axis=0)
            th[q] = in ? gs[q] / gr[q] : 0.f;
            ssum += th[q];
            const float d = th[q] - th_prev[q];
            d2 = fmaf(d, d, d2);
        }
        __syncthreads();                                                   // (red is rewritten by the next round)
        rs = add + wave_sum(ssum);                                         // PXI:510
        d2 = wave_sum(d2);
        ++it;
        if (sqrtf(d2) < stop_thr) break;                                   // PXI:512-513 (uniform over the block)
#pragma unroll
        for (int q = 0; q < CPL; q++) th_prev[q] = th[q];
    }
    if (wid == 0) {
#pragma unroll
        for (int q = 0; q < CPL; q++) {
            const int c = lane + WAVE * q;
            if (c < LD) {
                const bool in = c < k;
                shp[c] = in ? gs[q] : 0.f;
                rte[c] = in ? gr[q] : 0.f;
                fac[c] = in ? th[q] : 0.f;
                e_last[c] = in ? et[q] : 0.f;
            }
        }
        if (lane == 0) rounds[0] = it;
    }
}

// ----------------------------------------------------------------------------------------
// stochastic-VI row kernels (one wavefront per row, lane <-> factor): the numpy statements the reference
// executes around update_phi_csr in an SVI batch / partial_fit (PXI:300-325, 352-377, 443-473)
// ----------------------------------------------------------------------------------------
// shp[r] = w_new * (prior + e[r] (*) acc[t]) + w_old * shp[r]   for r = row_list[t]
//   batch side:  w_new = 1, w_old = 0            (reset to the prior + this batch's phi, PXI:304,308-314)
//   other side:  w_new = step*multiplier, w_old = 1-step          (PXI:316 / PXI:368)
template <int LD>
__global__ __launch_bounds__(BLOCK) void svi_shape_rows_kernel(const int64_t *__restrict__ row_list, int64_t nrows,
                                                               const float *__restrict__ acc,
                                                               const float *__restrict__ e, float *__restrict__ shp,
                                                               float prior, float w_new, float w_old, int k,
                                                               int acc_by_row) {
    constexpr int CPL = (LD + WAVE - 1) / WAVE;
    const int lane = threadIdx.x & (WAVE - 1);
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * WPB;
    for (int64_t t = (int64_t)blockIdx.x * WPB + wid; t < nrows; t += nwaves) {
        const int64_t r = row_list[t];
        const int64_t ar = acc_by_row ? r : t;   // accumulator table indexed by row id, or aligned with the list
#pragma unroll
        for (int q = 0; q < CPL; q++) {
            const int c = lane + WAVE * q;
            if (c < k) {
                const size_t o = (size_t)r * LD + c;
                const float a = acc ? acc[(size_t)ar * LD + c] : 0.f;
                const float fresh = fmaf(e[o], a, prior);
                shp[o] = blend_shape(w_new, fresh, w_old, shp[o]);
            }
        }
    }
}

// whole-side refresh: [rte = top/rs + cs_other]  ->  fac = shp/rte  ->  [rs = step*(add + sum_k fac) + (1-step)*rs]
// plus per-block column sums of fac (Theta[:,:] = Gamma_shp/Gamma_rte etc., PXI:300,318,322,352,370,374,472-473)
template <int LD>
__global__ __launch_bounds__(BLOCK) void svi_refresh_kernel(int64_t nrows, const float *__restrict__ shp,
                                                            float *__restrict__ rte, float *__restrict__ fac,
                                                            float *__restrict__ rs,
                                                            const float *__restrict__ cs_other,
                                                            float *__restrict__ cs_partial, float top, float add,
                                                            float step, float step_prev, int refresh_rte,
                                                            int blend_rs, int k) {
    constexpr int CPL = (LD + WAVE - 1) / WAVE;
    __shared__ float red[WPB][LD];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * WPB;
    float csl[CPL], csacc[CPL];
#pragma unroll
    for (int q = 0; q < CPL; q++) {
        const int c = lane + WAVE * q;
        csl[q] = (refresh_rte && c < k) ? cs_other[c] : 0.f;
        csacc[q] = 0.f;
    }
    // a streaming kernel with one short row per wavefront: R rows are loaded before any is processed, so a wave
    // keeps R*LD*4 bytes of reads in flight (the row loop is otherwise latency-bound at 16 waves per CU)
    constexpr int R = (CPL <= 4) ? 4 : (CPL <= 8 ? 2 : 1);
    for (int64_t r0 = (int64_t)blockIdx.x * WPB + wid; r0 < nrows; r0 += R * nwaves) {
        float sv[R][CPL], rv[R][CPL], rs_old[R];
#pragma unroll
        for (int i = 0; i < R; i++) {
            const int64_t r = r0 + i * nwaves;
            const bool live = r < nrows;
            rs_old[i] = live ? rs[r] : 1.f;
#pragma unroll
            for (int q = 0; q < CPL; q++) {
                const int c = lane + WAVE * q;
                const size_t o = (size_t)r * LD + c;
                sv[i][q] = (live && c < k) ? shp[o] : 0.f;
                rv[i][q] = (live && c < k && !refresh_rte) ? rte[o] : 1.f;
            }
        }
#pragma unroll
        for (int i = 0; i < R; i++) {
            const int64_t r = r0 + i * nwaves;
            if (r >= nrows) break;
            const float base = top / rs_old[i];
            float fsum = 0.f;
#pragma unroll
            for (int q = 0; q < CPL; q++) {
                const int c = lane + WAVE * q;
                if (c < LD) {
                    const size_t o = (size_t)r * LD + c;
                    float f = 0.f;
                    if (c < k) {
                        float rt = rv[i][q];
                        if (refresh_rte) {
                            rt = base + csl[q];
                            rte[o] = rt;
                        }
                        f = sv[i][q] / rt;
                    }
                    fac[o] = f;
                    fsum += f;
                    csacc[q] += f;
                }
            }
            if (blend_rs) {
                fsum = wave_sum(fsum);
                if (lane == 0) rs[r] = blend_rate(step, add, fsum, step_prev, rs_old[i]);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < CPL; q++) {
        const int c = lane + WAVE * q;
        if (c < LD) red[wid][c] = csacc[q];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < LD; c += BLOCK) {
        float t = red[0][c];
#pragma unroll
        for (int w2 = 1; w2 < WPB; w2++) t += red[w2][c];
        cs_partial[(size_t)blockIdx.x * LD + c] = t;
    }
}

// The batch side of a LAZY epoch step on its own (svi_side_kernel with rate_mode 0, no rate / mean table stored, no E row
// written -- the same float32 statements in the same order, bit for bit): the pass READS the side's shapes (1 GB at C5's
// user side) and little else, so it is a streaming kernel -- 8 rows' loads in flight per wave, the acc / e rows of the few
// flagged rows (one in sixteen of a C5 user batch) fetched on demand, 16-byte loads (the general kernel's `cond ? p[i] : zero4` had
// been lowered to four dword loads per lane: 137 VGPRs, 3 waves per SIMD, 4 KB in flight per wave, 3.2 TB/s on this pass).
//   flagged rows:  shp = w_new*(prior + e (*) acc[r]) (+ w_old*shp)                     PXI:304-316, 356-368
//   every row:     rte = top/rs_rate + cs_other (not stored);  fac = shp/rte (not stored); per-block column sums of fac
//   rs_mode 2: every row, 1: flagged rows:  rs = step*(add + sum_k fac) + step_prev*rs   PXI:324-325, 376-377
// The next group of VR rows is requested BEFORE the current group's divisions are issued (two groups alternate in registers)
// and the divisions go through div2_normal: 0.267 -> 0.227 ms on a 1M-row table, where a plain read of the table (torch.sum)
// takes 0.263 ms for 16/13 of the bytes -- the pass runs at the box's streaming rate (profiles/r06_svi_side_probe.txt).
// done_flag != 0: rows whose flag EQUALS it were finished by the sweep that formed their phi-sums (sweep_kernel MODE 3):
// nothing of theirs is read, written or summed here.
template <int LD>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(4, 8)))
void svi_lazy_batch_side_kernel(int64_t nrows, const uint8_t *__restrict__ flag, const float *__restrict__ acc,
                                   const float *__restrict__ e, float *__restrict__ shp, float *__restrict__ rs,
                                   const float *__restrict__ cs_other, float *__restrict__ cs_partial, float prior,
                                   float w_new, float w_old, float top, float add, float step, float step_prev,
                                   int rs_mode, int k, const float *__restrict__ rs_rate,
                                   float *__restrict__ rs_prev_out, int done_flag) {
    static_assert(LD >= 4 * WAVE, "float4-per-lane rows");
    constexpr int VPL = LD / (4 * WAVE);
    constexpr int VR = (VPL == 1) ? 4 : (VPL == 2 ? 2 : 1);
    __shared__ float red[WPB][LD];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * WPB;
    float4 cs4[VPL], acc4[VPL];
    bool act[VPL];
    const int nq4 = ((k + 15) / 16) * 4;      // float4s of a row in sectors that hold columns (the padding is not read)
#pragma unroll
    for (int v = 0; v < VPL; v++) {
        const int c = (v * WAVE + lane) * 4;
        act[v] = v * WAVE + lane < nq4;
        cs4[v] = make_float4(c < k ? cs_other[c] : 0.f, c + 1 < k ? cs_other[c + 1] : 0.f, c + 2 < k ? cs_other[c + 2] : 0.f,
                             c + 3 < k ? cs_other[c + 3] : 0.f);
        acc4[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t g = ((int64_t)blockIdx.x * WPB + wid) * WAVE; g < nrows; g += nwaves * WAVE) {
        const int64_t rl = g + lane;
        const bool lv = rl < nrows;
        const int fv = (lv && flag) ? (int)flag[rl] : 0;
        const unsigned long long dmask = __ballot(done_flag != 0 && fv == done_flag);
        const unsigned long long fmask = __ballot(fv != 0) & ~dmask;
        const float rs_l = lv ? rs[rl] : 1.f;
        const float rsr_l = (lv && rs_rate) ? rs_rate[rl] : rs_l;
        float rs_new_l = rs_l;
        const int cnt = (int)min((int64_t)WAVE, nrows - g);
        const float4 *gp = reinterpret_cast<const float4 *>(shp) + (size_t)g * (LD / 4);
        auto request = [&](int b0, float4 (&sv)[VR][VPL]) {      // rows g + b0 .. g + b0 + VR - 1 (wave-uniform bases)
#pragma unroll
            for (int i = 0; i < VR; i++) {
                const bool live = b0 + i < cnt && ((dmask >> (b0 + i)) & 1ull) == 0;
#pragma unroll
                for (int v = 0; v < VPL; v++) {
                    sv[i][v] = zero4;
                    if (live && act[v]) sv[i][v] = ld4(gp + (size_t)(b0 + i) * (LD / 4) + (v * WAVE + lane));
                }
            }
        };
        auto process = [&](int b0, const float4 (&sv)[VR][VPL]) {
#pragma unroll
            for (int i = 0; i < VR; i++) {
                if (b0 + i >= cnt) break;
                if (((dmask >> (b0 + i)) & 1ull) != 0) continue;          // (wave-uniform)
                const bool fl = ((fmask >> (b0 + i)) & 1ull) != 0;      // (wave-uniform)
                const float rs_old = __shfl(rs_l, b0 + i);
                const float base = top / __shfl(rsr_l, b0 + i);
                const size_t ou = (size_t)(g + b0 + i) * (LD / 4);      // (uniform)
                float fsum = 0.f;
#pragma unroll
                for (int v = 0; v < VPL; v++) {
                    const int c = (v * WAVE + lane) * 4;
                    float s4[4] = {sv[i][v].x, sv[i][v].y, sv[i][v].z, sv[i][v].w};
                    if (fl) {
                        float4 aq = zero4, eq = zero4;
                        if (act[v]) {
                            aq = ld4(reinterpret_cast<const float4 *>(acc) + ou + (v * WAVE + lane));
                            eq = ld4(reinterpret_cast<const float4 *>(e) + ou + (v * WAVE + lane));
                        }
                        const float a4[4] = {aq.x, aq.y, aq.z, aq.w}, e4[4] = {eq.x, eq.y, eq.z, eq.w};
#pragma unroll
                        for (int e2 = 0; e2 < 4; e2++) {
                            const float fresh = fmaf(e4[e2], a4[e2], prior);
                            const float sx = blend_shape(w_new, fresh, w_old, s4[e2]);
                            s4[e2] = (c + e2 < k) ? sx : 0.f;          // (pad columns: what the tables hold there)
                        }
                        (reinterpret_cast<float4 *>(shp) + ou)[v * WAVE + lane] = make_float4(s4[0], s4[1], s4[2], s4[3]);
                    }
                    const hpf_v2f r01 = {base + cs4[v].x, base + cs4[v].y}, r23 = {base + cs4[v].z, base + cs4[v].w};
                    const hpf_v2f q01 = div2_normal(hpf_v2f{s4[0], s4[1]}, r01), q23 = div2_normal(hpf_v2f{s4[2], s4[3]}, r23);
                    const float f4[4] = {(c + 0 < k) ? q01.x : 0.f, (c + 1 < k) ? q01.y : 0.f, (c + 2 < k) ? q23.x : 0.f,
                                         (c + 3 < k) ? q23.y : 0.f};
#pragma unroll
                    for (int e2 = 0; e2 < 4; e2++) fsum += f4[e2];
                    acc4[v].x += f4[0];
                    acc4[v].y += f4[1];
                    acc4[v].z += f4[2];
                    acc4[v].w += f4[3];
                }
                if (rs_mode == 2 || (rs_mode == 1 && fl)) {
                    fsum = wave_sum(fsum);
                    if (lane == b0 + i) rs_new_l = blend_rate(step, add, fsum, step_prev, rs_old);
                }
            }
        };
        float4 bufA[VR][VPL], bufB[VR][VPL];
        request(0, bufA);
        for (int b0 = 0; b0 < cnt; b0 += 2 * VR) {
            if (b0 + VR < cnt) request(b0 + VR, bufB);
            process(b0, bufA);
            if (b0 + 2 * VR < cnt) request(b0 + 2 * VR, bufA);
            if (b0 + VR < cnt) process(b0 + VR, bufB);
        }
        const bool mine = lv && ((dmask >> lane) & 1ull) == 0;      // (a finished row's scalars were written by its sweep)
        if (mine && rs_prev_out) rs_prev_out[rl] = rsr_l;
        if (mine && (rs_mode == 2 || (rs_mode == 1 && ((fmask >> lane) & 1ull) != 0))) rs[rl] = rs_new_l;
    }
#pragma unroll
    for (int v = 0; v < VPL; v++) reinterpret_cast<float4 *>(&red[wid][0])[v * WAVE + lane] = acc4[v];
    __syncthreads();
    for (int c = threadIdx.x; c < LD; c += BLOCK) {
        float t = red[0][c];
#pragma unroll
        for (int w2 = 1; w2 < WPB; w2++) t += red[w2][c];
        cs_partial[(size_t)blockIdx.x * LD + c] = t;
    }
}

// One pass over ALL rows of a side that applies everything an SVI step does to that side (the statements of
// svi_shape_rows + svi_rate_rows + svi_refresh, row-local, same float32 operations in the same order):
//   flagged rows (flag[r] != 0: the batch's rows / the rows the batch touched):
//       shp = w_new*(prior + e (*) acc[r]) + w_old*shp                                  PXI:304-316, 356-368
//   rate_mode 0 (batch side), every row:   rte = top/rs + cs_other                      PXI:300 / 352
//   rate_mode 1 (other side), flagged:     rte = step*(top/rs + cs_other) + step_prev*rte      PXI:320 / 372
//   every row:                             fac = shp/rte ; per-block column sums        PXI:318,322 / 370,374
//   rs_mode 2: every row, 1: flagged rows: rs = step*(add + sum_k fac) + step_prev*rs   PXI:324-325,376-377,472-473
template <int LD>
__global__ __launch_bounds__(BLOCK) void svi_side_kernel(int64_t nrows, const uint8_t *__restrict__ flag,
                                                         const float *__restrict__ acc, const float *e,
                                                         float *__restrict__ shp, float *__restrict__ rte,
                                                         float *__restrict__ fac, float *__restrict__ rs,
                                                         const float *__restrict__ cs_other,
                                                         float *__restrict__ cs_partial, float prior, float w_new,
                                                         float w_old, float top, float add, float step,
                                                         float step_prev, int rate_mode, int rs_mode, int k,
                                                         const float *__restrict__ rs_rate,
                                                         float *__restrict__ rs_prev_out, float *e_out, int done_flag) {
    // done_flag != 0: rows whose flag EQUALS it were finished elsewhere (the sweep's fused epilogue, sweep_kernel MODE 2) --
    // nothing of theirs is read, written or summed here; the other flagged rows (a split row's flag differs) and the
    // unflagged rows are treated as ever.
    // e_out (may alias e -- which is why `e` carries no __restrict__: the row's loads must stay ordered before its stores, and
    // must not be routed through a read-only path): the step's rows also get their NEW E row, exp(psi(shp))/rte row-scaled, from the shape and rate
    // just formed -- what expect_kernel would compute from the tables at the start of the next step (same function, same
    // inputs, same bits), without reading them back; rows outside the step keep theirs (nothing of theirs changed in a
    // rate_mode 1 pass), so a side whose E table was current for all rows stays current.
    // rte / fac may be null: the table is not stored (rate_mode 0 only for rte).  A batch side's rate is rank-1,
    // rte = top/rs + cs_other, and its means are read through their column sums only, so an epoch keeps the row scalar
    // (rs_prev_out[r] = the rs the rate was formed with) + cs_other instead of two [rows][ld] tables; rs_rate: form the
    // rate from THAT scalar instead of rs (expanding a factored rate later, bit-identically).
    constexpr int CPL = (LD + WAVE - 1) / WAVE;
    __shared__ float red[WPB][LD];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * WPB;
    if constexpr (LD >= 4 * WAVE) {
        // rows of 256+ floats: a lane holds float4s (columns (v*64 + lane)*4 .. +3), VR rows in flight per wave.  The
        // whole-table pass of a lazy user epoch (1 GB of shapes read, nothing stored) ran at 2.2 TB/s; float4 loads
        // alone changed nothing -- what it was waiting for were the per-row single-address loads and single-lane stores
        // (below), tools/svi_side_probe.py: 0.46 -> 0.33 ms, a plain read of the table takes 0.27
        constexpr int VPL = LD / (4 * WAVE);
        constexpr int VR = (VPL == 1) ? 4 : (VPL == 2 ? 2 : 1);
        float4 cs4[VPL], acc4[VPL];
        bool ok[VPL][4], act[VPL];        // act: this lane's float4 lies in a 64-byte sector that holds columns (< k); the
        const int nq4 = ((k + 15) / 16) * 4;   // sectors of pure padding are not READ (k = 200: 13 of a row's 16)
#pragma unroll
        for (int v = 0; v < VPL; v++) {
            const int c = (v * WAVE + lane) * 4;
            act[v] = v * WAVE + lane < nq4;
#pragma unroll
            for (int e2 = 0; e2 < 4; e2++) ok[v][e2] = c + e2 < k;
            cs4[v] = make_float4(ok[v][0] ? cs_other[c] : 0.f, ok[v][1] ? cs_other[c + 1] : 0.f,
                                 ok[v][2] ? cs_other[c + 2] : 0.f, ok[v][3] ? cs_other[c + 3] : 0.f);
            acc4[v] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f), one4 = make_float4(1.f, 1.f, 1.f, 1.f);
        // A wave takes 64 CONSECUTIVE rows: their flags and row scalars are one coalesced load each (lane l holds row
        // g + l's; a row's are read back with a wave-uniform shuffle) and the new scalars go out as one store -- per row
        // they were three single-address loads and up to two single-lane stores next to one 1 KB row load.
        for (int64_t g = ((int64_t)blockIdx.x * WPB + wid) * WAVE; g < nrows; g += nwaves * WAVE) {
            const int64_t rl = g + lane;
            const bool lv = rl < nrows;
            const int fv = (lv && flag) ? (int)flag[rl] : 0;
            const unsigned long long dmask = __ballot(done_flag != 0 && fv == done_flag);
            const unsigned long long fmask = __ballot(fv != 0) & ~dmask;
            const float rs_l = lv ? rs[rl] : 1.f;
            const float rsr_l = (lv && rs_rate) ? rs_rate[rl] : rs_l;
            float rs_new_l = rs_l;
            const int cnt = (int)min((int64_t)WAVE, nrows - g);
            for (int b0 = 0; b0 < cnt; b0 += VR) {
                if (((dmask >> b0) & ((1ull << VR) - 1)) == ((1ull << VR) - 1)) continue;     // (all of them finished elsewhere)
                float4 sv[VR][VPL], rv[VR][VPL], av[VR][VPL], ev[VR][VPL];
                bool fl[VR], live[VR];
#pragma unroll
                for (int i = 0; i < VR; i++) {
                    live[i] = b0 + i < cnt && ((dmask >> (b0 + i)) & 1ull) == 0;
                    fl[i] = live[i] && ((fmask >> (b0 + i)) & 1ull) != 0;
                    const size_t o4 = (size_t)(live[i] ? g + b0 + i : 0) * (LD / 4) + lane;
#pragma unroll
                    for (int v = 0; v < VPL; v++) {
                        sv[i][v] = av[i][v] = ev[i][v] = zero4;
                        rv[i][v] = one4;
                        if (live[i] && act[v]) {       // (ld4: ONE 16-byte load each -- see its comment)
                            sv[i][v] = ld4(reinterpret_cast<const float4 *>(shp) + o4 + v * WAVE);
                            if (rate_mode != 0) rv[i][v] = ld4(reinterpret_cast<const float4 *>(rte) + o4 + v * WAVE);
                            if (fl[i]) {
                                av[i][v] = ld4(reinterpret_cast<const float4 *>(acc) + o4 + v * WAVE);
                                ev[i][v] = ld4(reinterpret_cast<const float4 *>(e) + o4 + v * WAVE);
                            }
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < VR; i++) {
                    if (b0 + i >= cnt) break;
                    if (!live[i]) continue;
                    const int64_t r = g + b0 + i;
                    const float rs_old = __shfl(rs_l, b0 + i);
                    const float base = top / __shfl(rsr_l, b0 + i);
                    const size_t o4 = (size_t)r * (LD / 4) + lane;
                    float fsum = 0.f;
#pragma unroll
                    for (int v = 0; v < VPL; v++) {
                        float s4[4] = {sv[i][v].x, sv[i][v].y, sv[i][v].z, sv[i][v].w};
                        float r4[4] = {rv[i][v].x, rv[i][v].y, rv[i][v].z, rv[i][v].w};
                        const float a4[4] = {av[i][v].x, av[i][v].y, av[i][v].z, av[i][v].w};
                        const float e4[4] = {ev[i][v].x, ev[i][v].y, ev[i][v].z, ev[i][v].w};
                        const float c4[4] = {cs4[v].x, cs4[v].y, cs4[v].z, cs4[v].w};
                        float f4[4];
#pragma unroll
                        for (int e2 = 0; e2 < 4; e2++) {
                            float f = 0.f;
                            if (ok[v][e2]) {
                                float sx = s4[e2];
                                if (fl[i]) {
                                    const float fresh = fmaf(e4[e2], a4[e2], prior);
                                    sx = blend_shape(w_new, fresh, w_old, sx);
                                }
                                float rt = r4[e2];
                                if (rate_mode == 0)
                                    rt = base + c4[e2];
                                else if (fl[i])
                                    rt = blend_rate(step, base, c4[e2], step_prev, rt);
                                f = sx / rt;
                                s4[e2] = sx;
                                r4[e2] = rt;
                            } else {
                                s4[e2] = 0.f;       // (pad columns: what the tables hold there)
                                r4[e2] = 0.f;
                            }
                            f4[e2] = f;
                            fsum += f;
                        }
                        acc4[v].x += f4[0];
                        acc4[v].y += f4[1];
                        acc4[v].z += f4[2];
                        acc4[v].w += f4[3];
                        if (fl[i]) reinterpret_cast<float4 *>(shp)[o4 + v * WAVE] = make_float4(s4[0], s4[1], s4[2], s4[3]);
                        if (rte && (rate_mode == 0 || fl[i]))
                            reinterpret_cast<float4 *>(rte)[o4 + v * WAVE] = make_float4(r4[0], r4[1], r4[2], r4[3]);
                        if (fac) reinterpret_cast<float4 *>(fac)[o4 + v * WAVE] = make_float4(f4[0], f4[1], f4[2], f4[3]);
                        if (e_out && fl[i]) {      // (kept for the E row below)
                            sv[i][v] = make_float4(s4[0], s4[1], s4[2], s4[3]);
                            rv[i][v] = make_float4(r4[0], r4[1], r4[2], r4[3]);
                        }
                    }
                    if (e_out && fl[i]) {          // (wave-uniform)
                        double en[VPL][4];
                        int ehi = 0;
#pragma unroll
                        for (int v = 0; v < VPL; v++) {
                            const float s4[4] = {sv[i][v].x, sv[i][v].y, sv[i][v].z, sv[i][v].w};
                            const float r4[4] = {rv[i][v].x, rv[i][v].y, rv[i][v].z, rv[i][v].w};
#pragma unroll
                            for (int e2 = 0; e2 < 4; e2++) {
                                en[v][e2] = ok[v][e2] ? expect_ratio(s4[e2], r4[e2]) : 0.0;
                                ehi = max(ehi, __double2hiint(en[v][e2]));
                            }
                        }
                        const double inv = row_pow2_scale(ehi);
#pragma unroll
                        for (int v = 0; v < VPL; v++)
                            reinterpret_cast<float4 *>(e_out)[o4 + v * WAVE] =
                                make_float4(ok[v][0] ? (float)(en[v][0] * inv) : 0.f, ok[v][1] ? (float)(en[v][1] * inv) : 0.f,
                                            ok[v][2] ? (float)(en[v][2] * inv) : 0.f, ok[v][3] ? (float)(en[v][3] * inv) : 0.f);
                    }
                    if (rs_mode == 2 || (rs_mode == 1 && fl[i])) {
                        fsum = wave_sum(fsum);
                        if (lane == b0 + i) rs_new_l = blend_rate(step, add, fsum, step_prev, rs_old);
                    }
                }
            }
            const bool mine = lv && ((dmask >> lane) & 1ull) == 0;  // (a finished row's scalars were written by its sweep)
            if (mine && rs_prev_out) rs_prev_out[rl] = rsr_l;
            if (mine && (rs_mode == 2 || (rs_mode == 1 && ((fmask >> lane) & 1ull) != 0))) rs[rl] = rs_new_l;
        }
#pragma unroll
        for (int v = 0; v < VPL; v++)
            reinterpret_cast<float4 *>(&red[wid][0])[v * WAVE + lane] = acc4[v];
        __syncthreads();
        for (int c = threadIdx.x; c < LD; c += BLOCK) {
            float t = red[0][c];
#pragma unroll
            for (int w2 = 1; w2 < WPB; w2++) t += red[w2][c];
            cs_partial[(size_t)blockIdx.x * LD + c] = t;
        }
        return;
    }
    float csl[CPL], csacc[CPL];
#pragma unroll
    for (int q = 0; q < CPL; q++) {
        const int c = lane + WAVE * q;
        csl[q] = (c < k) ? cs_other[c] : 0.f;
        csacc[q] = 0.f;
    }
    constexpr int R = (CPL <= 4) ? 2 : 1;   // rows in flight per wavefront
    for (int64_t r0 = (int64_t)blockIdx.x * WPB + wid; r0 < nrows; r0 += R * nwaves) {
        float sv[R][CPL], rv[R][CPL], av[R][CPL], ev[R][CPL], rs_old[R], rs_rt[R];
        bool fl[R], dn[R];
#pragma unroll
        for (int i = 0; i < R; i++) {       // (flags and scalars of all rows first: see the float4 path)
            const int64_t r = r0 + i * nwaves;
            const bool live = r < nrows;
            const int fv = (live && flag) ? (int)flag[r] : 0;
            dn[i] = done_flag != 0 && fv == done_flag;
            fl[i] = fv != 0 && !dn[i];
            rs_old[i] = live ? rs[r] : 1.f;
            rs_rt[i] = (live && rs_rate) ? rs_rate[r] : rs_old[i];
        }
#pragma unroll
        for (int i = 0; i < R; i++) {
            const int64_t r = r0 + i * nwaves;
            const bool live = r < nrows && !dn[i];
#pragma unroll
            for (int q = 0; q < CPL; q++) {
                const int c = lane + WAVE * q;
                const size_t o = (size_t)r * LD + c;
                const bool lc = live && c < k;
                sv[i][q] = lc ? shp[o] : 0.f;
                rv[i][q] = (lc && rate_mode != 0) ? rte[o] : 1.f;
                av[i][q] = (lc && fl[i]) ? acc[o] : 0.f;
                ev[i][q] = (lc && fl[i]) ? e[o] : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < R; i++) {
            const int64_t r = r0 + i * nwaves;
            if (r >= nrows) break;
            if (dn[i]) continue;
            const float base = top / rs_rt[i];
            float fsum = 0.f;
#pragma unroll
            for (int q = 0; q < CPL; q++) {
                const int c = lane + WAVE * q;
                if (c < LD) {
                    const size_t o = (size_t)r * LD + c;
                    float f = 0.f, s_new = 0.f, rt_new = 1.f;
                    if (c < k) {
                        float s = sv[i][q];
                        if (fl[i]) {
                            const float fresh = fmaf(ev[i][q], av[i][q], prior);
                            s = blend_shape(w_new, fresh, w_old, s);
                            shp[o] = s;
                        }
                        float rt = rv[i][q];
                        if (rate_mode == 0) {
                            rt = base + csl[q];
                            if (rte) rte[o] = rt;
                        } else if (fl[i]) {
                            rt = blend_rate(step, base, csl[q], step_prev, rt);
                            rte[o] = rt;
                        }
                        f = s / rt;
                        s_new = s;
                        rt_new = rt;
                    }
                    if (fac) fac[o] = f;
                    fsum += f;
                    csacc[q] += f;
                    if (e_out && fl[i] && c < k) {     // (kept for the E row below)
                        sv[i][q] = s_new;
                        rv[i][q] = rt_new;
                    }
                }
            }
            if (e_out && fl[i]) {                      // (wave-uniform)
                double en[CPL];
                int ehi = 0;
#pragma unroll
                for (int q = 0; q < CPL; q++) {
                    const int c = lane + WAVE * q;
                    en[q] = (c < k) ? expect_ratio(sv[i][q], rv[i][q]) : 0.0;
                    ehi = max(ehi, __double2hiint(en[q]));
                }
                const double inv = row_pow2_scale(ehi);
#pragma unroll
                for (int q = 0; q < CPL; q++) {
                    const int c = lane + WAVE * q;
                    if (c < LD) e_out[(size_t)r * LD + c] = (c < k) ? (float)(en[q] * inv) : 0.f;
                }
            }
            if (rs_prev_out && lane == 0) rs_prev_out[r] = rs_rt[i];
            if (rs_mode == 2 || (rs_mode == 1 && fl[i])) {
                fsum = wave_sum(fsum);
                if (lane == 0) rs[r] = blend_rate(step, add, fsum, step_prev, rs_old[i]);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < CPL; q++) {
        const int c = lane + WAVE * q;
        if (c < LD) red[wid][c] = csacc[q];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < LD; c += BLOCK) {
        float t = red[0][c];
#pragma unroll
        for (int w2 = 1; w2 < WPB; w2++) t += red[w2][c];
        cs_partial[(size_t)blockIdx.x * LD + c] = t;
    }
}

// listed rows: mode 0: rte[r] = step*(top/rs[r] + cs_other) + (1-step)*rte[r]       (PXI:320 / PXI:372)
//              mode 1: rs[r]  = step*(add + sum_k fac[r]) + (1-step)*rs[r]          (PXI:324-325, 376-377)
template <int LD>
__global__ __launch_bounds__(BLOCK) void svi_rate_rows_kernel(const int64_t *__restrict__ row_list, int64_t nrows,
                                                              float *__restrict__ rte, const float *__restrict__ fac,
                                                              float *__restrict__ rs,
                                                              const float *__restrict__ cs_other, float top,
                                                              float add, float step, float step_prev, int mode,
                                                              int k) {
    constexpr int CPL = (LD + WAVE - 1) / WAVE;
    const int lane = threadIdx.x & (WAVE - 1);
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * WPB;
    for (int64_t t = (int64_t)blockIdx.x * WPB + wid; t < nrows; t += nwaves) {
        const int64_t r = row_list[t];
        if (mode == 0) {
            const float base = top / rs[r];
#pragma unroll
            for (int q = 0; q < CPL; q++) {
                const int c = lane + WAVE * q;
                if (c < k) {
                    const size_t o = (size_t)r * LD + c;
                    rte[o] = blend_rate(step, base, cs_other[c], step_prev, rte[o]);
                }
            }
        } else {
            float fsum = 0.f;
#pragma unroll
            for (int q = 0; q < CPL; q++) {
                const int c = lane + WAVE * q;
                if (c < k) fsum += fac[(size_t)r * LD + c];
            }
            fsum = wave_sum(fsum);
            if (lane == 0) rs[r] = blend_rate(step, add, fsum, step_prev, rs[r]);
        }
    }
}

// ----------------------------------------------------------------------------------------
// listed-pair kernels: LPR lanes per (user,item) pair
// ----------------------------------------------------------------------------------------
template <int LPR, int VPL>
__device__ __forceinline__ float pair_dot(const float *__restrict__ T, const float *__restrict__ B, int u, int i,
                                          int j) {
    constexpr int LD = 4 * LPR * VPL;
    const float4 *tp = reinterpret_cast<const float4 *>(T + (size_t)u * LD);
    const float4 *bp = reinterpret_cast<const float4 *>(B + (size_t)i * LD);
    float p = dot4(tp[0 * LPR + j], bp[0 * LPR + j]);   // (not 0.f + ...: an add the compiler must keep for -0)
#pragma unroll
    for (int v = 1; v < VPL; v++) p += dot4(tp[v * LPR + j], bp[v * LPR + j]);
    return group_sum<LPR>(p);
}

template <int LPR, int VPL, bool FULL>
__global__ __launch_bounds__(BLOCK) void pair_llk_kernel(const float *__restrict__ T, const float *__restrict__ B,
                                                         const int32_t *__restrict__ ix_u,
                                                         const int32_t *__restrict__ ix_i,
                                                         const float *__restrict__ y, int64_t n,
                                                         double *__restrict__ partial) {
    constexpr int NG = WAVE / LPR;
    __shared__ double red[WPB][3];
    const int lane = threadIdx.x & (WAVE - 1);
    const int g = lane / LPR, j = lane % LPR;
    const int wid = threadIdx.x >> 6;
    const int64_t ngroups = (int64_t)gridDim.x * WPB * NG;
    const int64_t gid = ((int64_t)blockIdx.x * WPB + wid) * NG + g;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    // wave-uniform trip count (cross-lane reductions need whole groups; tails are masked by y = 0)
    const int64_t iters = (n + ngroups - 1) / ngroups;
    for (int64_t it = 0; it < iters; it++) {
        const int64_t p = it * ngroups + gid;
        const bool live = p < n;
        const int u = live ? ix_u[p] : 0;
        const int i = live ? ix_i[p] : 0;
        const float yy = live ? y[p] : 0.f;
        const float yhat = pair_dot<LPR, VPL>(T, B, u, i, j);
        if (live && j == 0) {
            if constexpr (FULL)
                a0 += (double)yy * log((double)yhat) - lgamma((double)yy + 1.0);
            else
                a0 += (double)(yy * logf(yhat));
            const float d = yy - yhat;
            a1 += (double)(d * d);
            a2 += (double)yhat;
        }
    }
    a0 = wave_sum_d(a0);
    a1 = wave_sum_d(a1);
    a2 = wave_sum_d(a2);
    if (lane == 0) {
        red[wid][0] = a0;
        red[wid][1] = a1;
        red[wid][2] = a2;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        double t = red[0][threadIdx.x];
#pragma unroll
        for (int w2 = 1; w2 < WPB; w2++) t += red[w2][threadIdx.x];
        partial[(size_t)blockIdx.x * 4 + threadIdx.x] = t;
    }
    if (threadIdx.x == 3) partial[(size_t)blockIdx.x * 4 + 3] = 0.0;
}

// train llk over the row-grouped layout: the user row is read once per segment, only the item rows
// are gathered (the listed-pair kernel gathers both).  Same terms as pair_llk_kernel.
template <int LPR, int VPL, bool FULL>
__global__ __launch_bounds__(BLOCK) void llk_sweep_kernel(const hpf_segment *__restrict__ segs, int64_t nseg,
                                                          const int32_t *__restrict__ idx,
                                                          const float *__restrict__ y,
                                                          const float *__restrict__ tab_self,
                                                          const float *__restrict__ tab_other,
                                                          double *__restrict__ partial) {
    constexpr int LD = 4 * LPR * VPL;
    constexpr int NG = WAVE / LPR;
    constexpr int U = HPF_U;
    __shared__ double red[WPB][3];
    const int lane = threadIdx.x & (WAVE - 1);
    const int g = lane / LPR;
    const int j = lane % LPR;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * WPB;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (int64_t sg = (int64_t)blockIdx.x * WPB + wid; sg < nseg; sg += nwaves) {
        const hpf_segment sgm = segs[sg];
        const int len = sgm.len & HPF_SEG_LEN_MASK;
        const float4 *selfp = reinterpret_cast<const float4 *>(tab_self + (size_t)sgm.row * LD);
        float4 rv[VPL];
#pragma unroll
        for (int v = 0; v < VPL; v++) rv[v] = selfp[v * LPR + j];
        const int32_t *ip = idx + sgm.begin;
        const float *yp = y + sgm.begin;
        for (int base = 0; base < len; base += WAVE) {
            const int n = min(WAVE, len - base);
            int myc = 0;
            float myy = 0.f;
            if (lane < n) {
                myc = ip[base + lane];
                myy = yp[base + lane];
            }
            int nsteps = (n + NG - 1) / NG;
            nsteps = (nsteps + U - 1) & ~(U - 1);
            for (int t0 = 0; t0 < nsteps; t0 += U) {
                float4 o[U][VPL];
                float yy[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int src = (t0 + u) * NG + g;
                    const int c = __shfl(myc, src);
                    yy[u] = __shfl(myy, src);
                    const float4 *op = reinterpret_cast<const float4 *>(tab_other + (size_t)c * LD);
#pragma unroll
                    for (int v = 0; v < VPL; v++) o[u][v] = op[v * LPR + j];
                }
#pragma unroll
                for (int u = 0; u < U; u++) {
                    float p = dot4(rv[0], o[u][0]);   // (not 0.f + ...: an add the compiler must keep for -0)
#pragma unroll
                    for (int v = 1; v < VPL; v++) p += dot4(rv[v], o[u][v]);
                    const float yhat = group_sum<LPR>(p);
                    if (j == 0 && (t0 + u) * NG + g < n) {  // slots past the chunk's n entries are padding
                        if constexpr (FULL)
                            a0 += (double)yy[u] * log((double)yhat) - lgamma((double)yy[u] + 1.0);
                        else
                            a0 += (double)(yy[u] * logf(yhat));
                        const float d = yy[u] - yhat;
                        a1 += (double)(d * d);
                        a2 += (double)yhat;
                    }
                }
            }
        }
    }
    a0 = wave_sum_d(a0);
    a1 = wave_sum_d(a1);
    a2 = wave_sum_d(a2);
    if (lane == 0) {
        red[wid][0] = a0;
        red[wid][1] = a1;
        red[wid][2] = a2;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        double t = red[0][threadIdx.x];
#pragma unroll
        for (int w2 = 1; w2 < WPB; w2++) t += red[w2][threadIdx.x];
        partial[(size_t)blockIdx.x * 4 + threadIdx.x] = t;
    }
    if (threadIdx.x == 3) partial[(size_t)blockIdx.x * 4 + 3] = 0.0;
}

template <int LPR, int VPL>
__global__ __launch_bounds__(BLOCK) void pair_dot_kernel(const float *__restrict__ T, const float *__restrict__ B,
                                                         const int32_t *__restrict__ ix_u,
                                                         const int32_t *__restrict__ ix_i, int64_t n,
                                                         float *__restrict__ out) {
    constexpr int NG = WAVE / LPR;
    const int lane = threadIdx.x & (WAVE - 1);
    const int g = lane / LPR, j = lane % LPR;
    const int wid = threadIdx.x >> 6;
    const int64_t ngroups = (int64_t)gridDim.x * WPB * NG;
    const int64_t gid = ((int64_t)blockIdx.x * WPB + wid) * NG + g;
    const int64_t iters = (n + ngroups - 1) / ngroups;
    for (int64_t it = 0; it < iters; it++) {
        const int64_t p = it * ngroups + gid;
        const bool live = p < n;
        const int u = live ? ix_u[p] : 0;
        const int i = live ? ix_i[p] : 0;
        const float yhat = pair_dot<LPR, VPL>(T, B, u, i, j);
        if (live && j == 0) out[p] = yhat;
    }
}

// out[r] = <vec, tab[r]> for every row r: the scoring GEMV of HPF.topN (INIT:1337: Theta[user].dot(Beta.T))
template <int LPR, int VPL>
__global__ __launch_bounds__(BLOCK) void score_rows_kernel(const float *__restrict__ vec,
                                                           const float *__restrict__ tab, int64_t nrows,
                                                           float *__restrict__ out) {
    constexpr int LD = 4 * LPR * VPL;
    constexpr int NG = WAVE / LPR;
    const int lane = threadIdx.x & (WAVE - 1);
    const int g = lane / LPR, j = lane % LPR;
    const int wid = threadIdx.x >> 6;
    float4 v[VPL];
#pragma unroll
    for (int q = 0; q < VPL; q++) v[q] = reinterpret_cast<const float4 *>(vec)[q * LPR + j];
    const int64_t ngroups = (int64_t)gridDim.x * WPB * NG;
    const int64_t gid = ((int64_t)blockIdx.x * WPB + wid) * NG + g;
    const int64_t iters = (nrows + ngroups - 1) / ngroups;
    for (int64_t it = 0; it < iters; it++) {
        const int64_t r = it * ngroups + gid;
        const bool live = r < nrows;
        const float4 *tp = reinterpret_cast<const float4 *>(tab + (size_t)(live ? r : 0) * LD);
        float p = dot4(v[0], tp[0 * LPR + j]);   // (not 0.f + ...: an add the compiler must keep for -0)
#pragma unroll
        for (int q = 1; q < VPL; q++) p += dot4(v[q], tp[q * LPR + j]);
        p = group_sum<LPR>(p);
        if (live && j == 0) out[r] = p;
    }
}

// Measurement aid (tools/gather_probe.py): the sweep's memory access pattern with the arithmetic stripped --
// every 16-lane group gathers one 256-byte row per step, 8 steps in flight, rows taken from idx[].  Its GB/s is
// the ceiling the sweep kernel can be held against for a given table size / index distribution.
__global__ __launch_bounds__(BLOCK) void gather_probe_kernel(const int32_t *__restrict__ idx, int64_t n,
                                                             const float *__restrict__ tab,
                                                             float *__restrict__ sink) {
    constexpr int LPR = 16, LD = 64, NG = WAVE / LPR, U = 8;
    const int lane = threadIdx.x & (WAVE - 1);
    const int g = lane / LPR, j = lane % LPR;
    const int64_t nwaves = (int64_t)gridDim.x * WPB;
    const int64_t w = (int64_t)blockIdx.x * WPB + (threadIdx.x >> 6);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // each wave owns contiguous chunks of 64 indices, like a segment chunk in sweep_kernel
    for (int64_t base = w * WAVE; base + WAVE <= n; base += nwaves * WAVE) {
        const int myc = idx[base + lane];
        for (int t0 = 0; t0 < LPR; t0 += U) {
            float4 o[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int c = __shfl(myc, (t0 + u) * NG + g);
                o[u] = reinterpret_cast<const float4 *>(tab + (size_t)c * LD)[j];
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                acc.x += o[u].x;
                acc.y += o[u].y;
                acc.z += o[u].z;
                acc.w += o[u].w;
            }
        }
    }
    const float r = acc.x + acc.y + acc.z + acc.w;
    if (r == 123.456f) sink[0] = r;  // keeps the loads alive without a store stream
}

// ---------------------------------------------------------------------------------------------------------------
// initialize_parameters (PXI:127-138): the reference fills its four [n,k] tables from ONE MT19937 stream
// (numpy Generator.random(dtype=float32): one 32-bit output per value, (y >> 8) * 2^-24), as prior + 0.01*U.
// The stream's state words come from hpf_mt19937.hip (the recurrence, walked by many workgroups at once after a
// polynomial jump-ahead); everything else -- tempering, the float conversion, the affine map, the padded table layout,
// the ratio -- is the chip-wide elementwise pass below over the stored words.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void uniform_rows_kernel(const uint32_t *__restrict__ raw, float *__restrict__ out,
                                                             const float *__restrict__ den, float *__restrict__ ratio,
                                                             long long nrows, int k, int ld, float base, float scale) {
#pragma clang fp contract(off)
    const long long total = nrows * k;
    for (long long j = (long long)blockIdx.x * BLOCK + threadIdx.x; j < total; j += (long long)gridDim.x * BLOCK) {
        uint32_t y = raw[j];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        const float u = (float)(y >> 8) * (1.0f / 16777216.0f);
        const float sc = scale * u;                  // two roundings, as numpy's `prior + 0.01 * draw`
        const float v = base + sc;
        const long long at = (j / k) * ld + (j % k);
        out[at] = v;
        if (ratio) ratio[at] = __fdiv_rn(v, den[at]);   // Theta = Gamma_shp / Gamma_rte (PXI:140-141)
    }
}

inline int clamp_grid(int64_t want, int grid_blocks) {
    int64_t g = grid_blocks > 0 ? grid_blocks : 2048;
    if (want < g) g = want;
    if (g < 1) g = 1;
    return (int)g;
}

inline int last_error() { return (int)hipGetLastError(); }

}  // namespace

// ----------------------------------------------------------------------------------------
// C ABI
// ----------------------------------------------------------------------------------------
#define HPF_DISPATCH_LD(ld, CALL)                    \
    switch (ld) {                                    \
        case 32: { CALL(8, 1); break; }              \
        case 64: { CALL(16, 1); break; }             \
        case 128: { CALL(32, 1); break; }            \
        case 256: { CALL(64, 1); break; }            \
        case 512: { CALL(64, 2); break; }            \
        case 1024: { CALL(64, 4); break; }           \
        default: return HPF_EUNSUPPORTED;            \
    }

#define HPF_DISPATCH_LD1(ld, CALL)                   \
    switch (ld) {                                    \
        case 32: { CALL(32); break; }                \
        case 64: { CALL(64); break; }                \
        case 128: { CALL(128); break; }              \
        case 256: { CALL(256); break; }              \
        case 512: { CALL(512); break; }              \
        case 1024: { CALL(1024); break; }            \
        default: return HPF_EUNSUPPORTED;            \
    }

extern "C" {

int hpf_hip_abi_version(void) { return HPF_HIP_ABI_VERSION; }

int hpf_hip_ld_for_k(int k) {
    if (k <= 0) return HPF_EINVAL;
    int ld = 32;
    while (ld < k) ld <<= 1;
    if (ld > 1024) return HPF_EUNSUPPORTED;
    return ld;
}

int hpf_hip_device_info(int *cu_count, char *arch, int arch_len) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return (int)e;
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (arch && arch_len > 0) {
        strncpy(arch, prop.gcnArchName, (size_t)arch_len - 1);
        arch[arch_len - 1] = 0;
    }
    return 0;
}

// float4s of a table row up to the end of the 64-byte sector that holds column k - 1
static inline int sector_float4s(int k) { return ((k + 15) / 16) * 4; }

static int sweep_impl(const hpf_segment *segs, int64_t nseg, const int32_t *idx, const float *y, const float *tab_self,
                      const float *tab_other, float *part, float *acc_rows, int acc_ld, int k, int ld, int short_rows,
                      int grid_blocks, const int64_t *nseg_dev, hpf_direct::Signal sig, hipStream_t st) {
    if (nseg == 0 && !sig.peers_dev) return 0;
    if (nseg < 0 || k <= 0 || ld != hpf_hip_ld_for_k(k)) return HPF_EINVAL;
    if (nseg > 0 && (!segs || !idx || !y || !tab_self || !tab_other || !part)) return HPF_EINVAL;
    if (acc_rows && (acc_ld < k || acc_ld > ld)) return HPF_EINVAL;
    const int grid = clamp_grid((nseg + WPB - 1) / WPB, grid_blocks);      // (>= 1: a launch that only signals)
    FinalizeArgs fa = {};
    fa.acc_rows = acc_rows;
    fa.acc_ld = acc_ld;
    fa.nseg_dev = nseg_dev;
    fa.sig_peers = sig.peers_dev;
    fa.sig_kind = sig.kind;
    fa.sig_epoch = sig.epoch;
    fa.nq4 = sector_float4s(k);
    const bool skip = fa.nq4 < ld / 4;
    // short rows (a batch or a shard of a many-rank run: ~16 nonzeros per row): half the gathers in flight per wave
    // fill just as well and the smaller register file buys occupancy (-15 % at N=8)
    constexpr int US = (HPF_U >= 8) ? HPF_U / 2 : HPF_U;
#define LAUNCH(LPR, VPL, UU_, SKIP_)                                                                                \
    hipLaunchKernelGGL((sweep_kernel<LPR, VPL, 0, UU_, SKIP_>), dim3(grid), dim3(BLOCK), 0, st, segs, nseg, idx, y, \
                       tab_self, tab_other, part, fa)
#define CALL(LPR, VPL)                                                                                              \
    if (short_rows) {                                                                                               \
        if (skip) LAUNCH(LPR, VPL, US, true); else LAUNCH(LPR, VPL, US, false);                                     \
    } else {                                                                                                        \
        if (skip) LAUNCH(LPR, VPL, HPF_U, true); else LAUNCH(LPR, VPL, HPF_U, false);                               \
    }
    HPF_DISPATCH_LD(ld, CALL)
#undef CALL
#undef LAUNCH
    return last_error();
}

int hpf_hip_sweep_f32(const hpf_segment *segs, int64_t nseg, const int32_t *idx, const float *y,
                      const float *tab_self, const float *tab_other, float *part, float *acc_rows, int acc_ld, int k,
                      int ld, int short_rows, int grid_blocks, const int64_t *nseg_dev, void *stream) {
    if (nseg > 0 && (!segs || !idx || !y || !tab_self || !tab_other || !part)) return HPF_EINVAL;
    return sweep_impl(segs, nseg, idx, y, tab_self, tab_other, part, acc_rows, acc_ld, k, ld, short_rows, grid_blocks,
                      nseg_dev, hpf_direct::Signal{nullptr, 0, 0}, (hipStream_t)stream);
}

static int sweep_finalize_impl(const hpf_segment *segs, int64_t nseg, const int32_t *idx, const float *y,
                               const float *tab_self, const float *tab_other, float *part, float *e_new, float *shp,
                               float *rte, float *fac, float *rs, float *rs_prev, const float *cs_other,
                               float *cs_partial, float prior_shp, float top_shp, float add_rte, int k, int ld,
                               int grid_blocks, float *cs_other_copy, hpf_direct::Signal sig, hipStream_t st) {
    if (!segs || !idx || !y || !tab_self || !tab_other || !part || !e_new || !rs || !cs_other || !cs_partial ||
        nseg <= 0 || k <= 0 || ld != hpf_hip_ld_for_k(k) || grid_blocks <= 0)
        return HPF_EINVAL;
    // grid NOT clamped: every block writes its cs_partial row
    FinalizeArgs fa = {cs_other, cs_partial, e_new, shp, rte, fac, rs, prior_shp, top_shp, add_rte, k, rs_prev,
                       nullptr, 0, nullptr};
    fa.sig_peers = sig.peers_dev;
    fa.sig_kind = sig.kind;
    fa.sig_epoch = sig.epoch;
    fa.cs_other_copy = cs_other_copy;
    fa.nq4 = sector_float4s(k);
    const bool skip = fa.nq4 < ld / 4;
#define CALL(LPR, VPL)                                                                                            \
    if (skip)                                                                                                     \
        hipLaunchKernelGGL((sweep_kernel<LPR, VPL, 1, HPF_U, true>), dim3(grid_blocks), dim3(BLOCK), 0, st, segs, \
                           nseg, idx, y, tab_self, tab_other, part, fa);                                          \
    else                                                                                                          \
        hipLaunchKernelGGL((sweep_kernel<LPR, VPL, 1>), dim3(grid_blocks), dim3(BLOCK), 0, st, segs, nseg, idx,   \
                           y, tab_self, tab_other, part, fa);
    HPF_DISPATCH_LD(ld, CALL)
#undef CALL
    return last_error();
}

int hpf_hip_sweep_finalize_f32(const hpf_segment *segs, int64_t nseg, const int32_t *idx, const float *y,
                               const float *tab_self, const float *tab_other, float *part, float *e_new, float *shp,
                               float *rte, float *fac, float *rs, float *rs_prev, const float *cs_other,
                               float *cs_partial, float prior_shp, float top_shp, float add_rte, int k, int ld,
                               int grid_blocks, void *stream) {
    return sweep_finalize_impl(segs, nseg, idx, y, tab_self, tab_other, part, e_new, shp, rte, fac, rs, rs_prev, cs_other,
                               cs_partial, prior_shp, top_shp, add_rte, k, ld, grid_blocks, nullptr,
                               hpf_direct::Signal{nullptr, 0, 0}, (hipStream_t)stream);
}

int hpf_hip_sweep_svi_f32(const hpf_segment *segs, int64_t nseg, const int32_t *idx, const float *y, const float *tab_self,
                          const float *tab_other, float *part, float *e_new, float *shp, float *rte, float *fac, float *rs,
                          const float *cs_other, float *cs_partial, float prior, float w_new, float w_old, float top,
                          float add, float step, float step_prev, int k, int ld, int short_rows, int grid_blocks,
                          const int64_t *nseg_dev, void *stream) {
    if (!segs || !idx || !y || !tab_self || !tab_other || !part || !shp || !rte || !rs || !cs_other || !cs_partial ||
        nseg <= 0 || k <= 0 || ld != hpf_hip_ld_for_k(k) || grid_blocks <= 0 || (e_new && e_new != tab_self))
        return HPF_EINVAL;
    // grid NOT clamped: every block writes its cs_partial row (zeros when it had no segment)
    FinalizeArgs fa = {};
    fa.cs_other = cs_other;
    fa.cs_partial = cs_partial;
    fa.e_new = e_new;
    fa.shp = shp;
    fa.rte = rte;
    fa.fac = fac;
    fa.rs = rs;
    fa.prior_shp = prior;
    fa.top_shp = top;
    fa.add_rte = add;
    fa.k = k;
    fa.nseg_dev = nseg_dev;
    fa.nq4 = sector_float4s(k);
    fa.w_new = w_new;
    fa.w_old = w_old;
    fa.step = step;
    fa.step_prev = step_prev;
    const bool skip = fa.nq4 < ld / 4;
    hipStream_t st = (hipStream_t)stream;
    constexpr int US = (HPF_U >= 8) ? HPF_U / 2 : HPF_U;
#define LAUNCH(LPR, VPL, UU_, SKIP_)                                                                                      \
    hipLaunchKernelGGL((sweep_kernel<LPR, VPL, 2, UU_, SKIP_>), dim3(grid_blocks), dim3(BLOCK), 0, st, segs, nseg, idx, y, \
                       tab_self, tab_other, part, fa)
#define CALL(LPR, VPL)                                                                                              \
    if (short_rows) {                                                                                               \
        if (skip) LAUNCH(LPR, VPL, US, true); else LAUNCH(LPR, VPL, US, false);                                     \
    } else {                                                                                                        \
        if (skip) LAUNCH(LPR, VPL, HPF_U, true); else LAUNCH(LPR, VPL, HPF_U, false);                               \
    }
    HPF_DISPATCH_LD(ld, CALL)
#undef CALL
#undef LAUNCH
    return last_error();
}

int hpf_hip_sweep_svi_batch_f32(const hpf_segment *segs, int64_t nseg, const int32_t *idx, const float *y, float *e_self,
                                const float *tab_other, float *part, float *shp, const float *rte_in, float *rte_out,
                                float *fac, float *rs, float *rs_prev_out, const float *rate_rs, const float *rate_cs,
                                float rate_top, const float *cs_other, float *cs_partial, float prior, float w_new,
                                float w_old, float top, float add, float step, float step_prev, int k, int ld,
                                int short_rows, int grid_blocks, const int64_t *nseg_dev, void *stream) {
    if (!segs || !idx || !y || !e_self || !tab_other || !part || !shp || !rs || !cs_other || !cs_partial ||
        (!rate_rs && !rte_in) || (rate_rs && !rate_cs) || nseg <= 0 || k <= 0 || ld != hpf_hip_ld_for_k(k) ||
        grid_blocks <= 0)
        return HPF_EINVAL;
    // grid NOT clamped: every block writes its cs_partial row (zeros when it had no segment)
    FinalizeArgs fa = {};
    fa.cs_other = cs_other;
    fa.cs_partial = cs_partial;
    fa.e_new = e_self;
    fa.shp = shp;
    fa.rte = rte_out;
    fa.fac = fac;
    fa.rs = rs;
    fa.rs_prev = rs_prev_out;
    fa.prior_shp = prior;
    fa.top_shp = top;
    fa.add_rte = add;
    fa.k = k;
    fa.nseg_dev = nseg_dev;
    fa.nq4 = sector_float4s(k);
    fa.w_new = w_new;
    fa.w_old = w_old;
    fa.step = step;
    fa.step_prev = step_prev;
    fa.rate_rs = rate_rs;
    fa.rate_cs = rate_cs;
    fa.rte_in = rte_in;
    fa.rate_top = rate_top;
    const bool skip = fa.nq4 < ld / 4;
    hipStream_t st = (hipStream_t)stream;
    constexpr int US = (HPF_U >= 8) ? HPF_U / 2 : HPF_U;
#define LAUNCH(LPR, VPL, UU_, SKIP_)                                                                                      \
    hipLaunchKernelGGL((sweep_kernel<LPR, VPL, 3, UU_, SKIP_>), dim3(grid_blocks), dim3(BLOCK), 0, st, segs, nseg, idx, y, \
                       (const float *)e_self, tab_other, part, fa)
#define CALL(LPR, VPL)                                                                                              \
    if (short_rows) {                                                                                               \
        if (skip) LAUNCH(LPR, VPL, US, true); else LAUNCH(LPR, VPL, US, false);                                     \
    } else {                                                                                                        \
        if (skip) LAUNCH(LPR, VPL, HPF_U, true); else LAUNCH(LPR, VPL, HPF_U, false);                               \
    }
    HPF_DISPATCH_LD(ld, CALL)
#undef CALL
#undef LAUNCH
    return last_error();
}

int hpf_hip_row_finalize_f32(const float *part, const int64_t *row_seg_ptr, const int64_t *row_list, int64_t nrows,
                             const float *e_old, float *e_new, float *shp, float *rte, float *fac, float *rs,
                             float *rs_prev, const float *cs_other, float *cs_partial, float prior_shp, float top_shp,
                             float add_rte, int k, int ld, int part_ld, int grid_blocks, void *stream) {
    if (!part || !e_old || !e_new || !rs || !cs_other || !cs_partial || nrows < 0 || k <= 0 ||
        ld != hpf_hip_ld_for_k(k) || grid_blocks <= 0 || part_ld < k || part_ld > ld)
        return HPF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    RowRanges rr = {};   // n = 0: identity
    // the grid is NOT clamped: cs_partial has exactly grid_blocks rows and all are written
#define CALL(LD)                                                                                                  \
    if (!row_seg_ptr && !row_list)                                                                                     \
        hipLaunchKernelGGL((row_finalize_kernel<LD, true>), dim3(grid_blocks), dim3(BLOCK), 0, st, part, row_seg_ptr,      \
                           row_list, nrows, e_old, e_new, shp, rte, fac, rs, rs_prev, cs_other, cs_partial, prior_shp,     \
                           top_shp, add_rte, k, part_ld, rr, LD);                                                          \
    else                                                                                                               \
        hipLaunchKernelGGL((row_finalize_kernel<LD, false>), dim3(grid_blocks), dim3(BLOCK), 0, st, part, row_seg_ptr,     \
                           row_list, nrows, e_old, e_new, shp, rte, fac, rs, rs_prev, cs_other, cs_partial, prior_shp,     \
                           top_shp, add_rte, k, part_ld, rr, LD);
    HPF_DISPATCH_LD1(ld, CALL)
#undef CALL
    return last_error();
}

int hpf_hip_row_finalize_ranges_f32(const float *acc, int nranges, const int64_t *range_rows,
                                    const int64_t *range_acc_begin, const int64_t *range_row_begin,
                                    const float *e_old, float *e_new, float *shp, float *rte, float *fac, float *rs,
                                    float *rs_prev, const float *cs_other, float *cs_partial, float prior_shp,
                                    float top_shp, float add_rte, int k, int ld, int acc_ld, int e_new_ld,
                                    int grid_blocks, void *stream) {
    if (!acc || !range_rows || !range_acc_begin || !range_row_begin || !e_old || !e_new || !rs || !cs_other ||
        !cs_partial || nranges <= 0 || nranges > HPF_MAX_ROW_RANGES || k <= 0 || ld != hpf_hip_ld_for_k(k) ||
        grid_blocks <= 0 || acc_ld < k || acc_ld > ld || e_new_ld < k || e_new_ld > ld)
        return HPF_EINVAL;
    RowRanges rr = {};
    rr.n = nranges;
    int64_t nrows = 0;
    for (int i = 0; i < nranges; i++) {
        if (range_rows[i] < 0 || range_acc_begin[i] < 0 || range_row_begin[i] < 0) return HPF_EINVAL;
        rr.v_begin[i] = nrows;
        rr.t_begin[i] = range_acc_begin[i];
        rr.row_begin[i] = range_row_begin[i];
        nrows += range_rows[i];
    }
    hipStream_t st = (hipStream_t)stream;
#define CALL(LD)                                                                                                       \
    hipLaunchKernelGGL((row_finalize_kernel<LD, true>), dim3(grid_blocks), dim3(BLOCK), 0, st, acc,                    \
                       (const int64_t *)nullptr, (const int64_t *)nullptr, nrows, e_old, e_new, shp, rte, fac, rs,     \
                       rs_prev, cs_other, cs_partial, prior_shp, top_shp, add_rte, k, acc_ld, rr, e_new_ld);
    HPF_DISPATCH_LD1(ld, CALL)
#undef CALL
    return last_error();
}

int hpf_hip_gather_payload_ld(int k) { return k > 0 ? ((k + 1 + 3) / 4) * 4 : HPF_EINVAL; }

static int item_shape_impl(const float *acc, const RowRanges &rr, int64_t nrows, const float *e_old, float *shp_out,
                           float *send, const float *rs, float *rs_prev, float prior_shp, float top_shp, int k, int ld,
                           int grid_blocks, hipStream_t st) {
    const int sld = hpf_hip_gather_payload_ld(k);
#define CALL(LD)                                                                                                      \
    hipLaunchKernelGGL((item_shape_kernel<LD>), dim3(grid_blocks), dim3(BLOCK), 0, st, acc, e_old, shp_out, send, sld, \
                       rs, rs_prev, prior_shp, top_shp, k, rr, nrows);
    HPF_DISPATCH_LD1(ld, CALL)
#undef CALL
    return last_error();
}

int hpf_hip_item_shape_rows_f32(const float *acc, int nranges, const int64_t *range_rows, const int64_t *range_acc_begin,
                                const int64_t *range_row_begin, const float *e_old, float *shp_out, float *send,
                                const float *rs, float *rs_prev, float prior_shp, float top_shp, int k, int ld,
                                int grid_blocks, void *stream) {
    if (!acc || !range_rows || !range_acc_begin || !range_row_begin || !e_old || !shp_out || !send || !rs || nranges <= 0 ||
        nranges > HPF_MAX_ROW_RANGES || k <= 0 || ld != hpf_hip_ld_for_k(k) || grid_blocks <= 0)
        return HPF_EINVAL;
    RowRanges rr = {};
    rr.n = nranges;
    int64_t nrows = 0;
    for (int i = 0; i < nranges; i++) {
        if (range_rows[i] < 0 || range_acc_begin[i] < 0 || range_row_begin[i] < 0) return HPF_EINVAL;
        rr.v_begin[i] = nrows;
        rr.t_begin[i] = range_acc_begin[i];
        rr.row_begin[i] = range_row_begin[i];
        nrows += range_rows[i];
    }
    if (nrows == 0) return 0;
    return item_shape_impl(acc, rr, nrows, e_old, shp_out, send, rs, rs_prev, prior_shp, top_shp, k, ld, grid_blocks,
                           (hipStream_t)stream);
}

static int item_apply_impl(const float *recv, const OwnerBlocks &own, const float *shp_own, float *e_tab, float *shp,
                           float *fac, float *rs, const float *cs_other, float *cs_partial, float add_rte, int k, int ld,
                           int rank, int world, int64_t nrows, int nranges, const int64_t *range_lo,
                           const int64_t *range_hi, int grid_blocks, hipStream_t st) {
    if ((!recv && own.n <= 0) || !shp_own || !e_tab || !rs || !cs_other || !cs_partial || k <= 0 ||
        ld != hpf_hip_ld_for_k(k) || rank < 0 || world <= 0 || rank >= world || nrows <= 0 || nranges <= 0 ||
        nranges > HPF_MAX_ROW_RANGES || !range_lo || !range_hi || grid_blocks <= 0)
        return HPF_EINVAL;
    if (grid_blocks % world != 0) return HPF_EINVAL;     // (gx blocks per rank's block of the gathered buffer)
    ApplyRanges ar = {};
    ar.n = nranges;
    int64_t t = 0;
    for (int i = 0; i < nranges; i++) {
        if (range_hi[i] <= range_lo[i] || (range_hi[i] - range_lo[i]) % world != 0) return HPF_EINVAL;
        ar.lo[i] = range_lo[i];
        ar.m[i] = (range_hi[i] - range_lo[i]) / world;
        ar.t0[i] = t;
        t += ar.m[i];
    }
    ar.total = t;
    const dim3 grid((unsigned)(grid_blocks / world), (unsigned)world);
    const int sld = hpf_hip_gather_payload_ld(k);
#define CALL(LPR, VPL)                                                                                                \
    hipLaunchKernelGGL((item_apply_kernel<LPR, VPL>), grid, dim3(BLOCK), 0, st, recv, sld, shp_own, e_tab, shp, fac,   \
                       rs, cs_other, cs_partial, add_rte, k, rank, nrows, ar, own);
    HPF_DISPATCH_LD(ld, CALL)
#undef CALL
    return last_error();
}

int hpf_hip_item_apply_rows_f32(const float *recv, const float *shp_own, float *e_tab, float *shp, float *fac, float *rs,
                                const float *cs_other, float *cs_partial, float add_rte, int k, int ld, int rank,
                                int world, int64_t nrows, int nranges, const int64_t *range_lo, const int64_t *range_hi,
                                int grid_blocks, void *stream) {
    if (!recv) return HPF_EINVAL;
    OwnerBlocks own = {};
    return item_apply_impl(recv, own, shp_own, e_tab, shp, fac, rs, cs_other, cs_partial, add_rte, k, ld, rank, world,
                           nrows, nranges, range_lo, range_hi, grid_blocks, (hipStream_t)stream);
}

int hpf_hip_colsum_reduce_f32(const float *cs_partial, int nblk, float *cs_out, int ld, void *stream) {
    if (!cs_partial || !cs_out || nblk <= 0 || ld < 32 || (ld & 3) || (reinterpret_cast<uintptr_t>(cs_partial) & 15))
        return HPF_EINVAL;
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3(ld / 4), dim3(1024), 0, (hipStream_t)stream,
                       cs_partial, nblk, cs_out, ld, (const hpf_p2p::Peers *)nullptr, 0, 0u, 0u, -1);
    return last_error();
}

int hpf_hip_colsum_f32(const float *tab, int64_t nrows, int ld, float *cs_partial, int grid_blocks, void *stream) {
    if (!tab || !cs_partial || nrows <= 0 || grid_blocks <= 0) return HPF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
#define CALL(LD) \
    hipLaunchKernelGGL((colsum_kernel<LD>), dim3(grid_blocks), dim3(BLOCK), 0, st, tab, nrows, cs_partial);
    HPF_DISPATCH_LD1(ld, CALL)
#undef CALL
    return last_error();
}

int hpf_hip_colsum_sequential_f32(const float *tab, int64_t nrows, int ld, float *cs_out, void *stream) {
    if (!tab || !cs_out || nrows < 0 || ld < 32 || (ld & 31) || (reinterpret_cast<uintptr_t>(tab) & 15)) return HPF_EINVAL;
    hipLaunchKernelGGL(colsum_sequential_kernel, dim3(ld / SEQ_COLS), dim3(SEQ_THREADS), 0, (hipStream_t)stream, tab, nrows,
                       ld, cs_out);
    return last_error();
}

int hpf_hip_expect_f32(const float *shp, const float *rte, float *e, const int64_t *row_list, const uint8_t *flag,
                       int64_t nrows, int k, int ld, const float *rate_rs, const float *rate_cs, float rate_top,
                       float *rte_out, void *stream) {
    if (nrows == 0) return 0;
    if (!shp || (!rte && !rate_rs) || (rate_rs && !rate_cs) || !e || nrows < 0 || k <= 0 || ld != hpf_hip_ld_for_k(k) ||
        (rte_out && !rate_rs))
        return HPF_EINVAL;
    const FactoredRate fr = {rate_rs, rate_cs, rate_top};
    hipStream_t st = (hipStream_t)stream;
    const int grid = (flag && !row_list) ? clamp_grid((nrows + WPB * WAVE - 1) / (WPB * WAVE), 2048)
                                         : clamp_grid((nrows + WPB - 1) / WPB, 2048);
#define CALL(LD) \
    hipLaunchKernelGGL((expect_kernel<LD>), dim3(grid), dim3(BLOCK), 0, st, shp, rte, e, row_list, flag, nrows, k, fr, rte_out);
    HPF_DISPATCH_LD1(ld, CALL)
#undef CALL
    return last_error();
}

int hpf_hip_segsum_f32(const float *part, const int64_t *row_seg_ptr, const int64_t *row_list, int64_t nrows,
                       float *acc, int ld, int acc_ld, int acc_by_row, void *stream) {
    if (nrows == 0) return 0;
    if (!part || !row_seg_ptr || !acc || nrows < 0 || ld < 32 || acc_ld <= 0 || acc_ld > ld) return HPF_EINVAL;
    const int grid = clamp_grid((nrows * acc_ld + BLOCK - 1) / BLOCK, 4096);
    hipLaunchKernelGGL(segsum_kernel, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, part, row_seg_ptr, row_list,
                       nrows, acc, ld, acc_ld, acc_by_row);
    return last_error();
}

int hpf_hip_pair_llk_f32(const float *T, const float *B, const int32_t *ix_u, const int32_t *ix_i, const float *y,
                         int64_t n, double *partial, int k, int ld, int full_llk, int grid_blocks, void *stream) {
    if (!T || !B || !partial || n < 0 || k <= 0 || ld != hpf_hip_ld_for_k(k) || grid_blocks <= 0) return HPF_EINVAL;
    if (n > 0 && (!ix_u || !ix_i || !y)) return HPF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    // grid not clamped: partial has 4*grid_blocks entries and all are written
#define CALL(LPR, VPL)                                                                                              \
    if (full_llk)                                                                                                   \
        hipLaunchKernelGGL((pair_llk_kernel<LPR, VPL, true>), dim3(grid_blocks), dim3(BLOCK), 0, st, T, B, ix_u,   \
                           ix_i, y, n, partial);                                                                    \
    else                                                                                                            \
        hipLaunchKernelGGL((pair_llk_kernel<LPR, VPL, false>), dim3(grid_blocks), dim3(BLOCK), 0, st, T, B, ix_u,  \
                           ix_i, y, n, partial);
    HPF_DISPATCH_LD(ld, CALL)
#undef CALL
    return last_error();
}

int hpf_hip_llk_sweep_f32(const hpf_segment *segs, int64_t nseg, const int32_t *idx, const float *y, const float *T,
                          const float *B, double *partial, int k, int ld, int full_llk, int grid_blocks,
                          void *stream) {
    if (!T || !B || !partial || nseg < 0 || k <= 0 || ld != hpf_hip_ld_for_k(k) || grid_blocks <= 0) return HPF_EINVAL;
    if (nseg > 0 && (!segs || !idx || !y)) return HPF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    // grid not clamped: partial has 4*grid_blocks entries and all are written
#define CALL(LPR, VPL)                                                                                              \
    if (full_llk)                                                                                                   \
        hipLaunchKernelGGL((llk_sweep_kernel<LPR, VPL, true>), dim3(grid_blocks), dim3(BLOCK), 0, st, segs, nseg,  \
                           idx, y, T, B, partial);                                                                  \
    else                                                                                                            \
        hipLaunchKernelGGL((llk_sweep_kernel<LPR, VPL, false>), dim3(grid_blocks), dim3(BLOCK), 0, st, segs, nseg, \
                           idx, y, T, B, partial);
    HPF_DISPATCH_LD(ld, CALL)
#undef CALL
    return last_error();
}

int hpf_hip_pair_dot_f32(const float *T, const float *B, const int32_t *ix_u, const int32_t *ix_i, int64_t n,
                         float *out, int k, int ld, void *stream) {
    if (n == 0) return 0;
    if (!T || !B || !ix_u || !ix_i || !out || n < 0 || k <= 0 || ld != hpf_hip_ld_for_k(k)) return HPF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
#define CALL(LPR, VPL)                                                                                          \
    {                                                                                                           \
        const int grid = clamp_grid((n + (WPB * (WAVE / LPR)) - 1) / (WPB * (WAVE / LPR)), 2048);              \
        hipLaunchKernelGGL((pair_dot_kernel<LPR, VPL>), dim3(grid), dim3(BLOCK), 0, st, T, B, ix_u, ix_i, n, out); \
    }
    HPF_DISPATCH_LD(ld, CALL)
#undef CALL
    return last_error();
}

int hpf_hip_svi_shape_rows_f32(const int64_t *row_list, int64_t nrows, const float *acc, const float *e, float *shp,
                               float prior, float w_new, float w_old, int k, int ld, int acc_by_row, void *stream) {
    if (nrows == 0) return 0;
    if (!row_list || !e || !shp || nrows < 0 || k <= 0 || ld != hpf_hip_ld_for_k(k)) return HPF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int grid = clamp_grid((nrows + WPB - 1) / WPB, 2048);
#define CALL(LD)                                                                                                   \
    hipLaunchKernelGGL((svi_shape_rows_kernel<LD>), dim3(grid), dim3(BLOCK), 0, st, row_list, nrows, acc, e, shp,  \
                       prior, w_new, w_old, k, acc_by_row);
    HPF_DISPATCH_LD1(ld, CALL)
#undef CALL
    return last_error();
}

int hpf_hip_svi_refresh_f32(int64_t nrows, const float *shp, float *rte, float *fac, float *rs, const float *cs_other,
                            float *cs_partial, float top, float add, float step, float step_prev, int refresh_rte,
                            int blend_rs, int k, int ld, int grid_blocks, void *stream) {
    if (!shp || !rte || !fac || !rs || !cs_partial || (refresh_rte && !cs_other) || nrows <= 0 || k <= 0 ||
        ld != hpf_hip_ld_for_k(k) || grid_blocks <= 0)
        return HPF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    // grid not clamped: every block writes its cs_partial row
#define CALL(LD)                                                                                                    \
    hipLaunchKernelGGL((svi_refresh_kernel<LD>), dim3(grid_blocks), dim3(BLOCK), 0, st, nrows, shp, rte, fac, rs,  \
                       cs_other, cs_partial, top, add, step, step_prev, refresh_rte, blend_rs, k);
    HPF_DISPATCH_LD1(ld, CALL)
#undef CALL
    return last_error();
}

int hpf_hip_svi_side_f32(int64_t nrows, const uint8_t *flag, const float *acc, const float *e, float *shp, float *rte,
                         float *fac, float *rs, const float *cs_other, float *cs_partial, float prior, float w_new,
                         float w_old, float top, float add, float step, float step_prev, int rate_mode, int rs_mode,
                         int k, int ld, int grid_blocks, const float *rs_rate, float *rs_prev_out, float *e_out,
                         int done_flag, void *stream) {
    if (!shp || !rs || !cs_other || !cs_partial || (flag && (!acc || !e)) || nrows <= 0 || k <= 0 ||
        ld != hpf_hip_ld_for_k(k) || grid_blocks <= 0 || (rate_mode != 0 && rate_mode != 1) || rs_mode < 0 ||
        rs_mode > 2 || (rate_mode == 1 && !rte) || (e_out && !flag) || done_flag < 0 || done_flag > 255 ||
        (done_flag != 0 && (!flag || rs_rate)))
        return HPF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    // grid not clamped: every block writes its cs_partial row
    if (ld >= 256 && rate_mode == 0 && !rte && !fac && !e_out) {
        // the batch side of a lazy epoch step: the streaming kernel of its own (same statements, same floats)
        switch (ld) {
#define CALLB(LD)                                                                                                      \
    case LD:                                                                                                           \
        hipLaunchKernelGGL((svi_lazy_batch_side_kernel<LD>), dim3(grid_blocks), dim3(BLOCK), 0, st, nrows, flag, acc, \
                           e, shp, rs, cs_other, cs_partial, prior, w_new, w_old, top, add, step, step_prev, rs_mode, \
                           k, rs_rate, rs_prev_out, done_flag);                                                        \
        break;
            CALLB(256)
            CALLB(512)
            CALLB(1024)
#undef CALLB
            default: return HPF_EUNSUPPORTED;
        }
        return last_error();
    }
#define CALL(LD)                                                                                                    \
    hipLaunchKernelGGL((svi_side_kernel<LD>), dim3(grid_blocks), dim3(BLOCK), 0, st, nrows, flag, acc, e, shp, rte, \
                       fac, rs, cs_other, cs_partial, prior, w_new, w_old, top, add, step, step_prev, rate_mode,    \
                       rs_mode, k, rs_rate, rs_prev_out, e_out, done_flag);
    HPF_DISPATCH_LD1(ld, CALL)
#undef CALL
    return last_error();
}

int hpf_hip_svi_rate_rows_f32(const int64_t *row_list, int64_t nrows, float *rte, const float *fac, float *rs,
                              const float *cs_other, float top, float add, float step, float step_prev, int mode,
                              int k, int ld, void *stream) {
    if (nrows == 0) return 0;
    if (!row_list || !rs || nrows < 0 || k <= 0 || ld != hpf_hip_ld_for_k(k) || (mode != 0 && mode != 1))
        return HPF_EINVAL;
    if ((mode == 0 && (!rte || !cs_other)) || (mode == 1 && !fac)) return HPF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int grid = clamp_grid((nrows + WPB - 1) / WPB, 2048);
#define CALL(LD)                                                                                                   \
    hipLaunchKernelGGL((svi_rate_rows_kernel<LD>), dim3(grid), dim3(BLOCK), 0, st, row_list, nrows, rte, fac, rs,  \
                       cs_other, top, add, step, step_prev, mode, k);
    HPF_DISPATCH_LD1(ld, CALL)
#undef CALL
    return last_error();
}

int hpf_hip_gather_probe_f32(const int32_t *idx, int64_t n, const float *tab, float *sink, int grid_blocks,
                             void *stream) {
    if (!idx || !tab || !sink || n <= 0 || grid_blocks <= 0) return HPF_EINVAL;
    hipLaunchKernelGGL(gather_probe_kernel, dim3(grid_blocks), dim3(BLOCK), 0, (hipStream_t)stream, idx, n, tab, sink);
    return last_error();
}

int hpf_hip_score_rows_f32(const float *vec, const float *tab, int64_t nrows, float *out, int k, int ld,
                           void *stream) {
    if (nrows == 0) return 0;
    if (!vec || !tab || !out || nrows < 0 || k <= 0 || ld != hpf_hip_ld_for_k(k)) return HPF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
#define CALL(LPR, VPL)                                                                                         \
    {                                                                                                          \
        const int grid = clamp_grid((nrows + (WPB * (WAVE / LPR)) - 1) / (WPB * (WAVE / LPR)), 2048);         \
        hipLaunchKernelGGL((score_rows_kernel<LPR, VPL>), dim3(grid), dim3(BLOCK), 0, st, vec, tab, nrows, out); \
    }
    HPF_DISPATCH_LD(ld, CALL)
#undef CALL
    return last_error();
}

int hpf_hip_fold_in_f32(const int32_t *idx, const float *y, int64_t n, const float *e_items, const float *cs_other,
                        float *shp, float *rte, float *fac, float *e_last, int32_t *rounds, float prior, float top,
                        float add, float rs, float stop_thr, int maxiter, int k, int ld, void *stream) {
    if (!e_items || !cs_other || !shp || !rte || !fac || !e_last || !rounds || n < 0 || (n > 0 && (!idx || !y)) ||
        maxiter < 0 || k <= 0 || ld != hpf_hip_ld_for_k(k))
        return HPF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
#define CALL(LD)                                                                                                     \
    hipLaunchKernelGGL((fold_in_kernel<LD>), dim3(1), dim3(BLOCK), 0, st, idx, y, n, e_items, cs_other, shp, rte, fac, \
                       e_last, rounds, prior, top, add, rs, stop_thr, maxiter, k);
    HPF_DISPATCH_LD1(ld, CALL)
#undef CALL
    return last_error();
}

int hpf_hip_uniform_rows_f32(const uint32_t *raw, float *out, const float *den, float *ratio, int64_t nrows, float base,
                             float scale, int k, int ld, void *stream) {
    if (nrows == 0) return 0;
    if (!raw || !out || nrows < 0 || k <= 0 || ld < k || (ratio && !den)) return HPF_EINVAL;
    const int grid = clamp_grid((nrows * k + BLOCK - 1) / BLOCK, 8192);
    hipLaunchKernelGGL(uniform_rows_kernel, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, raw, out, den, ratio,
                       (long long)nrows, k, ld, base, scale);
    return last_error();
}

}  // extern "C"

// ---- direct-exchange forms (hpf_internal.h; used by hpf_shard.hip) -------------------------------------------------------
namespace hpf_direct {

int sweep(const hpf_segment *segs, int64_t nseg, const int32_t *idx, const float *y, const float *tab_self,
          const float *tab_other, float *part, float *acc_rows, int acc_ld, int k, int ld, int short_rows, int grid_blocks,
          Signal sig, hipStream_t st) {
    return sweep_impl(segs, nseg, idx, y, tab_self, tab_other, part, acc_rows, acc_ld, k, ld, short_rows, grid_blocks,
                      nullptr, sig, st);
}

int sweep_finalize(const hpf_segment *segs, int64_t nseg, const int32_t *idx, const float *y, const float *tab_self,
                   const float *tab_other, float *part, float *e_new, float *shp, float *rte, float *fac, float *rs,
                   float *rs_prev, const float *cs_other, float *cs_partial, float prior_shp, float top_shp,
                   float add_rte, int k, int ld, int grid_blocks, float *cs_other_copy, Signal sig, hipStream_t st) {
    return sweep_finalize_impl(segs, nseg, idx, y, tab_self, tab_other, part, e_new, shp, rte, fac, rs, rs_prev, cs_other,
                               cs_partial, prior_shp, top_shp, add_rte, k, ld, grid_blocks, cs_other_copy, sig, st);
}

int item_apply_blocks(const float *const *blocks, int nblocks, int wait_kind, int local_flag, uint32_t epoch,
                      const hpf_p2p::Peers &pp, const float *shp_own, float *e_tab, float *shp, float *fac, float *rs,
                      const float *cs_other, float *cs_partial, float add_rte, int k, int ld, int rank, int world,
                      int64_t nrows, int nranges, const int64_t *range_lo, const int64_t *range_hi, int grid_blocks,
                      hipStream_t st) {
    if (!blocks || nblocks != world || world > HPF_P2P_MAX_RANKS) return HPF_EINVAL;
    OwnerBlocks own = {};
    own.n = nblocks;
    own.wait_kind = wait_kind;
    own.local_flag = local_flag;
    own.epoch = epoch;
    for (int p = 0; p < nblocks; p++) {
        if (!blocks[p]) return HPF_EINVAL;
        own.block[p] = blocks[p];
    }
    own.peers = pp;
    return item_apply_impl(nullptr, own, shp_own, e_tab, shp, fac, rs, cs_other, cs_partial, add_rte, k, ld, rank, world,
                           nrows, nranges, range_lo, range_hi, grid_blocks, st);
}

int colsum_reduce_allreduce(const float *cs_partial, int nblk, float *cs_out, int ld, const hpf_p2p::Peers *peers_dev,
                            int which, uint32_t epoch, uint32_t then_wait_kinds, int then_wait_self, hipStream_t st) {
    if (!cs_partial || !cs_out || nblk <= 0 || ld < 32 || (ld & 3) || (reinterpret_cast<uintptr_t>(cs_partial) & 15) ||
        !peers_dev || which < 0 || which >= HPF_P2P_NVEC)
        return HPF_EINVAL;
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3(ld / 4), dim3(1024), 0, st, cs_partial, nblk, cs_out, ld,
                       peers_dev, which, epoch, then_wait_kinds, then_wait_self);
    return last_error();
}

}  // namespace hpf_direct
