/*
 * hpf_oracle.c -- CPU restatement of the hpfrec full-batch CAVI loops.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle for the HIP path in
 * hpfrec_amd/csrc.  Nothing under hpfrec_amd/ may import, link or call it; only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * It restates, loop for loop, the arithmetic of the reference's Cython kernels
 * (citations are /root/reference/hpfrec/cython_loops.pxi = "PXI"):
 *
 *   hpf_oracle_update_phi_f32      PXI:551-591  (update_phi, both branches)
 *   hpf_oracle_scatter_f32         PXI:613-621  (update_G_n_L_sh, serial COO order)
 *   hpf_oracle_llk_f32             PXI:627-658  (llk_plus_rmse)
 *   hpf_oracle_sum_prediction_f32  PXI:816-825  (sum_prediction)
 *   hpf_oracle_predict_f32         PXI:803-810  (predict_multiple)
 *   hpf_oracle_update_phi_csr_f32  PXI:666-692  (update_phi_csr, SVI)
 *   hpf_oracle_scatter_csr_f32     PXI:730-746  (update_G_n_L_sh_csr, SVI, serial)
 *
 * Precision contract of the reference float build (hpfrec/cython_float.pxi:7-10):
 * storage is float; psi/log/exp of the plain branch run in double and are rounded
 * to float on store; the max-subtracted branch uses expf; sums over k are
 * sequential float adds; llk accumulators are long double.
 *
 * Third-party arithmetic that is NOT under /root/reference:
 *   - scipy.special.cython_special.psi (scipy 1.15.3 here; reference pins
 *     scipy>=1.11.1).  Restated below from the published Cephes/xsf algorithm
 *     (psi.c, Moshier; [1,2] rational approximation after Boost.Math, Maddock 2006).
 *     tests/test_oracle.py checks it bit-for-bit against scipy.special.psi.
 *   - scipy.linalg.cython_blas.sdot (OpenBLAS; accumulation order is CPU-dispatch
 *     dependent).  Restated as a sequential float accumulation: llk values are
 *     therefore compared with a tolerance, never bitwise ("BLAS order unpinned").
 *   - glibc log/exp/logf/expf/lgamma: used directly, same image here and on the GPU box.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef uint64_t ind_t; /* reference: ctypedef size_t ind_type (cython_float_nonwindows.pyx:8) */

/* ------------------------------------------------------------------------- */
/* digamma: restatement of scipy's real-argument psi (Cephes/xsf)            */
/* ------------------------------------------------------------------------- */
static const double PSI_A[7] = {
    8.33333333333333333333E-2, -2.10927960927960927961E-2, 7.57575757575757575758E-3,
    -4.16666666666666666667E-3, 3.96825396825396825397E-3, -8.33333333333333333333E-3,
    8.33333333333333333333E-2};

static const double PSI_P12[6] = {-0.0020713321167745952, -0.045251321448739056, -0.28919126444774784,
                                  -0.65031853770896507,   -0.32555031186804491,  0.25479851061131551};
static const double PSI_Q12[7] = {-0.55789841321675513e-6, 0.0021284987017821144, 0.054151797245674225,
                                  0.43593529692665969,     1.4606242909763515,    2.0767117023730469,
                                  1.0};

static const double EULER_GAMMA = 0.577215664901532860606512090082402431;

static inline double horner(double x, const double *c, int n) {
    /* Cephes polevl: highest-order coefficient first */
    double r = c[0];
    for (int i = 1; i <= n; i++) r = r * x + c[i];
    return r;
}

static double psi_unit_interval_1_2(double x) {
    const float Y = 0.99558162689208984f;
    const double root1 = 1569415565.0 / 1073741824.0;
    const double root2 = (381566830.0 / 1073741824.0) / 1073741824.0;
    const double root3 = 0.9016312093258695918615325266959189453125e-19;
    double g = x - root1;
    g -= root2;
    g -= root3;
    double r = horner(x - 1.0, PSI_P12, 5) / horner(x - 1.0, PSI_Q12, 6);
    return g * Y + g * r;
}

static double psi_large(double x) {
    double y = 0.0;
    if (x < 1.0e17) {
        double z = 1.0 / (x * x);
        y = z * horner(z, PSI_A, 6);
    }
    return log(x) - (0.5 / x) - y;
}

double hpf_oracle_digamma(double x) {
    double y = 0.0;
    if (isnan(x)) return x;
    if (x == INFINITY) return x;
    if (x == -INFINITY) return NAN;
    if (x == 0.0) return copysign(INFINITY, -x);
    if (x < 0.0) {
        /* reflection; never reached on the HPF path (all shapes > 0) */
        double ipart;
        double r = modf(x, &ipart);
        if (r == 0.0) return NAN;
        y = -M_PI / tan(M_PI * r);
        x = 1.0 - x;
    }
    if (x <= 10.0 && x == floor(x)) {
        int n = (int)x;
        for (int i = 1; i < n; i++) y += 1.0 / i;
        y -= EULER_GAMMA;
        return y;
    }
    if (x < 1.0) {
        y -= 1.0 / x;
        x += 1.0;
    } else if (x < 10.0) {
        while (x > 2.0) {
            x -= 1.0;
            y += 1.0 / x;
        }
    }
    if (1.0 <= x && x <= 2.0) {
        y += psi_unit_interval_1_2(x);
        return y;
    }
    y += psi_large(x);
    return y;
}

void hpf_oracle_digamma_vec(const double *x, double *out, int64_t n) {
    for (int64_t i = 0; i < n; i++) out[i] = hpf_oracle_digamma(x[i]);
}

/* ------------------------------------------------------------------------- */
/* phi (multinomial responsibilities times Y), COO order -- PXI:551-591        */
/* ------------------------------------------------------------------------- */
void hpf_oracle_update_phi_f32(const float *G_sh, const float *G_rt, const float *L_sh,
                               const float *L_rt, float *phi, const float *Y, ind_t k,
                               int sum_exp_trick, const ind_t *ix_u, const ind_t *ix_i, ind_t nY,
                               int nthreads) {
    int64_t n = (int64_t)nY;
    (void)nthreads;
    if (sum_exp_trick) {
        /* PXI:561-577: float store of the double sum, float max, expf */
#pragma omp parallel for schedule(static) num_threads(nthreads)
        for (int64_t i = 0; i < n; i++) {
            const float *gs = G_sh + k * ix_u[i], *gr = G_rt + k * ix_u[i];
            const float *ls = L_sh + k * ix_i[i], *lr = L_rt + k * ix_i[i];
            float *p = phi + (ind_t)i * k;
            float sumphi = 0.0f, maxval = -HUGE_VALF;
            for (ind_t j = 0; j < k; j++) {
                p[j] = (float)(hpf_oracle_digamma((double)gs[j]) - log((double)gr[j]) +
                               hpf_oracle_digamma((double)ls[j]) - log((double)lr[j]));
                if (p[j] > maxval) maxval = p[j];
            }
            for (ind_t j = 0; j < k; j++) {
                p[j] = expf(p[j] - maxval);
                sumphi += p[j];
            }
            float scale = Y[i] / sumphi;
            for (ind_t j = 0; j < k; j++) p[j] *= scale;
        }
    } else {
        /* PXI:580-591: exp in double of the double sum, rounded to float on store */
#pragma omp parallel for schedule(static) num_threads(nthreads)
        for (int64_t i = 0; i < n; i++) {
            const float *gs = G_sh + k * ix_u[i], *gr = G_rt + k * ix_u[i];
            const float *ls = L_sh + k * ix_i[i], *lr = L_rt + k * ix_i[i];
            float *p = phi + (ind_t)i * k;
            float sumphi = 0.0f;
            for (ind_t j = 0; j < k; j++) {
                p[j] = (float)exp(hpf_oracle_digamma((double)gs[j]) - log((double)gr[j]) +
                                  hpf_oracle_digamma((double)ls[j]) - log((double)lr[j]));
                sumphi += p[j];
            }
            float scale = Y[i] / sumphi;
            for (ind_t j = 0; j < k; j++) p[j] *= scale;
        }
    }
}

/* ------------------------------------------------------------------------- */
/* shape scatter, serial COO order (the deterministic default) -- PXI:613-621  */
/* ------------------------------------------------------------------------- */
void hpf_oracle_scatter_f32(float *G_sh, float *L_sh, const float *phi, ind_t k, const ind_t *ix_u,
                            const ind_t *ix_i, ind_t nY) {
    for (ind_t i = 0; i < nY; i++) {
        float *g = G_sh + ix_u[i] * k;
        float *l = L_sh + ix_i[i] * k;
        const float *p = phi + i * k;
        for (ind_t j = 0; j < k; j++) {
            g[j] += p[j];
            l[j] += p[j];
        }
    }
}

/* ------------------------------------------------------------------------- */
/* Poisson llk (+ squared error) -- PXI:627-658                                */
/* ------------------------------------------------------------------------- */
static inline float dot_f32(const float *a, const float *b, ind_t k) {
    /* stand-in for BLAS sdot (order unpinned, see header) */
    float s = 0.0f;
    for (ind_t j = 0; j < k; j++) s += a[j] * b[j];
    return s;
}

void hpf_oracle_llk_f32(const float *T, const float *B, const float *Y, const ind_t *ix_u,
                        const ind_t *ix_i, ind_t nY, ind_t k, long double *out, int nthreads,
                        int add_mse, int full_llk) {
    long double out1 = 0.0L, out2 = 0.0L;
    int64_t n = (int64_t)nY;
    (void)nthreads;
#pragma omp parallel for schedule(static) num_threads(nthreads) reduction(+ : out1, out2)
    for (int64_t i = 0; i < n; i++) {
        float yhat = dot_f32(T + ix_u[i] * k, B + ix_i[i] * k, k);
        if (full_llk)
            out1 += (double)Y[i] * log((double)yhat) - lgamma((double)Y[i] + 1.0);
        else
            out1 += Y[i] * logf(yhat);
        if (add_mse) {
            /* (Y[i] - yhat)**2 on C floats is a float expression (PXI:642,647) */
            float d = Y[i] - yhat;
            out2 += d * d;
        }
    }
    out[0] = out1;
    if (add_mse) out[1] = out2;
}

long double hpf_oracle_sum_prediction_f32(const float *M1, const float *M2, const ind_t *ix_u,
                                          const ind_t *ix_i, ind_t n, int k, int nthreads) {
    long double acc = 0.0L;
    int64_t nn = (int64_t)n;
    (void)nthreads;
#pragma omp parallel for schedule(static) num_threads(nthreads) reduction(+ : acc)
    for (int64_t i = 0; i < nn; i++) acc += dot_f32(M1 + ix_u[i] * (ind_t)k, M2 + ix_i[i] * (ind_t)k, (ind_t)k);
    return acc;
}

/* wrappers that hand long double results back through doubles + a split, so the
 * ctypes side does not depend on numpy's longdouble ABI */
void hpf_oracle_llk_f32_d(const float *T, const float *B, const float *Y, const ind_t *ix_u,
                          const ind_t *ix_i, ind_t nY, ind_t k, double *out_hi, double *out_lo,
                          int nthreads, int add_mse, int full_llk) {
    long double o[2] = {0.0L, 0.0L};
    hpf_oracle_llk_f32(T, B, Y, ix_u, ix_i, nY, k, o, nthreads, add_mse, full_llk);
    for (int t = 0; t < 2; t++) {
        out_hi[t] = (double)o[t];
        out_lo[t] = (double)(o[t] - (long double)out_hi[t]);
    }
}

void hpf_oracle_sum_prediction_f32_d(const float *M1, const float *M2, const ind_t *ix_u,
                                     const ind_t *ix_i, ind_t n, int k, int nthreads, double *hi,
                                     double *lo) {
    long double s = hpf_oracle_sum_prediction_f32(M1, M2, ix_u, ix_i, n, k, nthreads);
    *hi = (double)s;
    *lo = (double)(s - (long double)*hi);
}

void hpf_oracle_predict_f32(float *out, const float *M1, const float *M2, const ind_t *ix_u,
                            const ind_t *ix_i, ind_t n, int k, int nthreads) {
    int64_t nn = (int64_t)n;
    (void)nthreads;
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int64_t i = 0; i < nn; i++) out[i] = dot_f32(M1 + ix_u[i] * (ind_t)k, M2 + ix_i[i] * (ind_t)k, (ind_t)k);
}

/* ------------------------------------------------------------------------- */
/* SVI: phi over a list of CSR rows (always max-subtracted, expf) PXI:666-692  */
/* "G" is the batch side (rows listed in u_arr), "L" the gathered side.        */
/* ------------------------------------------------------------------------- */
void hpf_oracle_update_phi_csr_f32(const float *G_sh, const float *G_rt, const float *L_sh,
                                   const float *L_rt, float *phi, const float *Y, const ind_t *ix_i,
                                   const ind_t *st_ix_u, const ind_t *u_arr, ind_t k, ind_t nU,
                                   int nthreads) {
    int64_t n = (int64_t)nU;
    (void)nthreads;
#pragma omp parallel for schedule(dynamic) num_threads(nthreads)
    for (int64_t u = 0; u < n; u++) {
        ind_t uid = u_arr[u];
        const float *gs = G_sh + k * uid, *gr = G_rt + k * uid;
        for (ind_t y_ix = st_ix_u[uid]; y_ix < st_ix_u[uid + 1]; y_ix++) {
            const float *ls = L_sh + k * ix_i[y_ix], *lr = L_rt + k * ix_i[y_ix];
            float *p = phi + y_ix * k;
            float sumrow = 0.0f, maxval = -HUGE_VALF;
            for (ind_t j = 0; j < k; j++) {
                p[j] = (float)(hpf_oracle_digamma((double)gs[j]) - log((double)gr[j]) +
                               hpf_oracle_digamma((double)ls[j]) - log((double)lr[j]));
                if (p[j] > maxval) maxval = p[j];
            }
            for (ind_t j = 0; j < k; j++) {
                p[j] = expf(p[j] - maxval);
                sumrow += p[j];
            }
            float scale = Y[y_ix] / sumrow;
            for (ind_t j = 0; j < k; j++) p[j] *= scale;
        }
    }
}

/* PXI:730-746, run serially (the reference's prange races on the L side; the
 * single-thread order is the only reproducible one) */
void hpf_oracle_scatter_csr_f32(float *G_sh, float *L_sh, const float *phi, ind_t k, ind_t nU,
                                const ind_t *ix_i, const ind_t *st_ix_u, const ind_t *u_arr) {
    for (ind_t u = 0; u < nU; u++) {
        ind_t uid = u_arr[u];
        float *g = G_sh + uid * k;
        for (ind_t y_ix = st_ix_u[uid]; y_ix < st_ix_u[uid + 1]; y_ix++) {
            float *l = L_sh + ix_i[y_ix] * k;
            const float *p = phi + y_ix * k;
            for (ind_t j = 0; j < k; j++) {
                g[j] += p[j];
                l[j] += p[j];
            }
        }
    }
}

int hpf_oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
