"""CPU oracle for the hpfrec full-batch CAVI path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; nothing under hpfrec_amd/ does.  It restates the reference's driver
(/root/reference/hpfrec/cython_loops.pxi, "PXI" below) on top of the C loops in
hpf_oracle.c, using numpy for the rate updates exactly where the reference uses
numpy (PXI:236-259), so reduction orders (naive row adds for axis=0, numpy's
pairwise blocks for axis=1) are the reference's own.

Parity status: PINNED.  tests/golden/*.npz were produced by the real reference,
compiled and imported in the build container (tests/golden/make_golden.py); the
eight variational arrays of this oracle match those fixtures bit-for-bit
(tests/test_oracle.py).  llk scalars go through BLAS sdot in the reference, whose
accumulation order depends on OpenBLAS' CPU dispatch, so they are compared at 1e-6
relative ("BLAS order unpinned").
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_real_t = ctypes.c_float          # hpfrec/cython_float.pxi:9
obj_ind_type = ctypes.c_size_t     # hpfrec/cython_float_nonwindows.pyx:10


def build(force=False):
    so = os.path.join(_HERE, "libhpf_oracle.so")
    src = os.path.join(_HERE, "hpf_oracle.c")
    if force or (not os.path.exists(so)) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libhpf_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        vp, i64, u64, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64, ctypes.c_int
        L.hpf_oracle_digamma_vec.argtypes = [vp, vp, i64]
        L.hpf_oracle_update_phi_f32.argtypes = [vp, vp, vp, vp, vp, vp, u64, ci, vp, vp, u64, ci]
        L.hpf_oracle_scatter_f32.argtypes = [vp, vp, vp, u64, vp, vp, u64]
        L.hpf_oracle_llk_f32_d.argtypes = [vp, vp, vp, vp, vp, u64, u64, vp, vp, ci, ci, ci]
        L.hpf_oracle_sum_prediction_f32_d.argtypes = [vp, vp, vp, vp, u64, ci, ci, vp, vp]
        L.hpf_oracle_predict_f32.argtypes = [vp, vp, vp, vp, vp, u64, ci, ci]
        L.hpf_oracle_update_phi_csr_f32.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, u64, u64, ci]
        L.hpf_oracle_scatter_csr_f32.argtypes = [vp, vp, vp, u64, u64, vp, vp, vp]
        L.hpf_oracle_max_threads.restype = ci
        for f in (L.hpf_oracle_digamma_vec, L.hpf_oracle_update_phi_f32, L.hpf_oracle_scatter_f32,
                  L.hpf_oracle_llk_f32_d, L.hpf_oracle_sum_prediction_f32_d, L.hpf_oracle_predict_f32,
                  L.hpf_oracle_update_phi_csr_f32, L.hpf_oracle_scatter_csr_f32):
            f.restype = None
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data


def _f32(a):
    return np.require(a, dtype=np.float32, requirements=["C_CONTIGUOUS", "ALIGNED"])


def _ind(a):
    return np.require(a, dtype=np.uint64, requirements=["C_CONTIGUOUS", "ALIGNED"])


def digamma(x):
    x = np.require(x, dtype=np.float64, requirements=["C_CONTIGUOUS"])
    out = np.empty_like(x)
    lib().hpf_oracle_digamma_vec(_p(x), _p(out), x.size)
    return out


# --------------------------------------------------------------------------- #
# PXI:117-143
# --------------------------------------------------------------------------- #
def initialize_parameters(Theta, Beta, random_seed, a, a_prime, b_prime, c, c_prime, d_prime):
    nU, k = Theta.shape
    nI = Beta.shape[0]
    bitgen = np.random.MT19937(seed=random_seed if random_seed > 0 else None)
    rng = np.random.Generator(bitgen)
    # draw order is part of the contract: user rate, item rate, user shape, item shape
    u_rate = rng.random(size=(nU, k), dtype=np.float32)
    i_rate = rng.random(size=(nI, k), dtype=np.float32)
    u_shape = rng.random(size=(nU, k), dtype=np.float32)
    i_shape = rng.random(size=(nI, k), dtype=np.float32)
    Gamma_rte = a_prime + 0.01 * u_rate
    Lambda_rte = c_prime + 0.01 * i_rate
    Gamma_shp = a_prime + 0.01 * u_shape
    Lambda_shp = c_prime + 0.01 * i_shape
    k_rte = np.full((nU, 1), b_prime, dtype=np.float32)
    t_rte = np.full((nI, 1), d_prime, dtype=np.float32)
    Theta[:, :] = Gamma_shp / Gamma_rte
    Beta[:, :] = Lambda_shp / Lambda_rte
    return Gamma_shp, Gamma_rte, Lambda_shp, Lambda_rte, k_rte, t_rte


class Hyper:
    """float32-rounded hyper-parameters and the derived constants of PXI:173-174,209-210."""

    def __init__(self, k, a, a_prime, b_prime, c, c_prime, d_prime):
        f = np.float32
        self.k = int(k)
        self.a, self.a_prime, self.b_prime = f(a), f(a_prime), f(b_prime)
        self.c, self.c_prime, self.d_prime = f(c), f(c_prime), f(d_prime)
        self.k_shp = f(self.a_prime + f(self.k) * self.a)
        self.t_shp = f(self.c_prime + f(self.k) * self.c)
        self.add_k_rte = f(self.a_prime / self.b_prime)
        self.add_t_rte = f(self.c_prime / self.d_prime)


class State:
    """The eight arrays of the variational state, laid out as in the reference."""

    names = ("Theta", "Beta", "Gamma_shp", "Gamma_rte", "Lambda_shp", "Lambda_rte", "k_rte", "t_rte")

    def __init__(self, nU, nI, hyper, random_seed):
        self.Theta = np.empty((nU, hyper.k), dtype=np.float32)
        self.Beta = np.empty((nI, hyper.k), dtype=np.float32)
        (self.Gamma_shp, self.Gamma_rte, self.Lambda_shp, self.Lambda_rte, self.k_rte,
         self.t_rte) = initialize_parameters(self.Theta, self.Beta, random_seed, float(hyper.a),
                                             float(hyper.a_prime), float(hyper.b_prime), float(hyper.c),
                                             float(hyper.c_prime), float(hyper.d_prime))

    def as_dict(self):
        return {n: getattr(self, n).copy() for n in self.names}


def cavi_iteration(st, hy, Y, ix_u, ix_i, phi, sum_exp_trick=0, nthreads=1, exact_colsums=False):
    """One full-batch sweep, PXI:232-259, in the reference's statement order.

    exact_colsums=True is NOT the reference: the two column sums (PXI:236, PXI:255) are then accumulated in float64
    instead of numpy's row-by-row float32 adds, whose own error reaches 1e-4 relative at 10^5..10^6 rows (SURVEY.md
    section 7).  Used only to show where a deviation at that scale comes from."""
    L = lib()
    k = hy.k
    nY = Y.shape[0]

    def colsum(a):
        if exact_colsums:
            return a.sum(axis=0, keepdims=True, dtype=np.float64).astype(np.float32)
        return a.sum(axis=0, keepdims=True)

    L.hpf_oracle_update_phi_f32(_p(st.Gamma_shp), _p(st.Gamma_rte), _p(st.Lambda_shp), _p(st.Lambda_rte),
                                _p(phi), _p(Y), k, int(sum_exp_trick), _p(ix_u), _p(ix_i), nY, int(nthreads))
    st.Gamma_rte = float(hy.k_shp) / st.k_rte + colsum(st.Beta)
    st.Gamma_shp[:, :] = float(hy.a)
    st.Lambda_shp[:, :] = float(hy.c)
    L.hpf_oracle_scatter_f32(_p(st.Gamma_shp), _p(st.Lambda_shp), _p(phi), k, _p(ix_u), _p(ix_i), nY)
    st.Theta[:, :] = st.Gamma_shp / st.Gamma_rte
    st.Lambda_rte = float(hy.t_shp) / st.t_rte + colsum(st.Theta)
    st.Beta[:, :] = st.Lambda_shp / st.Lambda_rte
    st.k_rte = float(hy.add_k_rte) + st.Theta.sum(axis=1, keepdims=True)
    st.t_rte = float(hy.add_t_rte) + st.Beta.sum(axis=1, keepdims=True)


def llk_plus_rmse(Theta, Beta, Y, ix_u, ix_i, nthreads=1, add_mse=1, full_llk=0):
    """PXI:627-658 -> (sum Y log(yhat) [- lgamma], sum sq err) as np.longdouble."""
    hi = np.zeros(2)
    lo = np.zeros(2)
    lib().hpf_oracle_llk_f32_d(_p(Theta), _p(Beta), _p(Y), _p(ix_u), _p(ix_i), Y.shape[0], Theta.shape[1],
                               _p(hi), _p(lo), int(nthreads), int(add_mse), int(full_llk))
    return np.longdouble(hi) + np.longdouble(lo)


def sum_prediction(Theta, Beta, ix_u, ix_i, nthreads=1):
    hi = np.zeros(1)
    lo = np.zeros(1)
    lib().hpf_oracle_sum_prediction_f32_d(_p(Theta), _p(Beta), _p(ix_u), _p(ix_i), ix_u.shape[0],
                                          Theta.shape[1], int(nthreads), _p(hi), _p(lo))
    return np.longdouble(hi[0]) + np.longdouble(lo[0])


def train_llk(st, Y, ix_u, ix_i, nthreads=1, full_llk=0):
    """PXI:75-79: nnz term minus (sum_u Theta).(sum_i Beta); returns (llk, rmse)."""
    e = llk_plus_rmse(st.Theta, st.Beta, Y, ix_u, ix_i, nthreads, 1, full_llk)
    llk = e[0] - st.Theta.sum(axis=0).dot(st.Beta.sum(axis=0))
    return llk, np.sqrt(e[1] / Y.shape[0])


def calc_llk(Y, ix_u, ix_i, Theta, Beta, k, nthreads, full_llk):
    """PXI:525-534 (HPF.eval_llk): nnz term minus the sum of predictions over listed pairs."""
    Y, ix_u, ix_i = _f32(Y), _ind(ix_u), _ind(ix_i)
    e = llk_plus_rmse(Theta, Beta, Y, ix_u, ix_i, nthreads, 0, full_llk)
    return e[0] - sum_prediction(Theta, Beta, ix_u, ix_i, nthreads)


def predict_arr(M1, M2, ix_u, ix_i, nthreads=1):
    """PXI:538-543"""
    ix_u, ix_i = _ind(ix_u), _ind(ix_i)
    out = np.zeros(ix_u.shape[0], dtype=np.float32)
    lib().hpf_oracle_predict_f32(_p(out), _p(M1), _p(M2), _p(ix_u), _p(ix_i), ix_u.shape[0], M1.shape[1],
                                 int(nthreads))
    return out


def fit_full_batch(Y, ix_u, ix_i, nU, nI, k, maxiter, random_seed, a=0.3, a_prime=0.3, b_prime=1.0,
                   c=0.3, c_prime=0.3, d_prime=1.0, sum_exp_trick=0, nthreads=1, capture_at=(),
                   state=None):
    """Run `maxiter` full-batch iterations (PXI:227-259) from the reference's initialisation.

    Returns (state, captures) where captures[it] is a dict of the eight arrays after
    `it` iterations (1-based count) for every it in capture_at.
    """
    Y, ix_u, ix_i = _f32(Y), _ind(ix_u), _ind(ix_i)
    hy = Hyper(k, a, a_prime, b_prime, c, c_prime, d_prime)
    st = state if state is not None else State(nU, nI, hy, random_seed)
    phi = np.empty((Y.shape[0], k), dtype=np.float32)
    caps = {}
    for it in range(maxiter):
        cavi_iteration(st, hy, Y, ix_u, ix_i, phi, sum_exp_trick, nthreads)
        if (it + 1) in capture_at:
            caps[it + 1] = st.as_dict()
    return st, caps


def partial_fit_step(st, hy, Y_batch, ix_u_batch, ix_i_batch, users_this_batch, items_this_batch,
                     step_size_batch, multiplier_batch, user_batch, nthreads=1):
    """PXI:423-473 (cython partial_fit): one SVI step on caller-supplied triplets."""
    L = lib()
    f = np.float32
    Y_batch, ix_u_batch, ix_i_batch = _f32(Y_batch), _ind(ix_u_batch), _ind(ix_i_batch)
    k = hy.k
    n = Y_batch.shape[0]
    step = float(f(step_size_batch))
    mult = float(f(multiplier_batch))
    step_prev = float(f(1) - f(step_size_batch))
    phi = np.empty((n, k), dtype=np.float32)
    L.hpf_oracle_update_phi_f32(_p(st.Gamma_shp), _p(st.Gamma_rte), _p(st.Lambda_shp), _p(st.Lambda_rte),
                                _p(phi), _p(Y_batch), k, 1, _p(ix_u_batch), _p(ix_i_batch), n, int(nthreads))
    if user_batch:
        st.Gamma_rte[:, :] = float(hy.k_shp) / st.k_rte + st.Beta.sum(axis=0, keepdims=True)
        Lambda_shp_prev = st.Lambda_shp[items_this_batch, :].copy()
    else:
        st.Lambda_rte[:, :] = float(hy.t_shp) / st.t_rte + st.Theta.sum(axis=0, keepdims=True)
        Gamma_shp_prev = st.Gamma_shp[users_this_batch, :].copy()
    st.Gamma_shp[users_this_batch, :] = float(hy.a)
    st.Lambda_shp[items_this_batch, :] = float(hy.c)
    L.hpf_oracle_scatter_f32(_p(st.Gamma_shp), _p(st.Lambda_shp), _p(phi), k, _p(ix_u_batch), _p(ix_i_batch), n)
    if user_batch:
        st.Lambda_shp[items_this_batch, :] = (step * mult * st.Lambda_shp[items_this_batch, :]
                                              + step_prev * Lambda_shp_prev)
        st.Theta[:, :] = st.Gamma_shp / st.Gamma_rte
        st.Lambda_rte[items_this_batch, :] = (
            step * (float(hy.t_shp) / st.t_rte[items_this_batch] + st.Theta.sum(axis=0, keepdims=False))
            + step_prev * st.Lambda_rte[items_this_batch, :])
        st.Beta[:, :] = st.Lambda_shp / st.Lambda_rte
    else:
        st.Gamma_shp[users_this_batch, :] = (step * mult * st.Gamma_shp[users_this_batch, :]
                                             + step_prev * Gamma_shp_prev)
        st.Beta[:, :] = st.Lambda_shp / st.Lambda_rte
        st.Gamma_rte[users_this_batch, :] = (
            step * (float(hy.k_shp) / st.k_rte[users_this_batch] + st.Beta.sum(axis=0, keepdims=False))
            + step_prev * st.Gamma_rte[users_this_batch, :])
        st.Theta[:, :] = st.Gamma_shp / st.Gamma_rte
    st.k_rte[:, :] = step * (float(hy.add_k_rte) + st.Theta.sum(axis=1, keepdims=True)) + step_prev * st.k_rte
    st.t_rte[:, :] = step * (float(hy.add_t_rte) + st.Beta.sum(axis=1, keepdims=True)) + step_prev * st.t_rte


def csc_data(ix_u, ix_i, Y, nU, nI):
    """get_csc_data, PXI:22-25 (scipy's coo -> csc: rows ascending inside a column, duplicate pairs summed)."""
    from scipy.sparse import coo_array
    X = coo_array((Y, (ix_u, ix_i)), shape=(nU, nI)).tocsc()
    return _ind(X.indptr), _ind(X.indices), _f32(X.data)


def svi_inputs_like_reference(Y, ix_u, ix_i, nU, nI):
    """What HPF.fit hands to fit_hpf in stochastic mode with users_per_batch > 0: the triplets sorted by user with
    pandas' default (unstable) sort -- twice, INIT:520 and INIT:600 -- and the CSR start indices from scipy's
    coo -> csr (INIT:589-599).  Returns (Y, ix_u, ix_i, st_ix_u)."""
    import pandas as pd
    from scipy.sparse import coo_array
    df = pd.DataFrame({"UserId": _ind(ix_u), "ItemId": _ind(ix_i), "Count": _f32(Y)})
    df.sort_values("UserId", inplace=True)
    X = coo_array((df["Count"].to_numpy(copy=False), (df["UserId"].to_numpy(copy=False), df["ItemId"].to_numpy(copy=False))),
                  shape=(nU, nI), dtype=np.float32).tocsr()
    df.sort_values("UserId", inplace=True)
    return (_f32(df["Count"].to_numpy()), _ind(df["UserId"].to_numpy()), _ind(df["ItemId"].to_numpy()),
            _ind(X.indptr))


def fit_svi(Y, ix_u, ix_i, st_ix_u, nU, nI, k, maxiter, random_seed, users_per_batch, items_per_batch,
            step_size=None, a=0.3, a_prime=0.3, b_prime=1.0, c=0.3, c_prime=0.3, d_prime=1.0, nthreads=1, state=None,
            exact_colsums=False):
    """The stochastic epochs of fit_hpf, PXI:262-377, statement for statement (alloc_full_phi=True form: phi rows
    indexed by the nonzero's position; the `_small` form only differs in where a phi row is stored).

    Y, ix_u, ix_i: as fit_hpf receives them (sorted by user when users_per_batch > 0, st_ix_u their CSR start
    indices -- see svi_inputs_like_reference).  The scatter runs serially in batch order: the reference's `prange`
    over batch rows races on the gathered side (PXI:745-746), nthreads=1 is its only reproducible order and the one
    the golden vectors were captured with.  `nthreads` here only parallelises phi (race-free).

    exact_colsums=True is NOT the reference (see cavi_iteration): Theta.sum(axis=0) / Beta.sum(axis=0) in float64."""
    L = lib()

    def colsum(arr, keepdims):
        if exact_colsums:
            return arr.sum(axis=0, keepdims=keepdims, dtype=np.float64).astype(np.float32)
        return arr.sum(axis=0, keepdims=keepdims)

    f = np.float32
    Y, ix_u, ix_i = _f32(Y), _ind(ix_u), _ind(ix_i)
    hy = Hyper(k, a, a_prime, b_prime, c, c_prime, d_prime)
    st = state if state is not None else State(nU, nI, hy, random_seed)
    if step_size is None:
        step_size = lambda x: 1 / np.sqrt(x + 2)      # noqa: E731  (INIT:208 default)
    k_shp, t_shp = float(hy.k_shp), float(hy.t_shp)
    add_k_rte, add_t_rte = float(hy.add_k_rte), float(hy.add_t_rte)
    nY = Y.shape[0]
    phi = np.empty((nY, k), dtype=np.float32)
    if items_per_batch > 0:
        items_numeration = np.arange(nI, dtype=obj_ind_type)
        nbatches_i = int(np.ceil(float(nI) / float(items_per_batch)))
        st_ix_i_copy, ix_u_copy, Ycopy = csc_data(ix_u, ix_i, Y, nU, nI)
        phi_i = np.empty((Ycopy.shape[0], k), dtype=np.float32)
    if users_per_batch != 0:
        users_numeration = np.arange(nU, dtype=obj_ind_type)
        nbatches_u = int(np.ceil(float(nU) / float(users_per_batch)))
        st_ix_u = _ind(st_ix_u)
    rng = np.random.default_rng(seed=random_seed if random_seed > 0 else None)

    def unique_other(rows, indptr, idx):
        if rows.shape[0] == 0:
            return np.empty(0, dtype=obj_ind_type)
        return np.unique(np.concatenate([idx[indptr[r]: indptr[r + 1]] for r in rows]))

    for i in range(maxiter):
        step = float(f(step_size(i)))                       # <real_t> step_size(i), boxed back to a Python float
        step_prev = float(f(1 - step))                      # cdef real_t
        if users_per_batch > 0 and items_per_batch > 0:
            user_epoch = ((i + 1) % 2) == 0
        else:
            user_epoch = users_per_batch > 0 and items_per_batch == 0
        if user_epoch:
            rng.shuffle(users_numeration)
            for bt in range(nbatches_u):
                st_b, end_b = bt * users_per_batch, min(nU, (bt + 1) * users_per_batch)
                users_tb = np.ascontiguousarray(users_numeration[st_b:end_b])
                mult = float(f(float(nU) / float(end_b - st_b)))
                items_tb = unique_other(users_tb, st_ix_u, ix_i)
                L.hpf_oracle_update_phi_csr_f32(_p(st.Gamma_shp), _p(st.Gamma_rte), _p(st.Lambda_shp), _p(st.Lambda_rte),
                                                _p(phi), _p(Y), _p(ix_i), _p(st_ix_u), _p(users_tb), k,
                                                users_tb.shape[0], int(nthreads))
                st.Gamma_rte = k_shp / st.k_rte + colsum(st.Beta, True)
                Lambda_shp_prev = st.Lambda_shp[items_tb, :].copy()
                st.Gamma_shp[users_tb, :] = float(hy.a)
                st.Lambda_shp[items_tb, :] = float(hy.c)
                L.hpf_oracle_scatter_csr_f32(_p(st.Gamma_shp), _p(st.Lambda_shp), _p(phi), k, users_tb.shape[0], _p(ix_i),
                                             _p(st_ix_u), _p(users_tb))
                st.Lambda_shp[items_tb, :] = step * mult * st.Lambda_shp[items_tb, :] + step_prev * Lambda_shp_prev
                st.Theta[:, :] = st.Gamma_shp / st.Gamma_rte
                st.Lambda_rte[items_tb, :] = (step * (t_shp / st.t_rte[items_tb] + colsum(st.Theta, False))
                                              + step_prev * st.Lambda_rte[items_tb, :])
                st.Beta[:, :] = st.Lambda_shp / st.Lambda_rte
                st.k_rte[users_tb] = (step * (add_k_rte + st.Theta[users_tb].sum(axis=1, keepdims=True))
                                      + step_prev * st.k_rte[users_tb])
                st.t_rte[items_tb] = (step * (add_t_rte + st.Beta[items_tb].sum(axis=1, keepdims=True))
                                      + step_prev * st.t_rte[items_tb])
        else:
            rng.shuffle(items_numeration)
            for bt in range(nbatches_i):
                st_b, end_b = bt * items_per_batch, min(nI, (bt + 1) * items_per_batch)
                items_tb = np.ascontiguousarray(items_numeration[st_b:end_b])
                mult = float(f(float(nI) / float(end_b - st_b)))
                users_tb = unique_other(items_tb, st_ix_i_copy, ix_u_copy)
                L.hpf_oracle_update_phi_csr_f32(_p(st.Lambda_shp), _p(st.Lambda_rte), _p(st.Gamma_shp), _p(st.Gamma_rte),
                                                _p(phi_i), _p(Ycopy), _p(ix_u_copy), _p(st_ix_i_copy), _p(items_tb), k,
                                                items_tb.shape[0], int(nthreads))
                st.Lambda_rte = t_shp / st.t_rte + colsum(st.Theta, True)
                Gamma_shp_prev = st.Gamma_shp[users_tb, :].copy()
                st.Gamma_shp[users_tb, :] = float(hy.a)
                st.Lambda_shp[items_tb, :] = float(hy.c)
                L.hpf_oracle_scatter_csr_f32(_p(st.Lambda_shp), _p(st.Gamma_shp), _p(phi_i), k, items_tb.shape[0],
                                             _p(ix_u_copy), _p(st_ix_i_copy), _p(items_tb))
                st.Gamma_shp[users_tb, :] = step * mult * st.Gamma_shp[users_tb, :] + step_prev * Gamma_shp_prev
                st.Beta[:, :] = st.Lambda_shp / st.Lambda_rte
                st.Gamma_rte[users_tb, :] = (step * (k_shp / st.k_rte[users_tb] + colsum(st.Beta, False))
                                             + step_prev * st.Gamma_rte[users_tb, :])
                st.Theta[:, :] = st.Gamma_shp / st.Gamma_rte
                st.k_rte[users_tb] = (step * (add_k_rte + st.Theta[users_tb].sum(axis=1, keepdims=True))
                                      + step_prev * st.k_rte[users_tb])
                st.t_rte[items_tb] = (step * (add_t_rte + st.Beta[items_tb].sum(axis=1, keepdims=True))
                                      + step_prev * st.t_rte[items_tb])
    return st


def max_threads():
    return int(lib().hpf_oracle_max_threads())
