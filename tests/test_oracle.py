"""The CPU oracle against the golden vectors produced by the real reference
(tests/golden/make_golden.py).  Bit-exact for the eight variational arrays."""
import os

import numpy as np
import pytest
import scipy.special as sp

import datagen
from conftest import GOLDEN
from oracle import hpf_oracle as O


def test_digamma_bit_exact_vs_scipy():
    rng = np.random.default_rng(0)
    for x in (np.exp(rng.uniform(np.log(1e-3), np.log(1e8), 200_000)), rng.uniform(0.25, 12, 200_000),
              np.float32(rng.uniform(0.29, 3, 100_000)).astype(np.float64), np.arange(1, 20, dtype=np.float64)):
        assert np.array_equal(O.digamma(x), sp.psi(x))


@pytest.mark.parametrize("trick,fname,its", [(0, "c1_full.npz", (1, 2, 5, 10, 20)), (1, "c1_trick.npz", (1, 10))])
def test_c1_arrays_bit_exact(trick, fname, its):
    df, nU, nI = datagen.readme_counts()
    Y, iu, ii = datagen.triplets(df)
    assert Y.shape[0] == 6347
    g = np.load(os.path.join(GOLDEN, fname))
    st, caps = O.fit_full_batch(Y, iu, ii, nU, nI, 30, max(its), 123, capture_at=its, sum_exp_trick=trick)
    for it in its:
        for n in O.State.names:
            assert np.array_equal(caps[it][n], g["it%d_%s" % (it, n)]), (it, n)


def test_c1_nondefault_hyper_bit_exact():
    df, nU, nI = datagen.readme_counts()
    Y, iu, ii = datagen.triplets(df)
    g = np.load(os.path.join(GOLDEN, "c1_hyper.npz"))
    st, caps = O.fit_full_batch(Y, iu, ii, nU, nI, 7, 5, 5, a=0.05, a_prime=0.7, b_prime=2.0, c=0.02, c_prime=1.3,
                                d_prime=0.5, capture_at=(5,))
    for n in O.State.names:
        assert np.array_equal(caps[5][n], g["it5_%s" % n]), n


def test_c1_llk_scalars():
    df, nU, nI = datagen.readme_counts()
    Y, iu, ii = datagen.triplets(df)
    g = np.load(os.path.join(GOLDEN, "c1_full.npz"))
    st, caps = O.fit_full_batch(Y, iu, ii, nU, nI, 30, 20, 123, capture_at=(10,))
    # BLAS sdot order is unpinned -> tolerance, not bitwise
    assert abs(float(O.train_llk(st, Y, iu, ii)[0]) / g["train_llk_it20"] - 1) < 1e-6
    assert abs(float(O.calc_llk(Y, iu, ii, st.Theta, st.Beta, 30, 1, 0)) / g["eval_llk_it20"] - 1) < 1e-6
    assert abs(float(O.calc_llk(Y, iu, ii, st.Theta, st.Beta, 30, 1, 1)) / g["eval_llk_full_it20"] - 1) < 1e-6


def test_mid_bit_exact_and_thread_invariance():
    df, nU, nI = datagen.mid_counts()
    Y, iu, ii = datagen.triplets(df)
    g = np.load(os.path.join(GOLDEN, "mid_full.npz"))
    assert int(g["nnz"]) == Y.shape[0]
    st, caps = O.fit_full_batch(Y, iu, ii, nU, nI, 50, 10, 123, capture_at=(1, 5, 10), nthreads=O.max_threads())
    for it in (1, 5, 10):
        for n in O.State.names:
            assert np.array_equal(caps[it][n][::10], g["it%d_%s_rows" % (it, n)]), (it, n)
            assert np.array_equal(caps[it][n].astype(np.float64).sum(axis=0), g["it%d_%s_colsum64" % (it, n)])
    assert abs(float(O.train_llk(st, Y, iu, ii)[0]) / g["train_llk_it10"] - 1) < 1e-6


def test_partial_fit_bit_exact():
    batches, nU, nI = datagen.partial_fit_batches()
    g = np.load(os.path.join(GOLDEN, "c1_partial_fit.npz"))
    hy = O.Hyper(30, 0.3, 0.3, 1.0, 0.3, 0.3, 1.0)
    st = O.State(nU, nI, hy, 123)
    niter = 0
    step_fn = lambda x: 1 / np.sqrt(x + 2)
    for b, (kind, bdf) in enumerate(batches):
        Y, iu, ii = datagen.triplets(bdf)
        users = np.unique(iu)
        items = np.unique(ii)
        # hpfrec/__init__.py:834-847,912: first call uses step 1.0 (niter None), then step_size(niter);
        # multiplier is nusers/len(users_in_batch) for both batch types
        step = 1.0 if b == 0 else step_fn(niter)
        O.partial_fit_step(st, hy, Y, iu, ii, users, items, step, float(nU) / users.shape[0], kind == "users")
        niter += 1
        assert niter == int(g["call%d_niter" % (b + 1)])
        for n in O.State.names:
            assert np.array_equal(getattr(st, n), g["call%d_%s" % (b + 1, n)]), (b, n)


@pytest.mark.parametrize("tag,upb,ipb", [("both", 20, 25), ("users", 30, 0), ("items", 0, 40), ("both_fullphi", 20, 25)])
def test_svi_epochs_bit_exact(tag, upb, ipb):
    """O.fit_svi (the stochastic epochs of fit_hpf, PXI:262-377) against the reference's ncores=1 captures: 4 epochs
    on the README data, users-only / items-only / alternating."""
    df, nU, nI = datagen.readme_counts()
    Y, iu, ii = datagen.triplets(df)
    g = np.load(os.path.join(GOLDEN, "c1_svi.npz"))
    st_ix_u = np.zeros(1, np.uint64)
    if upb > 0:
        Y, iu, ii, st_ix_u = O.svi_inputs_like_reference(Y, iu, ii, nU, nI)
    st = O.fit_svi(Y, iu, ii, st_ix_u, nU, nI, 30, 4, 123, upb, ipb)
    for n in O.State.names:
        assert np.array_equal(getattr(st, n), g["%s_%s" % (tag, n)]), (tag, n)


SVI_LARGE = (("both", 2, 8192, 8192), ("both", 3, 8192, 8192), ("users", 2, 8192, 0), ("items", 2, 0, 8192))


@pytest.mark.parametrize("tag,epochs,upb,ipb", SVI_LARGE)
def test_svi_large_bit_exact(tag, epochs, upb, ipb):
    """60k x 50k, 2.4M nonzeros, k = 50, 8192-row batches: O.fit_svi against the REAL reference (tests/golden/
    svi_large.npz, made by make_golden.py svi_large at ncores=1) at a size where the whole-table float32 column sums
    every batch takes (PXI:300,318,352,370) round visibly.  Sub-sampled rows bit-exact, float64 column sums of every
    array bit-exact: item + user epoch, item-user-item, users only, items only."""
    u, i, y, nU, nI = datagen.svi_large_counts()
    g = np.load(os.path.join(GOLDEN, "svi_large.npz"))
    assert int(g["nnz"]) == y.shape[0] >= 2_000_000
    st_ix_u = np.zeros(1, np.uint64)
    if upb > 0:
        y, u, i, st_ix_u = O.svi_inputs_like_reference(y, u, i, nU, nI)
    st = O.fit_svi(y, u, i, st_ix_u, nU, nI, 50, epochs, 123, upb, ipb, nthreads=O.max_threads())
    for n in O.State.names:
        v = getattr(st, n)
        assert np.array_equal(v[::125], g["%s_ep%d_%s_rows" % (tag, epochs, n)]), (tag, epochs, n)
        assert np.array_equal(v.astype(np.float64).sum(axis=0), g["%s_ep%d_%s_colsum64" % (tag, epochs, n)]), (tag, n)


def test_large_bit_exact():
    """200k x 50k, 5.4M nonzeros, k = 50: the oracle against the REAL reference (tests/golden/large_full.npz, made by
    make_golden.py large_full) where numpy's sequential float32 column sums over 2e5 rows (PXI:236,255) carry visible
    rounding -- the size class the GPU path's 1e-4 claim is about.  Sub-sampled rows bit-exact, float64 column sums of
    every array bit-exact, after 1, 3 and 5 iterations."""
    u, i, y, nU, nI = datagen.large_counts()
    g = np.load(os.path.join(GOLDEN, "large_full.npz"))
    assert int(g["nnz"]) == y.shape[0] >= 5_000_000
    st, caps = O.fit_full_batch(y, u, i, nU, nI, 50, 5, 123, capture_at=(1, 3, 5), nthreads=O.max_threads())
    for it in (1, 3, 5):
        for n in O.State.names:
            step = 400 if caps[it][n].shape[0] == nU else 100
            assert np.array_equal(caps[it][n][::step], g["it%d_%s_rows" % (it, n)]), (it, n)
            assert np.array_equal(caps[it][n].astype(np.float64).sum(axis=0), g["it%d_%s_colsum64" % (it, n)]), (it, n)
    assert abs(float(O.train_llk(st, y, u, i)[0]) / g["train_llk_it5"] - 1) < 1e-6
