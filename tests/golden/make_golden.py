"""Generate the golden vectors under tests/golden/ from the REAL reference.

Runs only in the build container, where the reference has been compiled in a scratch
copy OUTSIDE this repo:

    cp -r /root/reference /tmp/hpfrec_oracle && chmod -R u+w /tmp/hpfrec_oracle
    cd /tmp/hpfrec_oracle && python3 setup.py build_ext --inplace
    cd /root/repo && python tests/golden/make_golden.py

Only outputs (arrays/scalars the reference computed) are written; inputs are
regenerated from seeds by tests/datagen.py wherever the tests run.  All captures use
use_float=True, ncores=1, allow_inconsistent_math=False, reindex=False -- the
reference's deterministic path.
"""
import contextlib
import io
import os
import sys
import warnings

import numpy as np
import pandas as pd

REF = os.environ.get("HPFREC_REF_BUILD", "/tmp/hpfrec_oracle")
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from hpfrec import HPF  # noqa: E402  (the reference, imported from the scratch build)
import datagen  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
NAMES = ("Theta", "Beta", "Gamma_shp", "Gamma_rte", "Lambda_shp", "Lambda_rte", "k_rte", "t_rte")


def quiet(fn, *a, **kw):
    with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return fn(*a, **kw)


def grab(m):
    return {n: np.array(getattr(m, n)) for n in NAMES}


def fit_ref(df, k, maxiter, seed=123, **kw):
    args = dict(k=k, maxiter=maxiter, random_seed=seed, ncores=1, reindex=False, verbose=False,
                stop_crit="maxiter", check_every=None, keep_all_objs=True, allow_inconsistent_math=False,
                use_float=True)
    args.update(kw)
    m = HPF(**args)
    quiet(m.fit, df.copy())
    return m


def c1_full():
    df, nU, nI = datagen.readme_counts()
    out = {}
    for it in (1, 2, 5, 10, 20):
        m = fit_ref(df, 30, it)
        assert m.niter == it - 1
        for n, v in grab(m).items():
            out["it%d_%s" % (it, n)] = v
    for it in (10, 20):
        m = fit_ref(df, 30, it, verbose=True, check_every=it)
        out["train_llk_it%d" % it] = np.float64(m.train_llk)
    m = fit_ref(df, 30, 20)
    out["eval_llk_it20"] = np.float64(quiet(m.eval_llk, df.copy())["llk"])
    out["eval_llk_full_it20"] = np.float64(quiet(m.eval_llk, df.copy(), full_llk=True)["llk"])
    m = fit_ref(df, 30, 10, verbose=True, check_every=10, full_llk=True)
    out["train_llk_full_it10"] = np.float64(m.train_llk)
    # stopping rule: train-llk criterion, record where it stops
    m = fit_ref(df, 30, 200, stop_crit="train-llk", check_every=5, stop_thr=1e-3)
    out["trainllk_stop_niter"] = np.int64(m.niter)
    m = fit_ref(df, 30, 200, stop_crit="diff-norm", check_every=5, stop_thr=1e-1)
    out["diffnorm_stop_niter"] = np.int64(m.niter)
    np.savez_compressed(os.path.join(OUT, "c1_full.npz"), **out)
    print("c1_full", len(out))


def c1_trick():
    df, nU, nI = datagen.readme_counts()
    out = {}
    for it in (1, 10):
        m = fit_ref(df, 30, it, sum_exp_trick=True)
        for n, v in grab(m).items():
            out["it%d_%s" % (it, n)] = v
    np.savez_compressed(os.path.join(OUT, "c1_trick.npz"), **out)
    print("c1_trick", len(out))


def c1_hyper():
    """non-default hyper-parameters, small a/c (stress for the digamma recurrence) and k=7"""
    df, nU, nI = datagen.readme_counts()
    out = {}
    m = fit_ref(df, 7, 5, a=0.05, a_prime=0.7, b_prime=2.0, c=0.02, c_prime=1.3, d_prime=0.5, seed=5)
    for n, v in grab(m).items():
        out["it5_%s" % n] = v
    np.savez_compressed(os.path.join(OUT, "c1_hyper.npz"), **out)
    print("c1_hyper", len(out))


def mid_full():
    df, nU, nI = datagen.mid_counts()
    out = {"nnz": np.int64(df.shape[0])}
    for it in (1, 5, 10):
        m = fit_ref(df, 50, it)
        for n, v in grab(m).items():
            out["it%d_%s_rows" % (it, n)] = v[::10].copy()
            out["it%d_%s_colsum64" % (it, n)] = v.astype(np.float64).sum(axis=0)
    m = fit_ref(df, 50, 10, verbose=True, check_every=10)
    out["train_llk_it10"] = np.float64(m.train_llk)
    np.savez_compressed(os.path.join(OUT, "mid_full.npz"), **out)
    print("mid_full", len(out))


def large_full():
    """200k x 50k, >= 5M nonzeros, k = 50 through the REAL extension: sub-sampled rows (every 400th user, every 100th
    item) and float64 column sums of all eight arrays after 1, 3 and 5 iterations.  Pins the large-scale parity claim
    (numpy's naive float32 axis-0 sums, PXI:236,255, are the noisy side) with the reference itself."""
    u, i, y, nU, nI = datagen.large_counts()
    df = pd.DataFrame({"UserId": u.astype(np.int64), "ItemId": i.astype(np.int64), "Count": y})
    out = {"nnz": np.int64(df.shape[0])}
    for it in (1, 3, 5):
        m = fit_ref(df, 50, it)
        assert m.Theta.shape == (nU, 50) and m.Beta.shape == (nI, 50)
        for n, v in grab(m).items():
            step = 400 if v.shape[0] == nU else 100
            out["it%d_%s_rows" % (it, n)] = v[::step].copy()
            out["it%d_%s_colsum64" % (it, n)] = v.astype(np.float64).sum(axis=0)
        print("large_full it", it, flush=True)
    m = fit_ref(df, 50, 5, verbose=True, check_every=5)
    out["train_llk_it5"] = np.float64(m.train_llk)
    np.savez_compressed(os.path.join(OUT, "large_full.npz"), **out)
    print("large_full", len(out))


def c1_partial_fit():
    batches, nU, nI = datagen.partial_fit_batches()
    m = HPF(k=30, reindex=False, keep_data=False, random_seed=123, ncores=1, use_float=True)
    out = {}
    for b, (kind, bdf) in enumerate(batches):
        if b == 0:
            quiet(m.partial_fit, bdf.copy(), batch_type=kind, nusers=nU, nitems=nI)
        else:
            quiet(m.partial_fit, bdf.copy(), batch_type=kind)
        for n, v in grab(m).items():
            out["call%d_%s" % (b + 1, n)] = v
        out["call%d_niter" % (b + 1)] = np.int64(m.niter)
    np.savez_compressed(os.path.join(OUT, "c1_partial_fit.npz"), **out)
    print("c1_partial_fit", len(out))


def c1_svi():
    df, nU, nI = datagen.readme_counts()
    out = {}
    for tag, kw in (("both", dict(users_per_batch=20, items_per_batch=25)),
                    ("users", dict(users_per_batch=30)),
                    ("items", dict(items_per_batch=40)),
                    ("both_fullphi", dict(users_per_batch=20, items_per_batch=25, alloc_full_phi=True))):
        m = fit_ref(df, 30, 4, **kw)
        for n, v in grab(m).items():
            out["%s_%s" % (tag, n)] = v
    np.savez_compressed(os.path.join(OUT, "c1_svi.npz"), **out)
    print("c1_svi", len(out))


def svi_large():
    """60k x 50k, >= 2M nonzeros, k = 50, users_per_batch = items_per_batch = 8192 through the REAL extension at
    ncores=1: sub-sampled rows (every 125th) and float64 column sums of all eight arrays after 2 epochs (an item epoch
    then a user epoch, PXI:265-268) and 3 (item, user, item); users-only and items-only runs after 2 epochs.  Pins the
    stochastic path (PXI:262-377) where its per-batch whole-table float32 column sums (PXI:300,318,352,370) carry
    visible rounding, with the reference itself."""
    u, i, y, nU, nI = datagen.svi_large_counts()
    df = pd.DataFrame({"UserId": u.astype(np.int64), "ItemId": i.astype(np.int64), "Count": y})
    out = {"nnz": np.int64(df.shape[0])}
    for tag, its, kw in (("both", 2, dict(users_per_batch=8192, items_per_batch=8192)),
                         ("both", 3, dict(users_per_batch=8192, items_per_batch=8192)),
                         ("users", 2, dict(users_per_batch=8192)),
                         ("items", 2, dict(items_per_batch=8192))):
        m = fit_ref(df, 50, its, **kw)
        assert m.Theta.shape == (nU, 50) and m.Beta.shape == (nI, 50)
        for n, v in grab(m).items():
            out["%s_ep%d_%s_rows" % (tag, its, n)] = v[::125].copy()
            out["%s_ep%d_%s_colsum64" % (tag, its, n)] = v.astype(np.float64).sum(axis=0)
        print("svi_large", tag, its, flush=True)
    np.savez_compressed(os.path.join(OUT, "svi_large.npz"), **out)
    print("svi_large", len(out))


def c1_predict():
    df, nU, nI = datagen.readme_counts()
    m = fit_ref(df, 30, 20, keep_data=True)
    out = {}
    rs = np.random.RandomState(3)
    pu, pi = rs.randint(nU, size=500), rs.randint(nI, size=500)
    out["pairs_u"], out["pairs_i"] = pu, pi
    out["predict_pairs"] = m.predict(user=pu, item=pi)
    out["predict_scalar_10_11"] = np.float32(m.predict(user=10, item=11))
    for u in (0, 10, 57):
        out["topN_u%d_seen_excluded" % u] = np.array(m.topN(user=u, n=10, exclude_seen=True))
        out["topN_u%d_all" % u] = np.array(m.topN(user=u, n=10, exclude_seen=False))
    out["topN_u10_pool"] = np.array(m.topN(user=10, n=3, exclude_seen=False, items_pool=np.arange(5, 40)))
    # fold-in for a new user (README.md:131-139)
    rs2 = np.random.RandomState(2)
    new = pd.DataFrame({"ItemId": rs2.choice(np.arange(nI), size=20, replace=False),
                        "Count": rs2.gamma(1, 1, size=20).astype("int32")})
    new = new.loc[new.Count > 0].reset_index(drop=True)
    out["new_user_items"] = new.ItemId.to_numpy()
    out["new_user_counts"] = new.Count.to_numpy()
    out["predict_factors"] = quiet(m.predict_factors, new.copy(), random_seed=1)
    np.savez_compressed(os.path.join(OUT, "c1_predict.npz"), **out)
    print("c1_predict", len(out))


def c1_boundary():
    """The module-level helpers the reference exports "for ctpfrec" (PXI:20-113), called directly on the extension
    module: train / validation llk and RMSE of assess_convergence (PXI:66-79: errs[0], errs[1]), the val-set
    expression of eval_after_term (PXI:105), get_csc_data and get_unique_items_batch."""
    from hpfrec import cython_loops_float as c
    df, nU, nI = datagen.readme_counts()
    Y, iu, ii = datagen.triplets(df)
    m = fit_ref(df, 30, 10)
    Theta, Beta = np.array(m.Theta), np.array(m.Beta)
    Yv, iuv, iiv = datagen.boundary_valset(nU, nI)
    out = {}
    k = 30
    for tag, full in (("", 0), ("_full", 1)):
        for has_val in (0, 1):
            errs = np.zeros(2, dtype=c.obj_long_double_type)
            conv, crit = quiet(c.assess_convergence, 9, 10, "train-llk", -1e300, 1e-3, Theta, Theta.copy(), Beta,
                               Y.shape[0], Y, iu, ii, Yv.shape[0], Yv, iuv, iiv, errs, k, 1, 1, full, has_val)
            which = "val" if has_val else "train"
            out["assess_%s_llk%s" % (which, tag)] = np.float64(errs[0])
            out["assess_%s_rmse%s" % (which, tag)] = np.float64(errs[1])
            assert not conv and float(crit) == float(errs[0])       # first check only records
            errs2 = np.zeros(2, dtype=c.obj_long_double_type)
            last = quiet(c.eval_after_term, "maxiter", 1, 1, full, k, Y.shape[0], Yv.shape[0], has_val, Theta, Beta,
                         errs2, Y, iu, ii, Yv, iuv, iiv)
            out["after_term_%s_llk%s" % (which, tag)] = np.float64(last)
            out["after_term_%s_rmse%s" % (which, tag)] = np.float64(errs2[1])
    # second check: the stopping rule fires when 1 - llk/last_crit <= stop_thr
    errs = np.zeros(2, dtype=c.obj_long_double_type)
    conv, _ = quiet(c.assess_convergence, 19, 10, "train-llk", out["assess_train_llk"] * (1 - 5e-4), 1e-3, Theta,
                    Theta.copy(), Beta, Y.shape[0], Y, iu, ii, Yv.shape[0], Yv, iuv, iiv, errs, k, 1, 0, 0, 0)
    out["assess_second_check_converged"] = np.int64(bool(conv))
    # diff-norm branch
    Tp = (Theta * np.float32(1.01)).astype(np.float32)
    conv, crit = quiet(c.assess_convergence, 9, 10, "diff-norm", -1e300, 1e-9, Theta, Tp, Beta, Y.shape[0], Y, iu, ii,
                       0, Yv, iuv, iiv, np.zeros(2, dtype=c.obj_long_double_type), k, 1, 0, 0, 0)
    out["assess_diffnorm"] = np.float64(crit)
    assert np.array_equal(Tp, Theta)                                # Theta_prev is overwritten when not converged
    # fits that use the validation set
    vdf = pd.DataFrame({"UserId": iuv.astype(np.int64), "ItemId": iiv.astype(np.int64), "Count": Yv})
    mv = HPF(k=30, maxiter=200, random_seed=123, ncores=1, reindex=False, verbose=False, stop_crit="val-llk",
             check_every=5, stop_thr=1e-3, use_float=True)
    quiet(mv.fit, df.copy(), val_set=vdf.copy())
    out["valllk_stop_niter"] = np.int64(mv.niter)
    mv = HPF(k=30, maxiter=10, random_seed=123, ncores=1, reindex=False, verbose=True, stop_crit="maxiter",
             check_every=10, use_float=True)
    quiet(mv.fit, df.copy(), val_set=vdf.copy())
    out["maxiter_valset_last_llk"] = np.float64(mv.train_llk)      # eval_after_term's val expression (PXI:105)
    # CSC conversion with duplicate pairs (scipy merges them) and the batch helper
    rs = np.random.RandomState(1)
    du, di = rs.randint(nU, size=3000).astype(np.uint64), rs.randint(40, size=3000).astype(np.uint64)
    dy = (rs.gamma(1, 1, size=3000) + 1).astype(np.float32)
    ptr, ind, dat = c.get_csc_data(du, di, dy, nU, nI)
    out["csc_indptr"], out["csc_indices"], out["csc_data"] = ptr, ind, dat
    Ys, ius, iis, st = datagen.sorted_by_user(Y, iu, ii, nU, nI)
    users_b = np.array([5, 17, 3, 99, 42], dtype=np.uint64)
    items, st_pos = c.get_unique_items_batch(users_b, st, iis, 1, True)
    out["batch_items"], out["batch_st_pos"] = items, st_pos
    out["batch_items_only"] = c.get_unique_items_batch(users_b, st, iis, 1, False)
    np.savez_compressed(os.path.join(OUT, "c1_boundary.npz"), **out)
    print("c1_boundary", len(out))


if __name__ == "__main__":
    if len(sys.argv) > 1:                   # regenerate selected fixtures only: make_golden.py c1_boundary ...
        for name in sys.argv[1:]:
            globals()[name]()
        sys.exit(0)
    c1_boundary()
    c1_full()
    c1_trick()
    c1_hyper()
    mid_full()
    large_full()
    c1_partial_fit()
    c1_svi()
    svi_large()
    c1_predict()
