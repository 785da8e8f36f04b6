"""The drop-in HPF class: README sample flow (/root/reference/README.md:72-150) on the CPU stand-in
ops (host logic only), and on the GPU against the reference's golden outputs."""
import os
import warnings

import numpy as np
import pandas as pd
import pytest
import torch
from scipy.sparse import coo_array

import datagen
from conftest import GOLDEN
from hpfrec_amd import HPF
from test_host_logic import _fit

NAMES = ("Theta", "Beta", "Gamma_shp", "Gamma_rte", "Lambda_shp", "Lambda_rte", "k_rte", "t_rte")


def _maxrel(a, b):
    return float(np.max(np.abs(a - b) / np.abs(b)))


def _check_fit_against_golden(tol):
    df, nU, nI = datagen.readme_counts()
    g = np.load(os.path.join(GOLDEN, "c1_full.npz"))
    m = HPF(k=30, maxiter=10, random_seed=123, reindex=False, verbose=False, check_every=None, keep_all_objs=True)
    assert m.fit(df.copy()) is m
    assert m.is_fitted and m.niter == 9 and m.train_llk is None
    for n in NAMES:
        assert _maxrel(getattr(m, n), g["it10_%s" % n]) < tol, n
    return m, df, nU, nI, g


def test_constructor_validation():
    with pytest.raises(AssertionError):
        HPF(k=0)
    with pytest.raises(AssertionError):
        HPF(a=-1.0)
    with pytest.raises(ValueError):
        HPF(maxiter=None)
    with pytest.raises(ValueError):
        HPF(stop_crit="train-llk", check_every=None)
    with pytest.raises(ValueError):
        HPF(step_size=3)
    with pytest.raises(AssertionError):
        HPF(stop_crit="bogus")
    m = HPF(a=1, verbose=False)
    assert m.a == 1.0 and m.check_every == 0 and m.users_per_batch == 0
    assert HPF(reindex=False).produce_dicts is False


def test_fit_flow(any_backend, capsys):
    m, df, nU, nI, g = _check_fit_against_golden(5e-5)
    assert m.Theta.shape == (nU, 30) and m.Beta.shape == (nI, 30) and m.Theta.dtype == np.float32
    assert not hasattr(m, "input_df") and not hasattr(m, "val_set")
    # reindex=True: ids renumbered by first appearance; dicts built; predictions go through the mapping
    df2 = df.copy()
    df2["UserId"] = "u" + df2["UserId"].astype(str)
    df2["ItemId"] = df2["ItemId"] + 1000
    m2 = HPF(k=30, maxiter=10, random_seed=123, verbose=True, stop_crit="train-llk", check_every=5).fit(df2)
    out = capsys.readouterr().out
    assert "Hierarchical Poisson Factorization" in out and "Number of users: 100" in out and "train llk" in out
    first_u = pd.unique(df2["UserId"])
    assert list(m2.user_mapping_[:5]) == list(first_u[:5]) and m2.user_dict_[first_u[3]] == 3
    p = m2.predict(user=first_u[3], item=m2.item_mapping_[7])
    assert np.isclose(p, m2.Theta[3].dot(m2.Beta[7]), rtol=1e-6)
    assert np.isnan(m2.predict(user="nobody", item=m2.item_mapping_[7]))
    arr = m2.predict(user=[first_u[0], "nobody", first_u[2]], item=[1001, 1002, 999999])
    assert np.isfinite(arr[0]) and np.isnan(arr[1]) and np.isnan(arr[2])
    top = m2.topN(user=first_u[0], n=10, exclude_seen=True)
    seen = set(df2.loc[df2.UserId == first_u[0], "ItemId"])
    assert len(top) == 10 and not (set(top) & seen)
    top_all = m2.topN(user=first_u[0], n=5, exclude_seen=False)
    scores = m2.Theta[0].dot(m2.Beta.T)
    assert list(top_all) == list(m2.item_mapping_[np.argsort(-scores)[:5]])
    with pytest.raises(ValueError):
        m2.topN(user="nobody")
    ll = m2.eval_llk(df2.copy())
    assert ll["nobs"] == df2.shape[0] and np.isfinite(float(ll["llk"]))


def test_fit_inputs_coo_array_and_zero_filter(any_backend):
    df, nU, nI = datagen.readme_counts()
    m_df = HPF(k=8, maxiter=3, random_seed=1, reindex=False, verbose=False, check_every=None).fit(df.copy())
    X = coo_array((df.Count.to_numpy(), (df.UserId.to_numpy(), df.ItemId.to_numpy())), shape=(nU, nI))
    m_coo = HPF(k=8, maxiter=3, random_seed=1, verbose=False, check_every=None).fit(X)
    assert m_coo.reindex is False and np.array_equal(m_df.Theta, m_coo.Theta)
    m_arr = HPF(k=8, maxiter=3, random_seed=1, reindex=False, verbose=False, check_every=None).fit(df[["UserId", "ItemId", "Count"]].to_numpy())
    assert np.allclose(m_df.Theta, m_arr.Theta, rtol=1e-6)
    dfz = pd.concat([df, pd.DataFrame({"UserId": [1], "ItemId": [2], "Count": [0]})], ignore_index=True)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        m_z = HPF(k=8, maxiter=3, random_seed=1, reindex=False, verbose=False, check_every=None).fit(dfz)
    assert any("less than 1" in str(x.message) for x in w)
    assert np.array_equal(m_z.Theta, m_df.Theta)
    with pytest.raises(ValueError):
        HPF(verbose=False).fit([1, 2, 3])
    with pytest.raises(ValueError):
        HPF(stop_crit="val-llk", verbose=False).fit(df.copy())


def test_valset_stopping(any_backend):
    df, nU, nI = datagen.readme_counts()
    val = df.sample(200, random_state=1)
    m = HPF(k=10, maxiter=60, stop_crit="val-llk", check_every=5, stop_thr=1e-2, random_seed=2, verbose=False,
            reindex=False).fit(df.copy(), val_set=val.copy())
    assert m.niter < 59 and (m.niter + 1) % 5 == 0


@pytest.mark.gpu
def test_fit_predict_topn_vs_golden_on_gpu(hip_backend):
    m, df, nU, nI, g = _check_fit_against_golden(5e-5)
    gp = np.load(os.path.join(GOLDEN, "c1_predict.npz"))
    m20 = HPF(k=30, maxiter=20, random_seed=123, reindex=False, verbose=False, check_every=None).fit(df.copy())
    got = m20.predict(user=gp["pairs_u"], item=gp["pairs_i"])
    assert _maxrel(got, gp["predict_pairs"]) < 1e-4
    assert abs(m20.predict(user=10, item=11) / gp["predict_scalar_10_11"] - 1) < 1e-4
    for u in (0, 10, 57):
        # same ids; order may differ only where the reference's own scores tie within rounding
        assert set(m20.topN(user=u, n=10, exclude_seen=True)) == set(gp["topN_u%d_seen_excluded" % u])
        assert list(m20.topN(user=u, n=10, exclude_seen=False))[:3] == list(gp["topN_u%d_all" % u])[:3]
    assert list(m20.topN(user=10, n=3, exclude_seen=False, items_pool=np.arange(5, 40))) == list(gp["topN_u10_pool"])
    gl = np.load(os.path.join(GOLDEN, "c1_full.npz"))
    assert abs(float(m20.eval_llk(df.copy())["llk"]) / gl["eval_llk_it20"] - 1) < 1e-4
    mv = HPF(k=30, maxiter=10, random_seed=123, reindex=False, verbose=True, check_every=10).fit(df.copy())
    assert abs(float(mv.train_llk) / gl["train_llk_it10"] - 1) < 1e-4


def test_save_folder_and_flags(any_backend, tmp_path):
    df, nU, nI = datagen.readme_counts()
    df2 = df.copy()
    df2["UserId"] = df2["UserId"] + 500
    m = HPF(k=6, maxiter=3, random_seed=3, verbose=False, check_every=None, save_folder=str(tmp_path),
            keep_all_objs=False, produce_dicts=False, full_llk=True).fit(df2)
    files = set(os.listdir(str(tmp_path)))
    # the reference writes the eight arrays under extension-less names (cython_loops.pxi:410)
    assert {"Theta", "Beta", "Gamma_shp", "Gamma_rte", "Lambda_shp", "Lambda_rte", "kappa_rte", "tau_rte",
            "users.csv", "items.csv", "hyperparameters.txt"} <= files
    saved = np.loadtxt(os.path.join(str(tmp_path), "Theta"), delimiter=",")
    assert saved.shape == m.Theta.shape and np.allclose(saved, m.Theta, atol=1e-6)
    assert "k: 6" in open(os.path.join(str(tmp_path), "hyperparameters.txt")).read()
    assert not hasattr(m, "Gamma_shp") and m.user_dict_ is None          # keep_all_objs / produce_dicts off
    assert np.isfinite(m.predict(user=int(m.user_mapping_[0]), item=int(m.item_mapping_[0])))
    with pytest.raises(AssertionError):
        m.predict_factors(df[["ItemId", "Count"]].head(5))                 # needs keep_all_objs
    # diff-norm criterion stops, llk criteria reject counts < 1 rows with a warning
    g = np.load(os.path.join(GOLDEN, "c1_full.npz"))
    m3 = HPF(k=30, maxiter=200, stop_crit="diff-norm", check_every=5, stop_thr=1e-1, random_seed=123, verbose=False,
             reindex=False).fit(df.copy())
    assert m3.niter == int(g["diffnorm_stop_niter"])


def test_use_float_false_is_refused():
    df, nU, nI = datagen.readme_counts()
    with pytest.raises(NotImplementedError):
        HPF(use_float=False, verbose=False).fit(df.copy())


def test_state_stays_on_the_device_and_never_goes_stale(any_backend):
    """hpfrec_amd.resident: between calls the state lives on the device; reading / editing / assigning the public
    attributes is always reflected by the next device operation (no content fingerprints), and a stream of
    partial_fit calls moves only the batches over PCIe."""
    import copy
    import pickle
    batches, nU, nI = datagen.partial_fit_batches()
    k = 12
    m = HPF(k=k, reindex=False, keep_data=False, random_seed=5, verbose=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m.partial_fit(batches[0][1].copy(), nusers=nU, nitems=nI)
        st = m._state
        up0, down0 = st.stats["h2d_bytes"], st.stats["d2h_bytes"]
        assert up0 > 0 and down0 == 0                      # first call: everything went up, nothing came back
        for kind, bdf in batches[1:]:
            m.partial_fit(bdf.copy(), batch_type=kind)
    assert st.stats["h2d_bytes"] == up0 and st.stats["d2h_bytes"] == 0     # three more calls: no table crossed PCIe
    assert not st.host_ok["Theta"] and st.on_device("Theta")
    # the same sequence through the extension-level entry point (full upload/download per call) gives the same state
    m2 = HPF(k=k, reindex=False, keep_data=False, random_seed=5, verbose=False)
    be = m2._backend()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m2.partial_fit(batches[0][1].iloc[:0:1].copy() if False else batches[0][1].copy(), nusers=nU, nitems=nI)
    ref = {n: getattr(m2, n).copy() for n in NAMES}
    niter = 1
    for kind, bdf in batches[1:]:
        Y, iu, ii = datagen.triplets(bdf)
        users, items = np.unique(iu), np.unique(ii)
        be.partial_fit(Y, iu, ii, ref["Theta"], ref["Beta"], ref["Gamma_shp"], ref["Gamma_rte"], ref["Lambda_shp"],
                       ref["Lambda_rte"], ref["k_rte"], ref["t_rte"], be.cast_real_t(0.3 / 1.0), be.cast_real_t(0.3 / 1.0),
                       0.3, 0.3, be.cast_real_t(0.3 + k * 0.3), be.cast_real_t(0.3 + k * 0.3), k, users, items, 0,
                       be.cast_real_t(1 / np.sqrt(niter + 2)), be.cast_real_t(float(nU) / users.shape[0]), 1,
                       kind == "users")
        niter += 1
    for n in NAMES:                                              # reading the attributes downloads the tables ...
        # (to rounding: the resident model carries its column sums from step to step, a fresh upload recomputes them
        # with another launch geometry)
        assert _maxrel(getattr(m, n), ref[n]) < 1e-5, n
    assert st.stats["d2h_bytes"] > 0 and not st.on_device("Beta")   # ... and hands them out: device copies untrusted
    # in-place edits of a few rows at ANY later time (a sampled fingerprint would not notice them) reach topN
    B = m.Beta
    top = m.topN(user=3, n=5, exclude_seen=False)
    loser = int(np.setdiff1d(np.arange(nI), top)[0])
    B[loser] = 1e3
    assert m.topN(user=3, n=5, exclude_seen=False)[0] == loser
    B[loser] = 1e-9
    assert loser not in m.topN(user=3, n=5, exclude_seen=False)
    # a fit ENDS on the device: the tables it leaves there serve topN / fold-in / partial_fit without crossing PCIe
    # in either direction, until somebody reads an attribute
    m4 = HPF(k=k, reindex=False, keep_data=False, random_seed=5, verbose=False, maxiter=3, check_every=None).fit(batches[0][1].copy())
    st4 = m4._state
    assert st4.stats == {"h2d_bytes": 0, "d2h_bytes": 0} and st4.on_device("Beta") and not st4.host_ok["Theta"]
    r1 = m4.topN(user=3, n=5, exclude_seen=False)
    assert np.array_equal(m4.topN(user=3, n=5, exclude_seen=False), r1)
    f1 = m4.predict_factors(batches[1][1][["ItemId", "Count"]].iloc[:7].copy())
    assert st4.stats == {"h2d_bytes": 0, "d2h_bytes": 0}
    # ... and what comes down on request is what a fit through the extension-level entry point (host arrays out) gives
    Y4, iu4, ii4 = datagen.triplets(batches[0][1])
    _, ref4, _ = _fit(m4._backend(), Y4, iu4, ii4, int(m4.nusers), int(m4.nitems), k, 3, seed=5)
    for n in NAMES:
        assert np.array_equal(getattr(m4, n), ref4[n]), n
    assert st4.stats["d2h_bytes"] > 0 and st4.stats["h2d_bytes"] == 0
    assert np.array_equal(m4.topN(user=3, n=5, exclude_seen=False), r1)       # (Beta handed out: re-uploaded, same answer)
    assert np.allclose(m4.predict_factors(batches[1][1][["ItemId", "Count"]].iloc[:7].copy()), f1, rtol=1e-6)
    # re-assignment, predict and eval_llk see the current arrays too
    newB = np.ascontiguousarray(m.Beta[::-1])
    m.Beta = newB
    want = (m.Theta[[1, 2, 3]] * newB[[5, 6, 7]]).sum(axis=1)
    assert np.allclose(m.predict(np.array([1, 2, 3]), np.array([5, 6, 7])), want, rtol=1e-5)
    assert abs(m.predict(1, 5) / want[0] - 1) < 1e-5
    pairs = batches[0][1]
    a = m.eval_llk(pairs.copy())["llk"]
    m.Beta = newB * np.float32(2.0)
    assert abs(float(a - m.eval_llk(pairs.copy())["llk"])) > 1e-3
    # a later partial_fit starts from the edited arrays
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m.partial_fit(batches[1][1].copy())
    assert np.isfinite(m.Theta).all() and m.Beta[loser if False else 0].shape == (k,)
    # pickling / deep copies carry the host arrays
    m3 = pickle.loads(pickle.dumps(copy.deepcopy(m._state)))
    for n in NAMES:
        assert np.array_equal(m3.host[n], getattr(m, n)), n
    # attributes that were never assigned behave like missing attributes
    assert not hasattr(HPF(verbose=False), "Gamma_shp") and HPF(verbose=False).Theta is None


def test_refit_and_mixed_calls(any_backend, tmp_path, monkeypatch):
    """fit -> topN -> partial_fit -> predict -> fit on a smaller problem (tables rebuilt) -> topN / eval_llk: the
    resident state follows every change of shape and owner.  (A first fit leaves save_folder = "" behind, like the
    reference's -- INIT:632-633 -- so a REFIT writes hyperparameters.txt into the working directory, INIT:494-495:
    the test runs in a scratch directory.)"""
    monkeypatch.chdir(tmp_path)
    df, nU, nI = datagen.readme_counts()
    m = HPF(k=6, maxiter=3, reindex=False, verbose=False, check_every=None, random_seed=1)
    m.fit(df.copy())
    assert len(m.topN(user=3, n=5)) == 5
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m.partial_fit(df.loc[df.UserId.isin([1, 2, 3])].copy())
    assert m.niter == 3 and not m._state.host_ok["Beta"]
    p = m.predict(np.array([1, 2]), np.array([3, 4]))
    assert np.allclose(p, (m.Theta[[1, 2]] * m.Beta[[3, 4]]).sum(axis=1), rtol=1e-5)
    df2 = df.loc[(df.UserId < 40) & (df.ItemId < 50)].copy()
    m.fit(df2.copy())
    assert m.Theta.shape == (int(df2.UserId.max()) + 1, 6)
    assert len(m.topN(user=3, n=5)) == 5 and m._state.model.nU == m.Theta.shape[0]
    assert np.isfinite(float(m.eval_llk(df2.copy())["llk"]))
    assert os.path.exists(os.path.join(str(tmp_path), "hyperparameters.txt"))       # (the reference's quirk, kept)


def test_partial_fit_continues_from_the_tables_a_fit_left_on_the_device(any_backend):
    """fit() hands its device tables to the resident state (no download); a partial_fit / add_user right after it must
    see exactly the state a host round trip of all eight arrays would give it (pad columns, scratch tables and column
    sums of the adopted tables included)."""
    batches, nU, nI = datagen.partial_fit_batches()
    k = 12
    full = pd.concat([b for _, b in batches[:2]]).drop_duplicates(["UserId", "ItemId"])

    def fitted():
        m = HPF(k=k, reindex=False, keep_data=False, random_seed=9, verbose=False, maxiter=4, check_every=None)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m.fit(full.copy())
        return m

    a, b = fitted(), fitted()
    assert b._state.stats["d2h_bytes"] == 0
    for n in NAMES:                        # b: every array is read (downloaded and handed out: re-uploaded on next use)
        getattr(b, n)
    assert b._state.stats["d2h_bytes"] > 0 and a._state.stats["d2h_bytes"] == 0
    nxt = batches[2][1]
    nxt = nxt[(nxt["UserId"] < a.nusers) & (nxt["ItemId"] < a.nitems)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for m in (a, b):
            m.partial_fit(nxt.copy(), batch_type=batches[2][0])
    for n in NAMES:
        assert _maxrel(getattr(a, n), getattr(b, n)) < 2e-6, n
    new = nxt[["ItemId", "Count"]].iloc[:9]
    assert np.allclose(a.predict_factors(new.copy()), b.predict_factors(new.copy()), rtol=1e-5)


def test_pickle_carries_host_data_only_and_seen_edits_are_honoured(any_backend):
    """a fitted model pickles without device tensors (seen list, id lookups), the copy predicts the same;
    once `model.seen` has been handed out, topN(exclude_seen) follows in-place edits of it; the lazily filled id tables
    behave like the plain dicts of the reference for setdefault / clear / update."""
    import pickle
    df, nU, nI = datagen.readme_counts()
    m = HPF(k=8, maxiter=5, verbose=False, random_seed=1, check_every=None, stop_crit="maxiter")
    m.fit(df.copy())
    blob = pickle.dumps(m)
    import torch
    assert b"torch" not in blob[:200] and not any(torch.is_tensor(v) for v in m.__getstate__().values())
    m2 = pickle.loads(blob)
    u = int(df.UserId.iloc[0])
    assert np.array_equal(m.topN(u, n=5), m2.topN(u, n=5))
    assert np.allclose(m.predict(user=[u], item=[int(df.ItemId.iloc[0])]), m2.predict(user=[u], item=[int(df.ItemId.iloc[0])]))
    # seen handed out -> edits count
    best = m.topN(u, n=3, exclude_seen=True)
    seen = m.seen
    ui = m.user_dict_[u]
    st, n = int(m._st_ix_user[ui]), int(m._n_seen_by_user[ui])
    assert n > 0
    first_seen_item = int(seen[st])
    seen[st] = m.item_dict_[int(best[0])] if hasattr(m, "item_dict_") else int(best[0])   # pretend she saw the best one
    again = m.topN(u, n=3, exclude_seen=True)
    assert int(best[0]) not in [int(x) for x in again]
    seen[st] = first_seen_item
    # id tables: whole-table methods see the mapped ids
    d = m.user_dict_
    some = int(m.user_mapping_[3])
    assert d.setdefault(some, -5) == 3 and d[some] == 3
    d2 = type(d)(m.user_mapping_, getattr(d, "_sorted", None), getattr(d, "_codes", None))
    d2.update({some: 99})
    assert d2[some] == 99 and len(d2) == len(m.user_mapping_)
    d2.clear()
    assert len(d2) == 0 and some not in d2
