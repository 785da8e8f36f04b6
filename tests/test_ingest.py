"""hpfrec_amd.ingest: the Count filter, pd.factorize's first-appearance numbering and
the seen-items CSR of the reference (INIT:462-479, 587-606) on torch tensors, against pandas / scipy -- small cases
on CPU tensors, 24M triplets on the GPU."""
import warnings

import numpy as np
import pandas as pd
import pytest
import torch
from scipy.sparse import coo_array

import datagen
from hpfrec_amd import HPF, ingest


def _check_factorize(ids, device):
    codes, uniq = ingest.factorize(ingest.to_device_ids(ids, device))
    pc, pu = pd.factorize(ids)
    assert np.array_equal(codes.cpu().numpy(), pc)
    assert np.array_equal(uniq.cpu().numpy().astype(ids.dtype), pu)


def _check_seen(iu, ii, nU, nI, device):
    n, ptr, seen = ingest.seen_metadata(torch.from_numpy(iu.astype(np.int64)).to(device),
                                        torch.from_numpy(ii.astype(np.int64)).to(device), nU, nI)
    X = coo_array((np.ones(iu.shape[0], np.float32), (iu.astype(np.uint64), ii.astype(np.uint64))), shape=(nU, nI)).tocsr()
    assert X.indices.dtype == ingest.SEEN_INDEX_DTYPE and X.indptr.dtype == ingest.SEEN_INDEX_DTYPE
    assert np.array_equal(ptr.cpu().numpy(), X.indptr) and np.array_equal(seen.cpu().numpy(), X.indices)
    assert np.array_equal(n.cpu().numpy(), X.indptr[1:] - X.indptr[:-1])


def test_factorize_and_seen_small():
    rs = np.random.RandomState(0)
    for dtype in (np.int64, np.int32, np.uint64, np.float64):
        _check_factorize((rs.randint(0, 5000, size=100_000) * 7).astype(dtype), "cpu")
    _check_factorize(np.array([5], dtype=np.int64), "cpu")
    _check_factorize(np.empty(0, dtype=np.int64), "cpu")
    assert ingest.to_device_ids(np.array(["a", "b"], dtype=object), "cpu") is None
    assert ingest.to_device_ids(np.array([1.0, np.nan]), "cpu") is None
    _check_seen(rs.randint(0, 3000, size=100_000), rs.randint(0, 500, size=100_000), 3000, 500, "cpu")   # with duplicates
    _check_seen(np.array([2]), np.array([1]), 5, 4, "cpu")                                                  # empty rows


def _fit_pair(backend_name):
    """The same data fitted with integer ids (device ingest) and with string ids (pandas fallback): identical
    numbering, mappings (up to the id type), seen-items index and model."""
    df, nU, nI = datagen.readme_counts()
    df = df.sample(frac=1.0, random_state=3).reset_index(drop=True)      # first-appearance order != sorted order
    df["UserId"] = df["UserId"] * 3 + 11
    df.loc[::50, "Count"] = 0                                            # the Count filter has something to drop
    ds = df.copy()
    ds["UserId"] = "u" + ds["UserId"].astype(str)
    ds["ItemId"] = "i" + ds["ItemId"].astype(str)
    out = []
    for d in (df, ds):
        m = HPF(k=10, maxiter=3, random_seed=1, verbose=False, check_every=None)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            m.fit(d.copy())
        assert any("less than 1" in str(x.message) for x in w)
        out.append(m)
    a, b = out
    # with numeric ids the seen-items list is produced on the device and stays there until it is read: topN takes the
    # user's row and her seen items on the device, and answers the same before and after the list came down
    assert HPF.seen.device_of(a) is not None and HPF.seen.device_of(b) is None
    users5 = [a.user_mapping_[j] for j in (0, 3, 7, 11, 19)]
    on_device = [list(a.topN(user=u, n=6)) for u in users5]
    assert a._state.stats["d2h_bytes"] == 0                     # (neither Theta nor Beta came down for those queries)
    # once the list has been handed out the caller may edit it in place: the device copy is not trusted any more
    assert a.seen.shape[0] == b.seen.shape[0] and HPF.seen.device_of(a) is None
    a.seen = a.seen.copy()                                      # an assigned list is the host's too
    assert HPF.seen.device_of(a) is None
    assert [list(a.topN(user=u, n=6)) for u in users5] == on_device
    assert a.nusers == b.nusers and a.nitems == b.nitems
    # user_dict_ / item_dict_: the reference's plain dicts.  With numeric ids they answer single lookups from the sorted
    # renumbering and are only filled when something needs the whole table
    ud = a.user_dict_
    assert isinstance(ud, dict) and dict.__len__(ud) == 0
    assert all(ud[u] == j and u in ud for j, u in enumerate(a.user_mapping_[:50]))
    assert ud.get(-5) is None and -5 not in ud and dict.__len__(ud) == 0
    with pytest.raises(KeyError):
        ud[10 ** 9]
    assert ud == {u: j for j, u in enumerate(a.user_mapping_)} and len(ud) == a.nusers         # (filled now)
    assert a.item_dict_ == {i: j for j, i in enumerate(a.item_mapping_)}
    assert b.user_dict_ == {u: j for j, u in enumerate(b.user_mapping_)} and dict.__len__(b.item_dict_) == b.nitems
    assert np.array_equal(np.array(["u%d" % x for x in a.user_mapping_]), b.user_mapping_.astype(str))
    assert np.array_equal(np.array(["i%d" % x for x in a.item_mapping_]), b.item_mapping_.astype(str))
    assert a.user_mapping_.dtype == df["UserId"].dtype
    for name in ("seen", "_n_seen_by_user", "_st_ix_user"):
        assert np.array_equal(getattr(a, name), getattr(b, name)), name
        assert getattr(a, name).dtype == getattr(b, name).dtype, name
    assert np.array_equal(a.Theta, b.Theta) and np.array_equal(a.Beta, b.Beta)
    u0 = a.user_mapping_[0]
    assert list(a.topN(user=u0, n=5)) == [int(str(x)[1:]) for x in b.topN(user="u%d" % u0, n=5)]
    # stochastic mode: the user index (CSR starts, INIT:598) without the host sort
    for d in (df, ds):
        m = HPF(k=10, maxiter=2, random_seed=1, verbose=False, check_every=None, users_per_batch=20).fit(d.copy())
        out.append(m)
    assert np.array_equal(out[2]._st_ix_user, out[3]._st_ix_user) and out[2]._st_ix_user.dtype == out[3]._st_ix_user.dtype
    assert np.allclose(out[2].Theta, out[3].Theta, rtol=1e-5)


def test_class_device_ingest_equals_pandas_path_standin(cpu_ops_backend):
    _fit_pair("standin")


@pytest.mark.gpu
def test_class_device_ingest_equals_pandas_path_gpu(hip_backend):
    _fit_pair("hip")


@pytest.mark.gpu
def test_ingest_at_24m_triplets_vs_pandas_scipy():
    """24M triplets with raw (non-contiguous, unsorted) ids and repeated pairs: numbering and mappings equal
    pd.factorize, the seen-items CSR equals scipy's coo -> csr, the Count filter equals the boolean mask."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    rs = np.random.RandomState(5)
    n = 24_000_000
    raw_u = (rs.randint(0, 600_000, size=n).astype(np.int64) * 13 + 7)
    raw_i = np.minimum((250_000 * rs.random_sample(n) ** 2.5).astype(np.int64), 249_999) * 3 + 1
    cnt = rs.poisson(1.2, size=n).astype(np.int32)
    dev = torch.device("cuda", 0)
    du, di = ingest.to_device_ids(raw_u, dev), ingest.to_device_ids(raw_i, dev)
    keep = torch.from_numpy(cnt).to(dev) > 0.9
    assert np.array_equal(keep.cpu().numpy(), ~(cnt <= 0.9))
    du, di = du[keep], di[keep]
    hu, hi = raw_u[cnt > 0.9], raw_i[cnt > 0.9]
    cu, mu = ingest.factorize(du)
    ci, mi = ingest.factorize(di)
    pcu, pmu = pd.factorize(hu)
    pci, pmi = pd.factorize(hi)
    assert np.array_equal(cu.cpu().numpy(), pcu) and np.array_equal(mu.cpu().numpy(), pmu)
    assert np.array_equal(ci.cpu().numpy(), pci) and np.array_equal(mi.cpu().numpy(), pmi)
    nU, nI = pmu.shape[0], pmi.shape[0]
    nseen, ptr, seen = ingest.seen_metadata(cu, ci, nU, nI)
    X = coo_array((np.ones(pcu.shape[0], np.float32), (pcu.astype(np.uint64), pci.astype(np.uint64))), shape=(nU, nI)).tocsr()
    assert np.array_equal(ptr.cpu().numpy(), X.indptr) and np.array_equal(seen.cpu().numpy(), X.indices)
    assert X.indices.dtype == ingest.SEEN_INDEX_DTYPE
    assert int(nseen.sum()) == X.nnz < pcu.shape[0]            # repeated pairs were merged


def test_id_lookup_equals_categorical_codes():
    rs = np.random.RandomState(0)
    for dt in (np.int64, np.int32, np.float64, np.uint64):
        mapping = pd.unique((rs.randint(0, 10 ** 6, size=50_000) * 3).astype(dt))
        q = (rs.randint(0, 10 ** 6, size=120_000) * 3 + rs.randint(0, 2, size=120_000)).astype(dt)
        got = ingest.IdLookup(mapping, "cpu")(q).numpy()
        assert np.array_equal(got, pd.Categorical(q, mapping).codes), dt
    assert np.array_equal(ingest.IdLookup(np.empty(0, np.int64), "cpu")(np.array([1, 2])).numpy(), [-1, -1])


def _big_valset_and_predict():
    """A validation set / predict query large enough for the device lookup (>= 200k ids): same codes, same results
    as the pandas path (forced by lowering / raising the threshold)."""
    import hpfrec_amd
    rs = np.random.RandomState(4)
    n = 260_000
    df = pd.DataFrame({"UserId": rs.randint(0, 3000, size=n) * 7 + 1, "ItemId": rs.randint(0, 800, size=n) * 3 + 2,
                       "Count": (rs.gamma(1, 1, size=n) + 1).astype("int32")})
    df = df.loc[~df[["UserId", "ItemId"]].duplicated()].reset_index(drop=True)
    val = df.sample(n=210_000, random_state=1).reset_index(drop=True)
    val.loc[::1000, "UserId"] = 10 ** 7                      # ids the model never saw
    res = {}
    for thr in (200_000, 10 ** 9):
        old = hpfrec_amd._BIG_LOOKUP
        hpfrec_amd._BIG_LOOKUP = thr
        try:
            m = HPF(k=8, maxiter=4, random_seed=1, verbose=False, stop_crit="val-llk", check_every=2).fit(df.copy(), val.copy())
            p = m.predict(val["UserId"].to_numpy(), val["ItemId"].to_numpy())
            res[thr] = (m.niter, m.Theta.copy(), p, m.eval_llk(val.copy())["llk"])
            assert ("_id_lookup" in m.__dict__) == (thr == 200_000)
        finally:
            hpfrec_amd._BIG_LOOKUP = old
    a, b = res[200_000], res[10 ** 9]
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(np.isnan(a[2]), np.isnan(b[2]))
    assert np.isnan(a[2]).sum() == 210 and np.allclose(a[2][~np.isnan(a[2])], b[2][~np.isnan(b[2])], rtol=1e-6)
    assert abs(float(a[3]) / float(b[3]) - 1) < 1e-9


def test_big_valset_lookup_standin(cpu_ops_backend):
    _big_valset_and_predict()


@pytest.mark.gpu
def test_big_valset_lookup_gpu(hip_backend):
    _big_valset_and_predict()
