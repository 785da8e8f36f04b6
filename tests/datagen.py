"""Deterministic input generators shared by the golden-vector script and the tests.

Inputs are regenerated from seeds wherever the tests run (the GPU box has no
/root/reference); only the reference's *outputs* are stored under tests/golden/.
Legacy np.random.RandomState streams are used because they are frozen across numpy
versions.
"""
import numpy as np
import pandas as pd


def readme_counts():
    """BASELINE config C1: the generator of the reference's README sample
    (/root/reference/README.md:77-87): 100x100, 1e4 draws, duplicates dropped -> 6,347 triplets."""
    rs = np.random.RandomState(1)
    nusers, nitems, nobs = 100, 100, 10 ** 4
    df = pd.DataFrame({
        "UserId": rs.randint(nusers, size=nobs),
        "ItemId": rs.randint(nitems, size=nobs),
        "Count": (rs.gamma(1, 1, size=nobs) + 1).astype("int32"),
    })
    df = df.loc[~df[["UserId", "ItemId"]].duplicated()].reset_index(drop=True)
    return df, nusers, nitems


def mid_counts(nusers=3000, nitems=2000, nobs=2 * 10 ** 5, seed=7, power=2.0):
    """Mid-size case with a heavy-headed item popularity (a few very long CSC rows) and
    ragged user degrees, duplicates dropped."""
    rs = np.random.RandomState(seed)
    u = (nusers * rs.random_sample(nobs) ** 1.5).astype(np.int64)
    i = (nitems * rs.random_sample(nobs) ** power).astype(np.int64)
    y = (rs.gamma(1, 1, size=nobs) + 1).astype("int32")
    df = pd.DataFrame({"UserId": u, "ItemId": i, "Count": y})
    df = df.loc[~df[["UserId", "ItemId"]].duplicated()].reset_index(drop=True)
    # make sure the highest ids exist so that reindex=False sees nusers x nitems
    tail = pd.DataFrame({"UserId": [nusers - 1], "ItemId": [nitems - 1], "Count": np.array([1], dtype="int32")})
    df = pd.concat([df, tail], ignore_index=True)
    df = df.loc[~df[["UserId", "ItemId"]].duplicated()].reset_index(drop=True)
    return df, nusers, nitems


def triplets(df):
    """(Y float32, ix_u uint64, ix_i uint64) in the data frame's row order (the reference's
    casts, /root/reference/hpfrec/__init__.py:508-514)."""
    return (df["Count"].to_numpy().astype(np.float32),
            df["UserId"].to_numpy().astype(np.uint64),
            df["ItemId"].to_numpy().astype(np.uint64))


def partial_fit_batches():
    """The README partial_fit sequence (README.md:111-117), with fixed batches, plus one item batch."""
    df, nusers, nitems = readme_counts()
    rs = np.random.RandomState(11)
    ub = [np.unique(rs.randint(nusers, size=20)) for _ in range(3)]
    ib = [np.unique(rs.randint(nitems, size=25))]
    batches = [("users", df.loc[df.UserId.isin(b)].reset_index(drop=True)) for b in ub]
    batches += [("items", df.loc[df.ItemId.isin(b)].reset_index(drop=True)) for b in ib]
    return batches, nusers, nitems


def synthetic_hpf_shaped(nusers, nitems, nnz, seed=1, item_power=2.5, sigma=1.0):
    """Numpy version of the benchmark generator: log-normal user
    degrees, power-law item popularity, unique pairs, Y = 1 + floor(Gamma(1,1)).
    Used for small/mid parity cases; bench.py has the on-device equivalent."""
    rng = np.random.default_rng(seed)
    deg = rng.lognormal(mean=0.0, sigma=sigma, size=nusers)
    deg = np.maximum(1, np.round(deg * (nnz / deg.sum()))).astype(np.int64)
    u = np.repeat(np.arange(nusers, dtype=np.int64), deg)
    i = np.minimum((nitems * rng.random(u.shape[0]) ** item_power).astype(np.int64), nitems - 1)
    key = np.unique(u * nitems + i)
    u, i = key // nitems, key % nitems
    y = (1 + np.floor(rng.gamma(1.0, 1.0, size=u.shape[0]))).astype(np.float32)
    perm = rng.permutation(u.shape[0])
    return u[perm].astype(np.uint64), i[perm].astype(np.uint64), y[perm]


def large_counts(nusers=200_000, nitems=50_000, nnz=5_400_000, seed=9):
    """The large golden case (tests/golden/large_full.npz): the benchmark's shape of problem at 200k x 50k, >= 5M
    nonzeros -- big enough that numpy's float32 column sums over the rows (cython_loops.pxi:236,255) carry visible
    rounding noise, small enough for the real reference to run it in minutes.  The pair (nusers-1, nitems-1) is
    appended so that reindex=False sees the full shape."""
    u, i, y = synthetic_hpf_shaped(nusers, nitems, nnz, seed=seed)
    if not np.any((u == nusers - 1) & (i == nitems - 1)):
        u = np.concatenate([u, np.array([nusers - 1], np.uint64)])
        i = np.concatenate([i, np.array([nitems - 1], np.uint64)])
        y = np.concatenate([y, np.array([1.0], np.float32)])
    return u, i, y, nusers, nitems


def svi_large_counts(nusers=60_000, nitems=50_000, nnz=2_400_000, seed=13):
    """The stochastic golden case above toy size (tests/golden/svi_large.npz): 60k x 50k, >= 2M nonzeros, batches of
    8192 rows -- the whole-table float32 column sums every batch takes (cython_loops.pxi:300,318,352,370) run over
    5e4..6e4 rows.  The pair (nusers-1, nitems-1) is appended so that reindex=False sees the full shape."""
    return large_counts(nusers, nitems, nnz, seed)


def boundary_valset(nusers, nitems, n=400, seed=21):
    """A small validation set for the llk / RMSE fixtures of tests/golden/c1_boundary.npz."""
    rs = np.random.RandomState(seed)
    u = rs.randint(nusers, size=n).astype(np.uint64)
    i = rs.randint(nitems, size=n).astype(np.uint64)
    y = (rs.gamma(1, 1, size=n) + 1).astype("int32").astype(np.float32)
    return y, u, i


def sorted_by_user(Y, ix_u, ix_i, nusers, nitems):
    """Triplets stably sorted by user plus the CSR start indices (what the SVI path of the reference works on)."""
    order = np.argsort(ix_u, kind="stable")
    st = np.zeros(nusers + 1, dtype=np.uint64)
    st[1:] = np.cumsum(np.bincount(ix_u.astype(np.int64), minlength=nusers))
    return Y[order], ix_u[order], ix_i[order], st
