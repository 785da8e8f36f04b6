"""The N>1 path on CPU: world_size 2, 3, 4 and 8 over gloo (user-sharded driver, both exchange modes), numpy stand-in ops.  Every rank must end with
the full, identical model, equal (to rounding: the sum order changes) to the single-process run."""
import os
import socket

import numpy as np
import pytest

import datagen
import dist_worker
from conftest import spawn_ranks

NAMES = ("Theta", "Beta", "Gamma_shp", "Gamma_rte", "Lambda_shp", "Lambda_rte", "k_rte", "t_rte")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,case,mode", [(2, "c1", "scatter"), (3, "mid", "scatter"), (3, "c1", "scatter"),
                                             (8, "c1", "scatter"), (4, "mid", "scatter"), (8, "mid", "allreduce"),
                                             (3, "c1", "scatter-a2a"), (3, "c1", "scatter-one-range"),
                                             (3, "c1", "scatter-packed"),
                                             (3, "c1", "scatter-early"), (2, "mid", "scatter-early"), (8, "c1", "scatter-early"),
                                             (2, "c1", "allreduce"), (3, "mid", "allreduce")])
def test_sharded_equals_single(tmp_path, cpu_ops_backend, monkeypatch, world, case, mode):
    """mode: "scatter" = reduce-scatter / sharded item finalizer / all-gather (default); "allreduce" = all-reduce +
    replicated deferred finalizer.  (3, c1): 100 items over 3 ranks -> pad rows in the item tables; (8, c1): the
    driver's largest rank count, 12-13 users and 100 items (ranges padded to multiples of 8) per rank."""
    if mode == "scatter-a2a":                     # reduce-scatter as all-to-all + local sum
        mode = "scatter"
        monkeypatch.setenv("HPF_RS_ALLTOALL", "1")
    if mode == "scatter-packed":                  # new E rows all-gathered k-packed + unpacked (pad rows included)
        mode = "scatter"
        monkeypatch.setenv("HPF_AG_PACKED", "1")
    monkeypatch.setenv("HPF_GATHER_EARLY", "1" if mode == "scatter-early" else "0")
    if mode == "scatter-early":                   # split item finalizer: the all-gather before the user side (the default)
        mode = "scatter"
    if mode == "scatter-one-range":               # no exchange pipelining
        mode = "scatter"
        monkeypatch.setenv("HPF_AR_CHUNKS", "1")
    monkeypatch.setenv("HPF_SHARD_MODE", mode)
    k, its = 20, 5
    if case == "c1":
        df, nU, nI = datagen.readme_counts()
    else:
        df, nU, nI = datagen.mid_counts(nusers=600, nitems=400, nobs=20000)
    Y, iu, ii = datagen.triplets(df)
    Theta = np.empty((nU, k), np.float32)
    Beta = np.empty((nI, k), np.float32)
    i, temp, llk = cpu_ops_backend.fit_hpf(0.3, 0.3, 1.0, 0.3, 0.3, 1.0, Y, iu, ii, Theta, Beta, its, "maxiter", its, 1e-3,
                                           0, 0, None, 0, np.zeros(1, np.uint64), "", 123, 1, 1, 0, 0,
                                           np.empty(0, np.float32), np.empty(0, np.uint64), np.empty(0, np.uint64), 0, 1, 0)
    single = dict(zip(NAMES, (Theta, Beta) + tuple(temp)))
    spawn_ranks(dist_worker.run, lambda port: (world, port, str(tmp_path), k, its, case), world, str(tmp_path))
    outs = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    for r in range(world):
        assert int(outs[r]["niter"]) == i
        assert abs(float(outs[r]["llk"]) / float(llk) - 1) < 1e-6
        for n in NAMES:
            assert outs[r][n].shape == single[n].shape
            assert np.max(np.abs(outs[r][n] - single[n]) / np.abs(single[n])) < 1e-5, (r, n)
            assert np.array_equal(outs[r][n], outs[0][n]), (r, n)  # replicas agree bit for bit


@pytest.mark.parametrize("world,mode", [(3, "scatter"), (2, "allreduce")])
def test_unseeded_ranks_start_from_the_same_draw(tmp_path, cpu_ops_backend, monkeypatch, world, mode):
    """random_seed <= 0 means OS entropy (PXI:127): each rank would draw its own initial item tables and the replicas
    would drift apart; the sharded fit broadcasts rank 0's generator state, so all ranks still agree bit for bit."""
    monkeypatch.setenv("HPF_SHARD_MODE", mode)
    k, its = 12, 3
    spawn_ranks(dist_worker.run, lambda port: (world, port, str(tmp_path), k, its, "c1-entropy"), world, str(tmp_path))
    outs = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    for r in range(1, world):
        for n in NAMES:
            assert np.isfinite(outs[r][n]).all() and np.array_equal(outs[r][n], outs[0][n]), (r, n)
