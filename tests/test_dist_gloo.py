"""The N>1 path on CPU: world_size 2, 3, 4 and 8 over gloo (user-sharded driver, both call-by-call schedules), numpy stand-in ops.  Every rank must end with
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


@pytest.mark.parametrize("world,case,mode", [(2, "c1", "finalize-then-gather"), (3, "mid", "finalize-then-gather"),
                                             (3, "c1", "finalize-then-gather"), (8, "c1", "finalize-then-gather"),
                                             (4, "mid", "finalize-then-gather"), (3, "c1", "one-range"),
                                             (3, "c1", "gather-early"), (2, "mid", "gather-early"), (8, "c1", "gather-early"),
                                             (8, "mid", "gather-early"), (3, "mid", "auto"),
                                             (4, "few", "gather-early"), (8, "few", "finalize-then-gather")])
def test_sharded_equals_single(tmp_path, cpu_ops_backend, monkeypatch, world, case, mode):
    """mode = HPF_SCHEDULE: "finalize-then-gather" = reduce-scatter / one-part sharded item finalizer / all-gather of the E
    rows; "gather-early" = split finalizer, all-gather of the [numerators | base] rows before the user side (what "auto"
    resolves to without a GPU).  (3, c1): 100 items over 3 ranks -> pad rows in the item tables; (8, c1): the driver's
    largest rank count, 12-13 users and 100 items (ranges padded to multiples of 8) per rank."""
    if mode == "one-range":                       # no exchange pipelining
        mode = "finalize-then-gather"
        monkeypatch.setenv("HPF_ITEM_RANGES", "1")
    monkeypatch.setenv("HPF_SCHEDULE", mode)
    k, its = 20, 5
    if case == "few":             # more ranks than users: ranks without a single user (an empty shard) must keep in step
        rs = np.random.RandomState(5)
        nU, nI = 3, 40
        iu = np.repeat(np.arange(3), (30, 3, 12)).astype(np.uint64)
        ii = np.concatenate([rs.choice(40, n, replace=False) for n in (30, 3, 12)]).astype(np.uint64)
        Y = (rs.gamma(1, 1, size=iu.shape[0]) + 1).astype(np.int32).astype(np.float32)
    else:
        df, nU, nI = datagen.readme_counts() if case == "c1" else datagen.mid_counts(nusers=600, nitems=400, nobs=20000)
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


@pytest.mark.parametrize("world,mode", [(3, "finalize-then-gather"), (2, "gather-early")])
def test_unseeded_ranks_start_from_the_same_draw(tmp_path, cpu_ops_backend, monkeypatch, world, mode):
    """random_seed <= 0 means OS entropy (PXI:127): each rank would draw its own initial item tables and the replicas
    would drift apart; the sharded fit broadcasts rank 0's generator state, so all ranks still agree bit for bit."""
    monkeypatch.setenv("HPF_SCHEDULE", mode)
    k, its = 12, 3
    spawn_ranks(dist_worker.run, lambda port: (world, port, str(tmp_path), k, its, "c1-entropy"), world, str(tmp_path))
    outs = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    for r in range(1, world):
        for n in NAMES:
            assert np.isfinite(outs[r][n]).all() and np.array_equal(outs[r][n], outs[0][n]), (r, n)
