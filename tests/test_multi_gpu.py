"""The N>1 path on REAL links: one process per GPU, torch.distributed on the nccl backend (RCCL), the exchange crossing
xGMI -- SURVEY.md section 8(e).  Everything here needs a node with at least two GPUs and SKIPS on a one-GPU box (where
tests/test_hip_parity.py / test_full_size.py run the same schedules with the ranks sharing the GPU).  The reference has no
counterpart (single-node OpenMP, /root/reference/hpfrec/cython_loops.pxi:4); the arithmetic being distributed is
PXI:227-259 and the yardstick is the REAL reference's output (tests/golden/large_full.npz)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import datagen
from conftest import GOLDEN, spawn_ranks

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ("Theta", "Beta", "Gamma_shp", "Gamma_rte", "Lambda_shp", "Lambda_rte", "k_rte", "t_rte")
TUNING = ("HPF_SCHEDULE", "HPF_ITEM_RANGES", "HPF_DIRECT_PREFETCH", "HPF_FORCE_SHARDED", "HPF_NATIVE_SHARD",
          "HPF_SHARD_SWEEP_BPC", "HPF_ITEM_SWEEP_BPC", "HPF_BENCH_SELFTEST_GLOO", "HPF_VERIFY_FIRST", "HPF_TEST_NATIVE_GLOO")


def _gpus():
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def _worlds():
    """2 and every GPU of the node (one entry on a two-GPU node)."""
    n = min(_gpus(), 8)
    return sorted({2, n}) if n >= 2 else [2]


def _need(world):
    if _gpus() < world:
        pytest.skip("needs %d GPUs, this box has %d" % (world, _gpus()))


def _maxrel(a, b):
    return float(np.max(np.abs(a - b) / np.abs(b)))


@pytest.mark.parametrize("sched", ["direct", "gather-early", "finalize-then-gather"])
@pytest.mark.parametrize("world", _worlds())
def test_schedules_on_real_links_vs_the_reference(tmp_path, monkeypatch, world, sched):
    """The large golden's matrix (200k x 50k, 5.4M nonzeros, k = 50) fitted by `world` processes, one per GPU, RCCL
    underneath torch.distributed, each C-issued schedule in turn: 3 iterations -- so all three of the first-iteration
    checks against the call-by-call form run, on real links -- then every rank's sub-sampled rows and float64 column sums
    of all eight arrays within north_star's 1e-4 of what hpfrec itself computed, replicas bit-identical, the schedule the
    one asked for (no fall-back), the iteration issued from C."""
    _need(world)
    import dist_worker
    for v in TUNING:
        monkeypatch.delenv(v, raising=False)
    monkeypatch.setenv("HPF_SCHEDULE", sched)
    g = np.load(os.path.join(GOLDEN, "large_full.npz"))
    its = 3
    spawn_ranks(dist_worker.run, lambda port: (world, port, str(tmp_path), 50, its, "large", "cuda-per-rank"), world,
                str(tmp_path))
    outs = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    for r in range(world):
        assert str(outs[r]["schedule"]) == sched, (r, outs[r]["schedule"], str(outs[r]["failed_schedules"]))
        assert int(outs[r]["native_plans"]) >= 1 and int(outs[r]["checked_iterations"]) == 3
        for n in NAMES:
            assert _maxrel(outs[r][n + "_rows"], g["it%d_%s_rows" % (its, n)]) < 1e-4, (r, n)
            assert float(np.max(np.abs(outs[r][n + "_colsum64"] / g["it%d_%s_colsum64" % (its, n)] - 1))) < 1e-4, (r, n)
            assert np.array_equal(outs[r][n + "_rows"], outs[0][n + "_rows"]), (r, n)
            assert np.array_equal(outs[r][n + "_colsum64"], outs[0][n + "_colsum64"]), (r, n)


@pytest.mark.parametrize("sched", ["gather-early", "finalize-then-gather"])
def test_call_by_call_forms_on_rccl(tmp_path, monkeypatch, sched):
    """The Python-issued forms (torch.distributed collectives on RCCL, in order / overlapped on the exchange stream) --
    the checker of the C-issued iterations and the last resort of the fall-back ladder -- between two GPUs."""
    _need(2)
    import dist_worker
    for v in TUNING:
        monkeypatch.delenv(v, raising=False)
    monkeypatch.setenv("HPF_SCHEDULE", sched)
    monkeypatch.setenv("HPF_NATIVE_SHARD", "0")
    g = np.load(os.path.join(GOLDEN, "large_full.npz"))
    its = 3
    spawn_ranks(dist_worker.run, lambda port: (2, port, str(tmp_path), 50, its, "large", "cuda-per-rank"), 2, str(tmp_path))
    outs = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(2)]
    for r in range(2):
        assert str(outs[r]["schedule"]) == sched + ", call by call" and int(outs[r]["native_plans"]) == 0
        for n in NAMES:
            assert _maxrel(outs[r][n + "_rows"], g["it%d_%s_rows" % (its, n)]) < 1e-4, (r, n)
            assert np.array_equal(outs[r][n + "_rows"], outs[0][n + "_rows"]), (r, n)


@pytest.mark.parametrize("world", _worlds())
def test_direct_exchange_soak_on_real_links(tmp_path, monkeypatch, world):
    """120 iterations of the direct exchange between `world` GPUs, run TWICE: a pull that ever read a peer's buffer too
    early, too late, or from a stale cache line shows as a run-to-run or rank-to-rank difference (the arithmetic is
    deterministic: no atomics, fixed summation orders)."""
    _need(world)
    import dist_worker
    for v in TUNING:
        monkeypatch.delenv(v, raising=False)
    monkeypatch.setenv("HPF_SCHEDULE", "direct")
    its, k = 120, 20
    runs = []
    for rep in range(2):
        d = tmp_path / ("run%d" % rep)
        d.mkdir()
        spawn_ranks(dist_worker.run, lambda port: (world, port, str(d), k, its, "mid", "cuda-per-rank"), world, str(d))
        runs.append([np.load(os.path.join(str(d), "rank%d.npz" % r)) for r in range(world)])
    for r in range(world):
        assert str(runs[0][r]["schedule"]) == "direct" and int(runs[0][r]["native_plans"]) >= 1
        for n in NAMES:
            a = runs[0][r][n]
            assert np.isfinite(a).all(), (r, n)
            assert np.array_equal(a, runs[1][r][n]), (r, n, "run to run")
            assert np.array_equal(a, runs[0][0][n]), (r, n, "rank to rank")


@pytest.mark.parametrize("world", _worlds())
def test_sharded_equals_single_gpu_on_real_links(tmp_path, monkeypatch, world):
    """1 GPU against N GPUs at <= 1e-5 after 5 iterations (the summation order changes, nothing else)."""
    _need(world)
    import dist_worker
    from hpfrec_amd import cython_loops_float as be
    for v in TUNING:
        monkeypatch.delenv(v, raising=False)
    k, its = 20, 5
    df, nU, nI = datagen.mid_counts(nusers=600, nitems=400, nobs=20000)
    Y, iu, ii = datagen.triplets(df)
    Theta = np.empty((nU, k), np.float32)
    Beta = np.empty((nI, k), np.float32)
    i, temp, llk = be.fit_hpf(0.3, 0.3, 1.0, 0.3, 0.3, 1.0, Y, iu, ii, Theta, Beta, its, "maxiter", its, 1e-3, 0, 0, None, 0,
                              np.zeros(1, np.uint64), "", 123, 1, 1, 0, 0, np.empty(0, np.float32), np.empty(0, np.uint64),
                              np.empty(0, np.uint64), 0, 1, 0)
    single = dict(zip(NAMES, (Theta, Beta) + tuple(temp)))
    spawn_ranks(dist_worker.run, lambda port: (world, port, str(tmp_path), k, its, "mid", "cuda-per-rank"), world,
                str(tmp_path))
    outs = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    for r in range(world):
        assert int(outs[r]["niter"]) == i and abs(float(outs[r]["llk"]) / float(llk) - 1) < 1e-6
        assert str(outs[r]["schedule"]) == "direct"            # the library default, not fallen back
        for n in NAMES:
            assert _maxrel(outs[r][n], single[n]) < 1e-5, (r, n)
            assert np.array_equal(outs[r][n], outs[0][n]), (r, n)


@pytest.mark.parametrize("world", _worlds())
def test_p2p_primitives_across_devices(world):
    """tools/p2p_probe.py with one GPU per process: hipIpc of coarse- and fine-grained memory between DIFFERENT devices
    (hipIpcMemLazyEnablePeerAccess), a peer's values visible after its flag (system-scope release / acquire over the
    link), the rank-order granule all-reduce, and the bounded time-out."""
    _need(world)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", P2P_PROBE_DEVICE_PER_RANK="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "p2p_probe.py"), str(world)], env=env, cwd=ROOT,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-3000:]
    assert "world %d: exit 0" % world in out.stdout, out.stdout[-3000:]
    assert "time-out path OK" in out.stdout and "one GPU per rank" in out.stdout


def _bench(args, env_extra, timeout=1200):
    env = dict(os.environ, HPF_BENCH_WATCHDOG_S="600", **env_extra)
    for v in TUNING:
        if v not in env_extra:
            env.pop(v, None)
    for v in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):       # NO launcher: bench.py starts its own ranks
        env.pop(v, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                         timeout=timeout, cwd=ROOT)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stdout[-2000:] + out.stderr[-3000:]
    return json.loads(lines[0])


def _check_line(d, ranks):
    assert d["n_gpus"] == ranks and d["config"]["state_finite"] is True and d["value"] > 0
    assert abs(d["ms_per_step_long"] / d["ms_per_step"] - 1) < 0.5 and d["steps_long"] >= d["steps"]
    co = d["collective"]
    assert "error" not in co, co
    lp = co["link_probe"]
    assert "error" not in lp, lp
    assert lp["ranks"] == ranks and lp["values_ok"] is True and len(lp["per_rank"]) == ranks
    assert all(len(g["pull_GBps"]) == ranks - 1 and min(g["pull_GBps"].values()) > 0 for g in lp["per_rank"])
    assert lp["flag_round_trip_us_0_1"] > 0 and lp["vec_allreduce_us_max"] > 0
    lc = lp["library_collectives"]
    assert "error" not in lc and lc["all_reduce"]["busbw_GBps"] > 0 and lc["all_gather"]["ms"] > 0
    return lp


def test_bench_launches_its_own_ranks_selftest():
    """`python bench.py --gpus 2 --steps 6 --warmup 2` with NO launcher and no WORLD_SIZE (how a driver runs the N=1
    line): the script re-executes itself under torch.distributed.run.  On a one-GPU box the two ranks share the GPU
    (HPF_BENCH_SELFTEST_GLOO=1: a code-path test, not a measurement); one valid line, carrying collective.link_probe."""
    if _gpus() < 1:
        pytest.skip("no GPU")
    d = _bench(["--gpus", "2", "--steps", "6", "--warmup", "2", "--workload", "small"], {"HPF_BENCH_SELFTEST_GLOO": "1",
                                                                                      "HPF_DIRECT_TIMEOUT_MS": "60000"})
    lp = _check_line(d, 2)
    assert lp["shared_device"] is True and lp["library_collectives"]["backend"] == "gloo"
    assert d["config"]["iteration_issued_by"].startswith("one C call")


@pytest.mark.parametrize("world", _worlds())
def test_bench_launches_its_own_ranks_on_real_gpus(world):
    """The same command on a multi-GPU node: one rank per GPU on RCCL; the line's link probe reports every (rank, peer)
    pull rate over the links and RCCL's own bus bandwidth; the exchange spans N ranks on N different devices."""
    _need(world)
    d = _bench(["--gpus", str(world), "--steps", "6", "--warmup", "2", "--workload", "small"], {})
    lp = _check_line(d, world)
    assert lp["shared_device"] is False and lp["library_collectives"]["backend"] == "nccl"
    assert d["collective"]["ranks"] == world and d["collective"]["ranks_equal_n_gpus"] is True
    # the first C-issued iterations of every schedule that ran were checked against the call-by-call form over the links
    assert d["config"]["checked_iterations_passed"].get("direct") == 3, d["config"]
    assert d["config"]["schedules_struck_by_the_check"] == [], d["config"]
