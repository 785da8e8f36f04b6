#!/usr/bin/env python3
"""Benchmark of the HPF full-batch CAVI sweep on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one full CAVI iteration (phi -> shape accumulation -> all rate updates -> the
per-iteration all-reduce when N>1; cython_loops.pxi:232-259 of the reference) over the synthetic
MillionSong-shaped matrix of BASELINE config C3 (1M x 380k, ~48M nnz, k=50, fp32), data resident in
HBM before the timed region.  With N>1 the SAME matrix is sharded by users (strong scaling).

Prints ONE JSON line on rank 0 (see the bench contract in the task): metric/value/unit ... plus
  "roofline":     the dominant kernel (sweep_kernel) priced against HBM peak, timed with HIP events
                  on the launch stream inside the timed region; `traffic` = HBM-side bytes per launch from the
                  committed rocprofv3 --pmc passes, null when they were measured on another kernel source;
  "cpu_baseline": the CPU oracle (oracle/, a port of the reference's Cython loops) timed on this
                  node's host cores on a bounded sample of the same workload (rank 0, N=1 only).
N>1: the LIBRARY DEFAULT (HPF_SCHEDULE=auto: the direct, peer-mapped exchange) is timed first and its line is held; then at
most five alternatives run 20 iterations each (--autotune-all: the long list), every rank taking the same decision, and
only a candidate that beats the default by more than 2 % is benchmarked in its place (config.exchange_autotune); a
watchdog reports the held line should anything after it stop making progress.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from hpfrec_amd import cavi, cython_loops_float as backend  # noqa: E402
from hpfrec_amd.ops_hip import HipOps  # noqa: E402

HBM_PEAK = 8.0e12  # B/s, MI355X spec (/opt/skills/guides/MI355X_MICROARCH.md)

SCHEDULE_TEXT = {   # config.parallelism of an N>1 line: (world, item ranges)
    "direct": "users sharded x%d; item statistics in %d ranges PULLED from the peers' mapped exchange buffers by the shape "
              "half of the item finalizer (rank-order sum), finished rows pulled back under the user sweep; no collective library",
    "gather-early": "users sharded x%d; item statistics reduce-scattered in %d pipelined ranges (RCCL), split item finalizer, "
                    "[numerators | base rate] rows all-gathered under the user sweep",
    "finalize-then-gather": "users sharded x%d; item statistics reduce-scattered in %d pipelined ranges (RCCL), each rank "
                            "finalizes 1/N of the items, new E rows all-gathered under the next item sweep",
}

WORKLOADS = {
    # name: (nU, nI, target nnz, k, label)
    "c3": (1_000_000, 380_000, 48_000_000, 50, "C3 MillionSong-shaped synthetic 1M x 380k, 48M nnz, k=50, full batch"),
    "c2": (138_000, 27_000, 20_000_000, 50, "C2 MovieLens-20M-shaped synthetic 138k x 27k, 20M nnz, k=50, full batch"),
    "c4": (1_000_000, 380_000, 48_000_000, 100, "C4 MillionSong-shaped synthetic 1M x 380k, 48M nnz, k=100, full batch"),
    "small": (100_000, 30_000, 2_000_000, 50, "small synthetic 100k x 30k, 2M nnz, k=50 (debug)"),
    "c3u": (1_000_000, 380_000, 48_000_000, 50, "C3 shape with UNIFORM item popularity (no hot items; SURVEY 8d variant)"),
    "tiny": (2_000, 1_000, 50_000, 50, "tiny synthetic 2k x 1k, 50k nnz, k=50 (host-overhead probe)"),
    "k30": (1_000_000, 380_000, 48_000_000, 30, "C3 matrix with k=30 (ld=32: 8 nonzeros per wave step)"),
    "k200": (1_000_000, 380_000, 48_000_000, 200, "C3 matrix with k=200 (ld=256: 1 nonzero per wave step)"),
    # probes for the hot/cold split of the gathers: the nonzeros of C3 that fall on its 16k most
    # popular items (4 MB of E rows: L2-resident in every XCD), and the rest
    "hot": (1_000_000, 16_000, 13_500_000, 50, "probe: 1M x 16k, 13.5M nnz, k=50 (item table = 4 MB)"),
}


POWER = {"c3u": 1.0, "hot": 1.0}   # item-popularity exponent per workload (default 2.5)


def synth_on_device(nU, nI, nnz_target, device, seed=1, item_power=2.5, sigma=1.0):
    """SURVEY.md section 8d generator, on the GPU: log-normal user degrees, power-law item popularity
    (i = floor(nI * U^2.5)), unique (u,i) pairs, Y = 1 + floor(Gamma(1,1)).  Deterministic per seed."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    deg = torch.exp(sigma * torch.randn(nU, generator=g, device=device, dtype=torch.float64))
    deg = torch.clamp(torch.round(deg * (nnz_target * 1.012 / deg.sum())), min=1).to(torch.int64)
    total = int(deg.sum().item())
    u = torch.repeat_interleave(torch.arange(nU, device=device, dtype=torch.int64), deg, output_size=total)
    r = torch.rand(total, generator=g, device=device, dtype=torch.float64)
    i = torch.clamp((nI * r.pow(item_power)).to(torch.int64), max=nI - 1)
    key = torch.unique(u * nI + i)
    del u, i, r
    u = key // nI
    i = key - u * nI
    e = torch.rand(key.shape[0], generator=g, device=device, dtype=torch.float32)
    y = 1.0 + torch.floor(-torch.log1p(-e))
    perm = torch.randperm(key.shape[0], generator=g, device=device)  # the reference takes unsorted COO
    return u[perm], i[perm], y[perm]


class TimedOps(HipOps):
    """HipOps that brackets every launch with HIP events on the launch stream (torch's current
    stream is the stream the C ABI launches on)."""

    #: the launches bracketed in the timed region: the dominant kernel only (two event records per launch cost
    #: 1.4 % of a C3 iteration when all six launches of an iteration are bracketed); recording = "all" brackets every
    #: launch (the untimed breakdown pass)
    DOMINANT = ("sweep", "sweep_finalize", "sweep_prefinalize")

    def __init__(self, device):
        super().__init__(device)
        self.recording = False
        self.events = {}

    def _timed(self, name, fn, *a, **kw):
        if not self.recording or (self.recording != "all" and name not in self.DOMINANT):
            return fn(*a, **kw)
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*a, **kw)
        e1.record()
        self.events.setdefault(name, []).append((e0, e1))
        return out

    def sweep(self, *a, **kw):
        return self._timed("sweep", super().sweep, *a, **kw)

    def sweep_finalize(self, *a, **kw):
        return self._timed("sweep_finalize", super().sweep_finalize, *a, **kw)

    def sweep_prefinalize(self, *a, **kw):
        return self._timed("sweep_prefinalize", super().sweep_prefinalize, *a, **kw)

    def row_finalize(self, *a, **kw):
        return self._timed("row_finalize", super().row_finalize, *a, **kw)

    def colsum_reduce(self, *a, **kw):
        return self._timed("colsum_reduce", super().colsum_reduce, *a, **kw)

    def segsum(self, *a, **kw):
        return self._timed("segsum", super().segsum, *a, **kw)

    def summary(self):
        out = {}
        for name, evs in self.events.items():
            ms = [a.elapsed_time(b) for a, b in evs]
            out[name] = {"calls": len(ms), "avg_ms": float(np.mean(ms)), "total_ms": float(np.sum(ms))}
        return out


def cpu_baseline(nU, nI, k, nnz_full, sample_users=200_000, iters=4, device=None, triplets=None):
    """The CPU oracle (port of the reference's loops: materialised phi, per-nonzero double psi/log/exp, serial scatter,
    numpy rate updates) on this node's host cores.  With `triplets` (the benchmark's own matrix, host arrays) and enough
    host memory for phi (nnz*k*4 bytes: 9.7 GB at C3) the FULL workload is timed: 1 warm + 3 timed iterations
    (BASELINE.md section 4.3).  Otherwise a bounded sample: the first `sample_users` users of a same-shaped matrix (all
    items kept), extrapolated linearly in nnz (the reference's cost is per nonzero)."""
    from oracle import hpf_oracle as O
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import datagen
    full = False
    if triplets is not None:
        try:
            import psutil
            need = triplets[2].shape[0] * k * 4 * 1.25 + (nU + nI) * k * 4 * 24
            full = psutil.virtual_memory().available > need + (8 << 30)
        except Exception:   # noqa: BLE001
            full = False
    if full:
        iu, ii, Y = triplets
        sample_users, iters = nU, 3
    else:
        sample_users = min(sample_users, nU)
        nnz_s = int(nnz_full * (sample_users / nU))
        iu, ii, Y = datagen.synthetic_hpf_shaped(sample_users, nI, nnz_s, seed=1)
    cores = O.max_threads()
    hy = O.Hyper(k, 0.3, 0.3, 1.0, 0.3, 0.3, 1.0)
    st = O.State(sample_users, nI, hy, 123)
    init = {n: getattr(st, n).copy() for n in O.State.names}      # for the parity check below
    phi = np.empty((Y.shape[0], k), dtype=np.float32)
    Yc, iuc, iic = O._f32(Y), O._ind(iu), O._ind(ii)
    snaps = {}
    O.cavi_iteration(st, hy, Yc, iuc, iic, phi, 0, cores)  # warm (page-in phi)
    t0 = time.time()
    t_snap = 0.0
    for i in range(iters):
        O.cavi_iteration(st, hy, Yc, iuc, iic, phi, 0, cores)
        if i + 2 == 3:                                       # state after 3 iterations, for the parity check below
            ts = time.time()
            snaps[3] = {n: getattr(st, n).copy() for n in ("Theta", "Beta")}
            t_snap = time.time() - ts
    dt = (time.time() - t0 - t_snap) / iters
    total_its = iters + 1
    snaps[total_its] = {n: getattr(st, n) for n in ("Theta", "Beta")}
    per_full = dt * (nnz_full / Y.shape[0])
    out = {"value": 1.0 / per_full, "unit": "iters/s", "cores": cores, "kind": "port",
           "sample": ("the FULL workload: %d users x %d items, %d nnz, k=%d, %d timed iterations at %.2f s/iter (after 1 warm "
                      "iteration; phi materialised: %.1f GB)" % (sample_users, nI, Y.shape[0], k, iters, dt,
                                                                 Y.shape[0] * k * 4 / 1e9)) if full else
                     ("%d users x %d items, %d nnz, k=%d, %d timed iterations at %.2f s/iter; scaled by nnz to %d nnz"
                      % (sample_users, nI, Y.shape[0], k, iters, dt, nnz_full))}
    out["calibration_vs_reference"] = ("the port runs 12-15 % FASTER than the real Cython extension (port / reference time per "
                                       "iteration 0.85-0.88, timed side by side in the build container, 8 cores, 2M nnz: "
                                       "profiles/r02_cpu_calibration.txt) -- the reference itself cannot run on the GPU box")
    # (NOT a measurement: what the real extension would be expected to give on these cores -- the port's measured rate x the
    #  mid-point of the ratio the two showed side by side in the build container)
    out["estimates"] = {"reference_extension_iters_per_s": out["value"] * 0.865,
                        "how": "cpu_baseline.value x 0.865 (mid-point of the port / reference time ratio 0.85-0.88 of "
                               "profiles/r02_cpu_calibration.txt); an estimate, not timed here"}
    # the oracle is the checker: the HIP path on the same sample, same start, same numbers of iterations
    if device is not None:
        hyd = cavi.Hyper(k, 0.3, 0.3, 1.0, 0.3, 0.3, 1.0)
        m = cavi.FullBatchCavi(HipOps(device), device, torch.from_numpy(iu.astype(np.int64)),
                               torch.from_numpy(ii.astype(np.int64)), torch.from_numpy(Y), sample_users, nI, hyd)
        m.load_state(init["Gamma_shp"], init["Gamma_rte"], init["Lambda_shp"], init["Lambda_rte"], init["k_rte"],
                     init["t_rte"], init["Theta"], init["Beta"])
        got = {}
        for i in range(total_its):
            m.iterate(True)
            if i + 1 in snaps:
                got[i + 1] = {n: m.fetch(n) for n in ("Theta", "Beta")}
        # the same iterations with the two column sums in the reference's (numpy's) own order on the device -- a
        # diagnostic mode of the driver (HPF_COLSUM_ORDER=reference, hpf_hip_colsum_sequential_f32): what is left then
        # is everything BUT the summation order of Theta.sum(axis=0) / Beta.sum(axis=0)
        m.ref_sums = True
        m.load_state(init["Gamma_shp"], init["Gamma_rte"], init["Lambda_shp"], init["Lambda_rte"], init["k_rte"],
                     init["t_rte"], init["Theta"], init["Beta"])
        got_ref = {}
        for i in range(total_its):
            m.iterate(True)
            if i + 1 in snaps:
                got_ref[i + 1] = {n: m.fetch(n) for n in ("Theta", "Beta")}
        del m
        torch.cuda.empty_cache()

        def worst(a, b):
            return max(float(np.max(np.abs(a[n] - b[n]) / np.abs(b[n]))) for n in ("Theta", "Beta"))
        # numpy's float32 row-by-row column sums (PXI:236,255), which the port reproduces bit for bit, are themselves
        # ~1e-4 off at 10^5..10^6 rows; the same port with float64 column sums shows what is left without that
        snaps64 = {}
        if not full:        # (a diagnostic, bounded to the sample: the same port with float64 column sums)
            st64 = O.State(sample_users, nI, hy, 123)
            for i in range(total_its):
                O.cavi_iteration(st64, hy, Yc, iuc, iic, phi, 0, cores, exact_colsums=True)
                if i + 1 in snaps:
                    snaps64[i + 1] = {n: getattr(st64, n).copy() for n in ("Theta", "Beta")}
        out["parity_vs_gpu_on_sample"] = {
            "max_rel_dev_Theta_Beta": {"after_%d_iterations" % n: dict(
                vs_port=worst(got[n], snaps[n]),
                vs_port_with_the_column_sums_in_numpys_order_on_the_device=worst(got_ref[n], snaps[n]),
                **({"vs_port_with_float64_column_sums": worst(got[n], snaps64[n])} if n in snaps64 else {}))
                for n in sorted(snaps)},
            "pinned_by": "tests/golden/large_full.npz (the real reference at 200k x 50k, 5.4M nnz; oracle bit-exact, GPU "
                         "within 1e-4: tests/test_oracle.py::test_large_bit_exact, test_hip_parity.py::test_large_vs_golden)",
            "note": "element-wise parity is a short-horizon property (rounding noise grows ~x1.2 per iteration); the port "
                    "reproduces numpy's float32 row-by-row column sums of the reference (PXI:236,255), whose own error is "
                    "~1e-4 at 1e5..1e6 rows (SURVEY.md section 7); the GPU sums them in fp64 trees"}
    return out


def _svi_epoch_bytes(nU, nI, nnz, k, per=65536):
    """Algorithmic bytes of one stochastic epoch at users_per_batch = items_per_batch = per (mean of a user and an item
    epoch; fp32 values, int32 ids, every gather counted once, no cache credit -- the convention of SURVEY.md section 8d
    applied to the reference's statements PXI:277-377): per batch of B rows with n nonzeros touching R rows of the other
    side, the two sweeps n*(8 + 8k); the reference's whole-table statements (PXI:300,318,322 / 352,370,374) -- the batch
    side's rates, means and shapes of ALL its rows, 12k per row, and the other side's means, 8k per row; E rows and
    accumulators of the touched rows (B + R)*16k.  Summed over an epoch: sum n = nnz, sum B = the side's rows."""
    nb_u, nb_i = -(-nU // per), -(-nI // per)
    R_u, R_i = min(nI, nnz // nb_u), min(nU, nnz // nb_i)       # (upper bounds on the other side's rows per batch)
    b_user = nnz * (8 + 8 * k) + nb_u * (nU * 12 * k + nI * 8 * k + R_u * 16 * k) + nU * 16 * k
    b_item = nnz * (8 + 8 * k) + nb_i * (nI * 12 * k + nU * 8 * k + R_i * 16 * k) + nI * 16 * k
    return (b_user + b_item) / 2.0, nb_u, nb_i


def other_workloads(device, host_triplets, nU, nI, with_oracle=True):
    """The other single-GPU configurations of BASELINE.json, measured in the driver's own run AFTER the timed region and
    the long pass (never inside them; `value` does not depend on anything here): C5 stochastic epochs (PXI:262-377), C5
    read literally as HPF.partial_fit calls (PXI:423-473), C2 full batch.  Every entry names the kernel source it ran on."""
    import warnings
    import pandas as pd
    from hpfrec_amd import svi, HPF
    out = {"kernel_source_sha16": kernel_source_sha16(), "note": "untimed extras: measured after the timed region"}
    iu_h, ii_h, y_h = host_triplets
    nnz = int(y_h.shape[0])
    k5, per = 200, 65536
    IU, II = iu_h.astype(np.uint64), ii_h.astype(np.uint64)
    none_f, none_i = np.empty(0, np.float32), np.empty(0, np.uint64)

    def svi_fit(Y, iu, ii, st, n_u, n_i, epochs, upb=per, ipb=per, verbose=0):
        Theta = np.empty((n_u, k5), np.float32)
        Beta = np.empty((n_i, k5), np.float32)
        i, temp, _ = backend.fit_hpf(0.3, 0.3, 1.0, 0.3, 0.3, 1.0, Y, iu, ii, Theta, Beta, epochs, "maxiter", 0, 1e-3, upb, ipb,
                                     lambda x: 1 / np.sqrt(x + 2), 0, st, "", 123, verbose, 1, 0, 0, none_f, none_i, none_i,
                                     0, 1, 0)
        return i, dict(zip(("Gamma_shp", "Gamma_rte", "Lambda_shp", "Lambda_rte", "k_rte", "t_rte"), temp), Theta=Theta,
                       Beta=Beta)

    # ---- C5: stochastic epochs through fit_hpf, device-synchronised epoch loop (svi.SVI_TIMINGS) -----------------------
    try:
        os.environ["HPF_TIMING"] = "1"
        epochs = 8
        svi_fit(y_h, IU, II, np.zeros(1, np.uint64), nU, nI, 2)                     # warm: code objects, allocator
        i_last, fit = svi_fit(y_h, IU, II, np.zeros(1, np.uint64), nU, nI, epochs)
        loop = dict(svi.SVI_TIMINGS)
        ms_epoch = loop["seconds"] / loop["epochs"] * 1e3
        b_epoch, nb_u, nb_i = _svi_epoch_bytes(nU, nI, nnz, k5)
        finite = bool(np.isfinite(fit["Theta"]).all() and np.isfinite(fit["Beta"]).all() and (fit["Theta"] > 0).all())
        del fit
        out["c5_svi"] = {
            "workload": "C5 stochastic VI: C3 matrix (%d x %d, %d nnz), k=%d, users_per_batch = items_per_batch = %d (%d user / "
                        "%d item batches per epoch), fit_hpf from host arrays" % (nU, nI, nnz, k5, per, nb_u, nb_i),
            "epochs_timed": int(loop["epochs"]), "ms_per_epoch": ms_epoch,
            "ms_per_batch": ms_epoch / ((nb_u + nb_i) / 2.0),
            "timing": "device-synchronised wall time of the epoch loop alone (no llk check inside: stop_crit='maxiter', "
                      "verbose=0), %d epochs alternating item / user epochs; reference loop PXI:262-377" % loop["epochs"],
            "algorithmic_bytes_per_epoch": b_epoch, "frac_of_hbm_peak": b_epoch / (ms_epoch * 1e-3) / HBM_PEAK,
            "host_seconds_in_the_loop": {"preparing": loop.get("host_prepare_s"), "issuing": loop.get("host_issue_s")},
            "last_epoch_index": int(i_last), "state_finite": finite}
    except Exception as exc:   # noqa: BLE001  (the headline line must survive)
        out["c5_svi"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}
    finally:
        os.environ.pop("HPF_TIMING", None)
    torch.cuda.empty_cache()

    # ---- C5 parity in this run: 2 epochs on a <= 100k-user slice against the oracle's fit_svi -------------------------
    if with_oracle and "error" not in out["c5_svi"]:
        try:
            from oracle import hpf_oracle as O
            n_u = 100_000
            keep = iu_h < n_u
            Ys, ius, iis, st_ix_u = O.svi_inputs_like_reference(y_h[keep], iu_h[keep], ii_h[keep], n_u, nI)
            _, got = svi_fit(Ys, ius, iis, st_ix_u, n_u, nI, 2)
            # (the same fit with every column sum of the steps in numpy's own order on the device: HPF_COLSUM_ORDER=reference)
            os.environ["HPF_COLSUM_ORDER"] = "reference"
            try:
                _, got_ref = svi_fit(Ys, ius, iis, st_ix_u, n_u, nI, 2)
            finally:
                os.environ.pop("HPF_COLSUM_ORDER", None)
            dev_of = {}
            for exact in (False, True):
                st = O.fit_svi(Ys, ius, iis, st_ix_u, n_u, nI, k5, 2, 123, per, per, nthreads=O.max_threads(), exact_colsums=exact)
                dev_of["float64 column sums" if exact else "as it is"] = max(
                    float(np.max(np.abs(got[n] - getattr(st, n)) / np.abs(getattr(st, n)))) for n in O.State.names)
                if not exact:
                    dev_of["as it is, with HPF_COLSUM_ORDER=reference on the device"] = max(
                        float(np.max(np.abs(got_ref[n] - getattr(st, n)) / np.abs(getattr(st, n)))) for n in O.State.names)
            out["c5_svi"]["parity_in_this_run"] = {
                "slice": "users < %d of the same matrix (all items): %d nnz, 2 epochs (one item, one user), k=%d, %d-row "
                         "batches" % (n_u, int(Ys.shape[0]), k5, per),
                "max_rel_dev_all_eight_arrays_vs_oracle_fit_svi": dev_of,
                "note": "the oracle restates PXI:262-377 and is bit-exact to the reference's own captures "
                        "(tests/golden/svi_large.npz); numpy's sequential float32 column sums over 1e5..4e5 rows are the noisy "
                        "side (SURVEY.md section 7): the last figure replaces them by float64 sums in the oracle, the middle one "
                        "forms them in numpy's own order on the device (the reference's arithmetic, matched as it is)"}
            del got, got_ref, st
        except Exception as exc:   # noqa: BLE001
            out["c5_svi"]["parity_in_this_run"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}

    # ---- C5 read literally: HPF.partial_fit, 65,536-user and 65,536-item calls on the resident k=200 model ----------
    try:
        order = np.argsort(iu_h, kind="stable")
        U1, I1, Y1 = iu_h[order], ii_h[order], y_h[order]
        order = np.argsort(ii_h, kind="stable")
        U2, I2, Y2 = iu_h[order], ii_h[order], y_h[order]
        del order
        ptr_u = np.searchsorted(U1, np.arange(0, nU + per, per))
        ptr_i = np.searchsorted(I2, np.arange(0, nI + per, per))

        def frames(U, I, C, ptr):
            return [pd.DataFrame({"UserId": U[a:b], "ItemId": I[a:b], "Count": C[a:b]}) for a, b in zip(ptr[:-1], ptr[1:])
                    if b > a]
        ub, ib = frames(U1, I1, Y1, ptr_u), frames(U2, I2, Y2, ptr_i)
        m = HPF(k=k5, reindex=False, keep_data=False, random_seed=7, verbose=False)
        res = {}
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m.partial_fit(ub[0], nusers=nU, nitems=nI)        # uploads the initial state once
            m.partial_fit(ib[-1], batch_type="items")
            torch.cuda.synchronize()
            for name, batches, kind in (("user", ub, "users"), ("item", ib, "items")):
                t_call = []
                for b in batches:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    m.partial_fit(b, batch_type=kind)
                    torch.cuda.synchronize()
                    t_call.append(time.perf_counter() - t0)
                n_b = np.array([b.shape[0] for b in batches], dtype=np.float64)
                full = [t for t, b in zip(t_call, batches) if b[("UserId" if kind == "users" else "ItemId")].nunique() == per]
                res[name] = {"calls": len(batches), "median_ms_per_call": 1e3 * float(np.median(t_call)),
                             "median_ms_per_full_%d_row_call" % per: 1e3 * float(np.median(full)) if full else None,
                             "min_ms": 1e3 * float(min(t_call)), "max_ms": 1e3 * float(max(t_call)),
                             "triplets_per_call": {"mean": float(n_b.mean()), "min": float(n_b.min()), "max": float(n_b.max())},
                             "us_per_1000_triplets_over_all_calls": 1e9 * float(np.sum(t_call)) / float(n_b.sum())}
        th = m._state.rows("Theta", [0, nU - 1])
        res["state_finite"] = bool(np.isfinite(th).all() and (th > 0).all())
        res["workload"] = ("C5 read literally: HPF.partial_fit(batch_type=...) on the resident k=%d model of the C3 matrix, every "
                           "call = ALL interactions of <= %d users (items) as a pandas frame from host memory; wall time per "
                           "call, device-synchronised; reference PXI:423-473, INIT:856-927" % (k5, per))
        out["c5_partial_fit"] = res
        del m, ub, ib, U1, I1, Y1, U2, I2, Y2
    except Exception as exc:   # noqa: BLE001
        out["c5_partial_fit"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}
    torch.cuda.empty_cache()

    # ---- C4's per-GPU kernel on ONE GPU: the C3 matrix at k = 100 (the 8-GPU run itself is the driver's scaling bench) ----
    try:
        k4 = WORKLOADS["c4"][3]
        iu = torch.from_numpy(iu_h.astype(np.int64)).to(device)
        ii = torch.from_numpy(ii_h.astype(np.int64)).to(device)
        y = torch.from_numpy(y_h).to(device)
        hy = cavi.Hyper(k4, 0.3, 0.3, 1.0, 0.3, 0.3, 1.0)
        Theta = np.empty((nU, k4), np.float32)
        Beta = np.empty((nI, k4), np.float32)
        init = backend.initialize_parameters(Theta, Beta, 123, 0.3, 0.3, 1.0, 0.3, 0.3, 1.0)
        m4 = cavi.FullBatchCavi(HipOps(device), device, iu, ii, y, nU, nI, hy)
        m4.load_state(init[0], init[1], init[2], init[3], init[4], init[5], Theta, Beta)
        del iu, ii, y, init, Theta, Beta
        for _ in range(5):
            m4.iterate(True)
        torch.cuda.synchronize()
        steps4 = 60
        t0 = time.perf_counter()
        for _ in range(steps4):
            m4.iterate(True)
        torch.cuda.synchronize()
        ms4 = (time.perf_counter() - t0) / steps4 * 1e3
        b4 = nnz * (8 + 8 * k4) + nU * (12 + 20 * k4) + nI * (4 + 24 * k4)
        out["c4_k100_on_one_gpu"] = {
            "workload": "the C3 matrix at k=%d (ld=128) on ONE GPU, full batch, all six output tables stored: the kernel of BASELINE "
                        "config C4 without its sharding (the 8-GPU run is the driver's scaling bench)" % k4,
            "steps": steps4, "ms_per_step": ms4, "iters_per_s": 1e3 / ms4, "algorithmic_bytes_per_iteration": b4,
            "frac_of_hbm_peak": b4 / (ms4 * 1e-3) / HBM_PEAK, "state_finite": bool(torch.isfinite(m4.Theta).all().item())}
        del m4
    except Exception as exc:   # noqa: BLE001
        out["c4_k100_on_one_gpu"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}
    torch.cuda.empty_cache()

    # ---- C2: MovieLens-20M-shaped full batch ---------------------------------------------------------------------------
    try:
        n_u, n_i, nnz_t, k2, label2 = WORKLOADS["c2"]
        iu, ii, y = synth_on_device(n_u, n_i, nnz_t, device)
        nnz2 = int(iu.shape[0])
        hy = cavi.Hyper(k2, 0.3, 0.3, 1.0, 0.3, 0.3, 1.0)
        Theta = np.empty((n_u, k2), np.float32)
        Beta = np.empty((n_i, k2), np.float32)
        init = backend.initialize_parameters(Theta, Beta, 123, 0.3, 0.3, 1.0, 0.3, 0.3, 1.0)
        m2 = cavi.FullBatchCavi(HipOps(device), device, iu, ii, y, n_u, n_i, hy)
        m2.load_state(init[0], init[1], init[2], init[3], init[4], init[5], Theta, Beta)
        del iu, ii, y
        for _ in range(10):
            m2.iterate(True)
        torch.cuda.synchronize()
        steps2 = 400
        t0 = time.perf_counter()
        for _ in range(steps2):
            m2.iterate(True)
        torch.cuda.synchronize()
        ms2 = (time.perf_counter() - t0) / steps2 * 1e3
        b2 = nnz2 * (8 + 8 * k2) + n_u * (12 + 20 * k2) + n_i * (4 + 24 * k2)
        out["c2_full_batch"] = {"workload": label2, "nnz": nnz2, "steps": steps2, "ms_per_step": ms2, "iters_per_s": 1e3 / ms2,
                                "algorithmic_bytes_per_iteration": b2, "frac_of_hbm_peak": b2 / (ms2 * 1e-3) / HBM_PEAK,
                                "note": "all six output tables stored; at this size the gathered tables (27 MB + 7 MB of E rows) "
                                        "stay in L2 / Infinity Cache: every gather is still counted at face value",
                                "state_finite": bool(torch.isfinite(m2.Theta).all().item())}
        del m2
    except Exception as exc:   # noqa: BLE001
        out["c2_full_batch"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}
    torch.cuda.empty_cache()
    return out


class HbmActivitySampler:
    """Memory-controller (UMC) activity of one GPU sampled from the driver while the long confirmation pass runs -- the
    only HBM-side observable of this platform (rocprofv3 on gfx950 has no MALL / DRAM byte counter:
    profiles/r04_rocprofv3_memory_counters_available.txt).  Sources, first one that answers: the amdgpu sysfs files
    `mem_busy_percent` and `gpu_metrics` (average_umc_activity, a uint16 percentage at byte 14 of the v1.x table), then
    `amd-smi metric --usage --json`.  10 Hz (sysfs) / as fast as the tool answers (amd-smi).  Rank 0 only; never
    inside the timed region."""

    def __init__(self, index):
        self.index = index
        self.samples = {}
        self.errors = {}
        self._stop = False
        self._thread = None
        self.card = self._find_card(index)

    @staticmethod
    def _find_card(index):
        import glob
        try:
            pr = torch.cuda.get_device_properties(index)
            want = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        except Exception:   # noqa: BLE001
            want = None
        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device"))
        cards = [c for c in cards if os.path.exists(os.path.join(c, "gpu_metrics")) or
                 os.path.exists(os.path.join(c, "mem_busy_percent"))]
        for c in cards:
            try:
                if want and os.path.basename(os.path.realpath(c)).lower().startswith(want):
                    return c
            except Exception:   # noqa: BLE001
                pass
        return cards[index] if index < len(cards) else (cards[0] if len(cards) == 1 else None)

    def _read_sysfs(self):
        import struct
        got = {}
        if self.card is None:
            return got
        try:
            got["sysfs_mem_busy_percent"] = float(open(os.path.join(self.card, "mem_busy_percent")).read().strip())
        except Exception as exc:   # noqa: BLE001
            self.errors.setdefault("sysfs_mem_busy_percent", repr(exc)[:120])
        try:
            b = open(os.path.join(self.card, "gpu_metrics"), "rb").read()
            size, fmt, rev = struct.unpack_from("<HBB", b, 0)
            if fmt == 1 and rev >= 4 and len(b) >= 16:
                gfx, umc = struct.unpack_from("<HH", b, 12)
                if umc <= 100 or umc == 0xFFFF:
                    if umc != 0xFFFF:
                        got["gpu_metrics_average_umc_activity"] = float(umc)
                        got["gpu_metrics_average_gfx_activity"] = float(gfx)
                    else:
                        self.errors.setdefault("gpu_metrics", "average_umc_activity reads 0xFFFF (not reported)")
                else:
                    self.errors.setdefault("gpu_metrics", "unexpected value %d at the v1.%d umc offset" % (umc, rev))
            else:
                self.errors.setdefault("gpu_metrics", "table format %d.%d, %d bytes: layout not known here" % (fmt, rev, len(b)))
        except Exception as exc:   # noqa: BLE001
            self.errors.setdefault("gpu_metrics", repr(exc)[:120])
        return got

    def _read_smi(self):
        import shutil
        import subprocess
        exe = shutil.which("amd-smi")
        if not exe:
            self.errors.setdefault("amd-smi", "not on PATH")
            return {}
        try:
            out = subprocess.run([exe, "metric", "-g", str(self.index), "--usage", "--json"], capture_output=True, text=True,
                                 timeout=20)
            d = json.loads(out.stdout)
            while isinstance(d, (list, dict)) and not (isinstance(d, dict) and "usage" in d):
                d = d[0] if isinstance(d, list) else next(iter(d.values()))
            u = d["usage"].get("umc_activity")
            v = u.get("value") if isinstance(u, dict) else u
            return {"amd_smi_umc_activity": float(v)}
        except Exception as exc:   # noqa: BLE001
            self.errors.setdefault("amd-smi", repr(exc)[:160])
            return {}

    def _loop(self):
        use_smi = None
        while not self._stop:
            got = self._read_sysfs()
            if not got and use_smi is not False:
                got = self._read_smi()
                use_smi = bool(got)
            for k_, v in got.items():
                self.samples.setdefault(k_, []).append(v)
            if not got and use_smi is False:
                return
            time.sleep(0.1)

    def start(self):
        import threading
        self._thread = threading.Thread(target=self._loop, daemon=True)
        self._thread.start()

    def stop(self):
        self._stop = True
        if self._thread:
            self._thread.join(timeout=30)
        out = {"card": self.card, "sources": {}, "unavailable": self.errors or None}
        for k_, v in self.samples.items():
            v = v[1:] if len(v) > 2 else v        # (the first sample may predate the pass: the driver averages over ~1 s)
            out["sources"][k_] = {"samples": len(v), "mean": float(np.mean(v)), "max": float(np.max(v)),
                                  "median": float(np.median(v))}
        return out


def umc_calibration(device, index, seconds=1.2):
    """What the driver's UMC-activity percentage reads while the GPU moves a KNOWN number of DRAM bytes: device-to-device
    copies of a 2 GiB buffer (far beyond the 256 MB Infinity Cache: every byte is read from and written to HBM), timed
    with HIP events, and the same copies over a 32 MiB buffer (Infinity-Cache resident: the stacks should idle).  Gives
    percent per TB/s, so that the percentage sampled under the benchmark converts into DRAM bytes without assuming that
    100 % means the 8 TB/s peak."""
    out = {}
    for name, nbytes in (("copy_2GiB", 2 << 30), ("copy_32MiB_cache_resident", 32 << 20)):
        a = torch.empty(nbytes // 4, dtype=torch.float32, device=device).normal_()
        b = torch.empty_like(a)
        b.copy_(a)
        torch.cuda.synchronize()
        per = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        per[0].record()
        for _ in range(4):
            b.copy_(a)
        per[1].record()
        torch.cuda.synchronize()
        reps = int(max(8, seconds / (per[0].elapsed_time(per[1]) * 1e-3 / 4)))
        smp = HbmActivitySampler(index)
        smp.start()
        ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev[0].record()
        for _ in range(reps):
            b.copy_(a)
        ev[1].record()
        torch.cuda.synchronize()
        got = smp.stop()
        rate = 2.0 * nbytes * reps / (ev[0].elapsed_time(ev[1]) * 1e-3)
        out[name] = {"bytes_read_plus_written_per_s": rate, "seconds": ev[0].elapsed_time(ev[1]) * 1e-3,
                     "umc_activity": {k_: v["mean"] for k_, v in got["sources"].items()}}
        del a, b
    torch.cuda.empty_cache()
    return out


watchdog_state = {"done": False, "autotune": None, "meta": None}
_LINE_FD = [None]


def emit(line):
    """The ONE line of this run, on the process's original stdout (libraries that print banners to stdout -- RCCL does
    when a communicator is created -- have been pointed at stderr meanwhile, see main())."""
    data = (json.dumps(line) + "\n").encode()
    if _LINE_FD[0] is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        os.write(_LINE_FD[0], data)


def _arm_watchdog(seconds):
    """N>1 only: if the run makes no progress for `seconds` (no autotune candidate or phase completes), print a line
    from the best COMPLETED autotune candidate (rank 0) and leave."""
    import threading

    def fire():
        if watchdog_state["done"]:
            return
        at, meta = watchdog_state["autotune"], watchdog_state["meta"]
        if meta["rank"] == 0 and at:
            best = min(at, key=at.get)
            emit({
                "metric": "full-batch CAVI iters/sec (48M nnz, k=50)" if meta["workload_key"] == "c3" else
                          "full-batch CAVI iters/sec (%s)" % meta["workload_key"],
                "value": 1e3 / at[best], "unit": "iters/s", "n_gpus": meta["world"], "steps": meta.get("tune_iters", 20), "warmup": 3,
                "ms_per_step": at[best], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                "dtype": "f32", "data": "synthetic",
                "config": {"workload": meta["workload"], "users": meta["users"], "items": meta["items"],
                           "nnz": meta["nnz"], "k": meta["k"], "parallelism": "users sharded x%d, %s" % (meta["world"], best),
                           "exchange_autotune": {"ms_per_iteration": dict(at), "chosen": best},
                           "fallback": "watchdog: a later configuration or phase made no progress; this is the "
                                       "barrier-bracketed %d-iteration measurement of the best completed candidate"
                                       % meta.get("tune_iters", 20)},
                "roofline": None, "cpu_baseline": None})
        os._exit(0 if (at or meta["rank"] != 0) else 3)

    # "no progress": the deadline moves on whenever a candidate or a phase completes (watchdog_progress)
    watchdog_state["deadline"] = time.monotonic() + seconds
    watchdog_state["seconds"] = seconds

    def watch():
        while not watchdog_state["done"]:
            if time.monotonic() > watchdog_state["deadline"]:
                fire()
                return
            time.sleep(2.0)

    t = threading.Thread(target=watch, daemon=True)
    t.start()


_T0 = time.monotonic()


def phase_log(what):
    """HPF_BENCH_VERBOSE=1: elapsed seconds + phase on stderr (every rank) -- where a slow run spends its time."""
    if os.environ.get("HPF_BENCH_VERBOSE") == "1":
        sys.stderr.write("[bench +%7.1f s, rank %s] %s\n" % (time.monotonic() - _T0, os.environ.get("RANK", "0"), what))
        sys.stderr.flush()


def watchdog_progress():
    if "seconds" in watchdog_state:
        watchdog_state["deadline"] = time.monotonic() + watchdog_state["seconds"]


def _launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: become `torch.distributed.run --nproc-per-node N bench.py <same
    arguments>` (one rank per GPU, rendezvous on 127.0.0.1 at a free port).  Rank 0 of the children prints the line on
    the stdout this process was given."""
    import socket
    selftest = os.environ.get("HPF_BENCH_SELFTEST_GLOO") == "1"
    have = torch.cuda.device_count()
    if have < n and not selftest:
        raise SystemExit("bench.py: --gpus %d but this node shows %d GPU(s) (HPF_BENCH_SELFTEST_GLOO=1 runs the N-rank code "
                         "path with all ranks on one GPU: a self-test, not a benchmark)" % (n, have))
    s_ = socket.socket()
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
    s_.close()
    sys.stdout.flush()
    sys.stderr.flush()
    os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
                              "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)]
             + sys.argv[1:])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-events", action="store_true", help="do not bracket launches with HIP events")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the untimed extras (lean-iteration and llk-pass timings); used under rocprofv3 so that "
                         "per-kernel averages cover exactly the warm-up + timed iterations")
    ap.add_argument("--no-workloads", action="store_true",
                    help="skip the untimed `workloads` block (C5 stochastic epochs, C5 partial_fit calls, C4's kernel at k=100, C2) "
                         "of the N=1 line")
    ap.add_argument("--no-fuse", action="store_true", help="separate sweep and row-finalize launches")
    ap.add_argument("--no-autotune", action="store_true",
                    help="N>1: time the library default only")
    ap.add_argument("--autotune-all", action="store_true",
                    help="N>1: also try the one-range / three-range and sweep-grid variants")
    ap.add_argument("--lean", action="store_true",
                    help="skip the stores of the six [n,k] output tables in the timed iterations")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _launch_ranks(args.gpus)        # (does not return: this process becomes the launcher of N ranks of this script)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if os.environ.get("HPF_BENCH_VERBOSE") == "1" and rank == 0:      # where a slow run IS, every 20 s (stderr)
        import faulthandler
        faulthandler.dump_traceback_later(20, repeat=True, file=sys.stderr)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if rank != 0:
        # only rank 0 reports: whatever another rank's libraries leave in stdio at exit (RCCL banners ...) must not
        # trail rank 0's JSON line on the shared stdout
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
    else:
        # stdout carries exactly one line, the JSON: whatever a library prints to fd 1 meanwhile (RCCL's version banner
        # at communicator creation) goes to stderr
        sys.stdout.flush()
        _LINE_FD[0] = os.dup(1)
        os.dup2(2, 1)
    if os.environ.get("HPF_BENCH_SELFTEST_GLOO") == "1":
        local_rank = 0      # code-path self-test on a ONE-GPU box: all ranks on cuda:0, gloo instead of RCCL (not a benchmark)
    if world > 1 or os.environ.get("HPF_FORCE_SHARDED") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29544")
        torch.cuda.set_device(local_rank)
        if os.environ.get("HPF_BENCH_SELFTEST_GLOO") == "1":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        dist = None
        torch.cuda.set_device(0)
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (run `python bench.py --gpus N` and let it launch its own "
                         "ranks, or launch N ranks with torch.distributed.run)" % (args.gpus, world))
    device = torch.device("cuda", local_rank if world > 1 else 0)

    nU, nI, nnz_target, k, label = WORKLOADS[args.workload]
    if dist and world > 1:
        # one generator run, broadcast over xGMI: every rank shards exactly the same matrix
        if rank == 0:
            iu, ii, y = synth_on_device(nU, nI, nnz_target, device, item_power=POWER.get(args.workload, 2.5))
            n_t = torch.tensor([iu.shape[0]], dtype=torch.int64, device=device)
        else:
            n_t = torch.zeros(1, dtype=torch.int64, device=device)
        dist.broadcast(n_t, 0)
        if rank != 0:
            n_all = int(n_t.item())
            iu = torch.empty(n_all, dtype=torch.int64, device=device)
            ii = torch.empty(n_all, dtype=torch.int64, device=device)
            y = torch.empty(n_all, dtype=torch.float32, device=device)
        for t in (iu, ii, y):
            dist.broadcast(t, 0)
    else:
        iu, ii, y = synth_on_device(nU, nI, nnz_target, device, item_power=POWER.get(args.workload, 2.5))
    nnz = int(iu.shape[0])
    phase_log("matrix generated / broadcast: %d nnz" % nnz)

    ops = TimedOps(device)
    hy = cavi.Hyper(k, 0.3, 0.3, 1.0, 0.3, 0.3, 1.0)
    lu, li, ly, (u0, u1) = cavi.shard_users(iu, ii, y, nU, rank, world)
    host_triplets = None
    want_workloads = world == 1 and args.workload == "c3" and not args.no_workloads and not args.no_extras
    if world == 1 and (not args.no_cpu_baseline or want_workloads):   # the CPU baseline times the SAME matrix (host copies: 1 GB at C3)
        host_triplets = (iu.cpu().numpy(), ii.cpu().numpy(), y.cpu().numpy())
    del iu, ii, y
    Theta = np.empty((nU, k), np.float32)
    Beta = np.empty((nI, k), np.float32)
    init = backend.initialize_parameters(Theta, Beta, 123, 0.3, 0.3, 1.0, 0.3, 0.3, 1.0)
    s = slice(u0, u1)

    shared = {}

    def build_model():
        # (every model of the run is over the same triplets: the sparse layouts -- two sorts, the segment lists -- are built
        #  once and shared; a candidate of the exchange autotune then costs its tables and its exchange set-up only)
        m = cavi.FullBatchCavi(ops, device, lu, li, ly, u1 - u0, nI, hy, sides=shared.get("sides"))
        shared.setdefault("sides", m.sides())
        m.load_state(init[0][s], init[1][s], init[2], init[3], init[4][s], init[5], Theta[s], Beta)
        return m

    # N>1: the library default goes FIRST and its measurement is held (watchdog); a few alternatives follow.
    autotune = None
    TUNED = ("HPF_SCHEDULE", "HPF_ITEM_RANGES", "HPF_DIRECT_PREFETCH", "HPF_SHARD_SWEEP_BPC", "HPF_ITEM_SWEEP_BPC",
             "HPF_NATIVE_SHARD")
    if os.environ.get("HPF_BENCH_SELFTEST_GLOO") == "1" and dist is not None:
        # the one-GPU self-test has no RCCL between its ranks: gloo stands in for it behind the C-issued iteration's
        # collective callback (tests/dist_worker.py), so that the RCCL-shaped candidates are exercised too
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from dist_worker import gloo_collective
        dist.native_collective = lambda model_: gloo_collective(dist, model_)
    sharded = dist is not None and (world > 1 or os.environ.get("HPF_FORCE_SHARDED") == "1")
    store = not args.lean

    def joined_barrier(m):
        # the model's exchange stream may still have work in flight: join it before another communicator's barrier
        if m is not None and getattr(m, "dist", None):
            m._sync_scatter_streams()
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()

    def describe(m):
        m._scatter_views()                   # (the exchange is set up on first use: schedule, ranges, plan)
        sched = getattr(m, "schedule", None)
        return "%s/%d%s%s" % (sched, len(m.item_chunks or []),
                              "/no-prefetch" if sched == "direct" and os.environ.get("HPF_DIRECT_PREFETCH", "1") != "1" else "",
                              "" if m._plan is not None else "/python-issued")

    CANDIDATES = [   # (label, environment on top of the defaults) -- the default itself is not in the list
        ("direct, one item range", {"HPF_SCHEDULE": "direct", "HPF_ITEM_RANGES": "1"}),
        ("direct, the apply kernel pulls (no prefetch)", {"HPF_SCHEDULE": "direct", "HPF_DIRECT_PREFETCH": "0"}),
        ("gather-early on RCCL", {"HPF_SCHEDULE": "gather-early"}),
        ("finalize-then-gather on RCCL", {"HPF_SCHEDULE": "finalize-then-gather"}),
        ("direct, user sweep 3 workgroups per CU (room for the exchange stream's kernels)",
         {"HPF_SCHEDULE": "direct", "HPF_SHARD_SWEEP_BPC": "3"}),
    ]
    CANDIDATES_ALL = [
        ("gather-early on RCCL, one item range", {"HPF_SCHEDULE": "gather-early", "HPF_ITEM_RANGES": "1"}),
        ("direct, item sweeps 6 workgroups per CU", {"HPF_SCHEDULE": "direct", "HPF_ITEM_SWEEP_BPC": "6"}),
        ("direct, three item ranges", {"HPF_SCHEDULE": "direct", "HPF_ITEM_RANGES": "3"}),
    ]
    pinned = [v for v in TUNED if v in os.environ]
    model = build_model()
    if sharded and not args.no_autotune and not pinned:
        tune_iters = 6 if os.environ.get("HPF_BENCH_SELFTEST_GLOO") == "1" else 20
        times, failed, labels = {}, {}, {}
        watchdog_state["autotune"], watchdog_state["meta"] = times, dict(
            workload=label, users=nU, items=nI, nnz=nnz, k=k, world=world, rank=rank, workload_key=args.workload,
            tune_iters=tune_iters)
        _arm_watchdog(float(os.environ.get("HPF_BENCH_WATCHDOG_S", "240")))

        def short_run(m):
            m.iterate_many(4, store)
            joined_barrier(m)
            t0 = time.perf_counter()
            m.iterate_many(tune_iters, store)
            joined_barrier(m)
            return (time.perf_counter() - t0) / tune_iters * 1e3

        def agreed(t_ms):
            # every rank takes the same decision: a candidate counts only if it ran on ALL ranks; slowest rank's time
            flag = torch.tensor([0.0 if t_ms is None else 1.0, t_ms or 0.0], dtype=torch.float64, device=device)
            ok = flag[:1].clone()
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            dist.all_reduce(flag[1:], op=dist.ReduceOp.MAX)
            return float(flag[1].item()) if float(ok.item()) > 0 else None

        # 1. the library default, held
        default_key = "default: " + describe(model)
        phase_log("autotune: default model described (%s)" % default_key)
        t = agreed(short_run(model))
        phase_log("autotune: default timed")
        if t is not None:
            times[default_key] = t
        watchdog_progress()
        # 2. the alternatives
        envs = {default_key: {}}
        for lab, env in CANDIDATES + (CANDIDATES_ALL if args.autotune_all else []):
            t_ms, err, m = None, None, None
            os.environ.update(env)
            try:
                phase_log("autotune: building %s" % lab)
                m = build_model()
                phase_log("autotune: built")
                key = describe(m) + ("" if all(v in ("HPF_SCHEDULE", "HPF_ITEM_RANGES", "HPF_DIRECT_PREFETCH") for v in env)
                                     else "/" + ",".join("%s=%s" % kv for kv in sorted(env.items()) if "BPC" in kv[0]))
                if m.schedule != env["HPF_SCHEDULE"]:
                    err = "fell back to %s (%s)" % (m.schedule, m.native_error)
                elif m._scatter_views() is not None and m._plan is None:
                    err = "no C-issued plan (%s)" % (m.native_error or "backend without RCCL")
                else:
                    phase_log("autotune: exchange set up")
                    t_ms = short_run(m)
                    phase_log("autotune: timed")
                    m.flush_items()
            except Exception as exc:   # noqa: BLE001
                key = lab
                err = "%s: %s" % (type(exc).__name__, str(exc)[:200])
            t = agreed(t_ms)
            phase_log("autotune: agreed")
            del m       # (its blocks stay in the caching allocator: the next candidate's model has the same sizes and takes
            #              them over -- returning them to the driver and asking again cost tens of seconds per candidate
            #              when several ranks shared one GPU, HPF_BENCH_VERBOSE=1)
            phase_log("autotune: candidate released")
            for v in env:
                os.environ.pop(v, None)
            labels[key] = lab
            phase_log("autotune: candidate %s done (%s)" % (key, t))
            if t is not None:
                times[key] = t
                envs[key] = env
            else:
                failed[key] = err or "failed on another rank"
            watchdog_progress()
        best = min(times, key=times.get) if times else None
        # only a clear win displaces the default (2 %: run-to-run noise of a 20-iteration measurement)
        if best is not None and best != default_key and default_key in times and times[best] > 0.98 * times[default_key]:
            best = default_key
        autotune = {"ms_per_iteration": dict(times), "chosen": best, "default": default_key, "failed": failed,
                    "candidates": 1 + len(labels), "labels": labels}
        if best is not None and best != default_key:
            model.flush_items()
            del model
            os.environ.update(envs[best])
            model = build_model()
    del lu, li, ly, init, Theta, Beta
    if world == 1:
        torch.cuda.empty_cache()

    if args.no_fuse:
        model.set_fused(False)

    def fence():
        watchdog_progress()
        if dist:
            model._sync_scatter_streams()     # (exchange stream joined before another communicator's barrier)
            dist.barrier()
        torch.cuda.synchronize()

    phase_log("model ready")
    model.iterate_many(max(args.warmup, 1), store)   # untimed: code-object load, first touch
    for _ in range(12):     # N>1: the checked first iterations of the C-issued schedule (and of its fall-backs) stay untimed
        if not (sharded and getattr(model, "_plan", None) is not None and model._needs_first_check()):
            break
        model.iterate_many(1, store)
    fence()
    # N=1: every launch of the timed region is bracketed with HIP events (roofline.achieved comes from them).
    # N>1: an iteration is ~10 short launches, and two event records per launch cost ~9 % of it (measured with
    # tools/shard_probe.py), so the event-bracketed iterations are a separate pass right after the timed region.
    events_in_timed = (world == 1)
    ops.recording = events_in_timed and not args.no_events
    t0 = time.perf_counter()
    if events_in_timed:
        for _ in range(args.steps):
            model.iterate(store)
    else:
        model.iterate_many(args.steps, store)
    fence()
    dt = time.perf_counter() - t0
    phase_log("timed region done")
    ops.recording = False
    ev_steps = args.steps
    if not events_in_timed and not args.no_events:
        # (per-kernel HIP events bracket launches made through the Python ops: this pass issues the same schedule call
        # by call even when the timed region above was issued from C)
        ev_steps = min(args.steps, 10)
        ops.recording = "all"
        model.native = False
        for _ in range(ev_steps):
            model.iterate(store)
        fence()
        model.native = True
        ops.recording = False
    if dist:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # the same iterations once more over a FIXED >= 2 s window (never part of `value`): a --steps 20 timed region lasts
    # 0.07 s at C3 -- too short for any outside observer (a busy sampler, a wall clock around the process) to corroborate
    long_ms = long_steps = None
    sampled = None
    if not args.no_extras:
        long_steps = int(min(50_000, max(args.steps, np.ceil(2000.0 / (dt / args.steps * 1e3)))))
        held = (ops.events, ops.recording)
        ops.events, ops.recording = {}, (events_in_timed and not args.no_events)
        sampler = HbmActivitySampler(local_rank if world > 1 else 0) if rank == 0 else None
        fence()
        if sampler:
            sampler.start()
        t_l = time.perf_counter()
        if events_in_timed:
            for _ in range(long_steps):
                model.iterate(store)
        else:
            model.iterate_many(long_steps, store)
        fence()
        long_dt = time.perf_counter() - t_l
        if sampler:
            sampled = sampler.stop()
            if world == 1 and sampled["sources"]:
                sampled["calibration"] = umc_calibration(device, local_rank if world > 1 else 0)
        ops.events, ops.recording = held
        if dist:
            t = torch.tensor([long_dt], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            long_dt = float(t.item())
        long_ms = long_dt / long_steps * 1e3

    # what fit_hpf actually does between checks: the six [n,k] output tables (shapes, rates, Theta/Beta) are
    # not written.  Reported as an extra field; `value` above is the conservative all-tables-stored figure.
    # per-kernel breakdown of an iteration (every launch bracketed), outside the timed region
    breakdown = None
    if events_in_timed and not args.no_events and not args.no_extras:
        timed_events, ops.events, ops.recording = ops.events, {}, "all"
        for _ in range(4):
            model.iterate(store)
        fence()
        ops.recording = False
        breakdown = {n: v["total_ms"] / 4 for n, v in ops.summary().items()}
        ops.events = timed_events
    lean_ms = None
    if store and not args.no_extras:
        for _ in range(2):
            model.iterate(False)
        fence()
        t2 = time.perf_counter()
        for _ in range(args.steps):
            model.iterate(False)
        fence()
        lean_ms = (time.perf_counter() - t2) / args.steps * 1e3

    # the train-llk evaluation of the reference's default check_every=10, timed separately (never part of `value`)
    llk_ms = llk_val = None
    if not args.no_extras:
        model.iterate(True)             # the lean iterations above left Theta/Beta stale: one storing iteration first
        model.llk_terms(False)          # warm
        fence()
        t1 = time.perf_counter()
        for _ in range(3):
            terms = model.llk_terms(False)
            sub = model.colsum_dot()
        fence()
        llk_ms = (time.perf_counter() - t1) * 1e3 / 3
        llk_val = float(terms[0] - sub)

    # the same iteration with the two column sums of an iteration in the reference's (numpy's) own order: untimed extra
    ref_ms = ref_steps = None
    if world == 1 and not args.no_extras and hasattr(model, "ref_sums") and not model.ref_sums:
        model.ref_sums = True
        for _ in range(2):
            model.iterate(True)
        fence()
        ref_steps = max(5, min(args.steps, 20))
        t3 = time.perf_counter()
        for _ in range(ref_steps):
            model.iterate(True)
        fence()
        ref_ms = (time.perf_counter() - t3) / ref_steps * 1e3
        model.ref_sums = False
        model.iterate(True)

    phase_log("extras done")
    # sanity: the state must be finite after the run (a NaN run would be a meaningless number)
    model.flush_items()   # sharded runs: gather the item tables (each rank finalizes a slice of the items)
    finite = bool(torch.isfinite(model.Beta).all().item() and torch.isfinite(model.Theta).all().item())

    # N>1: what the exchange costs on its own and how much of it the iteration failed to hide (last: it spoils the state)
    collective = None
    links = None
    if sharded and world > 1:
        # what the links give the primitives of the exchange, measured with this job's ranks -- first, and on its own, so
        # that a run whose schedule fell back (or whose exchange report fails) still returns link data
        watchdog_progress()
        try:
            from hpfrec_amd import p2p as _p2p
            links = _p2p.link_probe(device, dist, rank, world)
        except Exception as exc:   # noqa: BLE001
            links = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}
        phase_log("link probe done")
        try:
            links["library_collectives"] = library_collective_probe(dist, world, device, nI * k)
        except Exception as exc:   # noqa: BLE001
            links["library_collectives"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}
        watchdog_progress()
    if sharded:
        phase_log("library collectives probed")
        try:
            collective = exchange_report(model, dist, world, device, dt / args.steps * 1e3, store, fence)
            if "ranks" in collective:       # the exchange really spans the job: N ranks, as the carrier itself reports
                collective["ranks_equal_n_gpus"] = bool(collective["ranks"] == world)
        except Exception as exc:   # noqa: BLE001  (the line must survive)
            collective = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}
        if links is not None:
            collective["link_probe"] = links

    phase_log("exchange report done")
    if rank == 0:
        ms = dt / args.steps * 1e3
        ksum = ops.summary()
        # algorithmic bytes, SURVEY.md section 8d: whole iteration, and per sweep launch
        b_iter = nnz * (8 + 8 * k) + nU * (12 + 20 * k) + nI * (4 + 24 * k)
        n_loc = model.nnz
        roof = None
        # dominant kernel: the sweep.  Fused mode: sweep_kernel<..,FUSE=true> also finishes the rows it
        # swept, so one launch owns half an iteration's algorithmic bytes (nnz term + that side's row
        # term).  Unfused: nnz*(4+4k) + rows*(8+4k) (segment descriptor + own row) per launch.
        dom = "sweep_finalize" if "sweep_finalize" in ksum else ("sweep" if "sweep" in ksum else None)
        if dom:
            if dom == "sweep_finalize":
                # both sides run sweep_kernel with the row finalizer fused in (sharded: the item side in ranges,
                # as prologue): one user-side launch + the item-side launches own this rank's iteration bytes
                # (scatter mode: this rank finalizes only its 1/N slice of the item rows)
                ni_rank = model.nI / world
                b_rank = n_loc * (8 + 8 * k) + model.nU * (12 + 20 * k) + ni_rank * (4 + 24 * k)
                fused = [ksum[n] for n in ("sweep_finalize", "sweep_prefinalize", "sweep") if n in ksum]
                t_both = sum(v["total_ms"] for v in fused) / ev_steps * 1e-3     # per iteration, both sides
                b_launch = b_rank / 2.0
                t_k = t_both / 2.0
            else:
                b_launch = n_loc * (4 + 4 * k) + ((model.nU + model.nI) / 2.0) * (8 + 4 * k)
                t_k = ksum[dom]["avg_ms"] * 1e-3
            ach = b_launch / t_k
            traffic, traffic_src = _pmc_traffic(args.workload, world)
            roof = {"bound": "hbm", "kernel": "sweep_kernel (%s)" % ("fused with row finalize" if dom == "sweep_finalize"
                                                                     else "unfused"),
                    "achieved": ach / 1e9, "peak": HBM_PEAK / 1e9,
                    "unit": "GB/s", "frac": ach / HBM_PEAK, "traffic": traffic, "traffic_source": traffic_src,
                    "algorithmic_bytes_per_launch": b_launch, "avg_launch_ms": t_k * 1e3,
                    "launches": ksum[dom]["calls"],
                    "iteration": {"algorithmic_bytes": b_iter,
                                  "frac_of_hbm_peak": b_iter / (ms * 1e-3) / HBM_PEAK if world == 1 else None},
                    "kernels_ms_per_step": breakdown if breakdown is not None else
                    {n: v["total_ms"] / ev_steps for n, v in ksum.items()},
                    "events": ("timed region (dominant kernel; the per-kernel breakdown is a separate pass of 4 "
                               "iterations)") if events_in_timed
                    else "separate pass of %d iterations after the timed region" % ev_steps}
            if world == 1 and dom == "sweep_finalize" and len(ops.events.get("sweep_finalize", [])) >= 2:
                # the two launches of an iteration apart: even = user side (CSR rows, gathers from the item table),
                # odd = item side (CSC rows, gathers from the user table)
                ev = ops.events["sweep_finalize"]
                side = lambda o: float(np.mean([a.elapsed_time(b) for a, b in ev[o::2]]))   # noqa: E731
                bu = n_loc * (4 + 4 * k) + model.nU * (12 + 20 * k)
                bi = n_loc * (4 + 4 * k) + model.nI * (4 + 24 * k)
                roof["per_side"] = {"user_launch_ms": side(0), "item_launch_ms": side(1),
                                    "user_frac_of_hbm_peak": bu / (side(0) * 1e-3) / HBM_PEAK,
                                    "item_frac_of_hbm_peak": bi / (side(1) * 1e-3) / HBM_PEAK}
            if traffic and _PMC_EXTRA:
                roof.update(_PMC_EXTRA)      # hbm_bytes (DRAM-destined share of `traffic`), mall_hit_rate (not measurable)
            if traffic:
                # the PMC bytes over the live launch duration.  FETCH_SIZE / WRITE_SIZE count at the L2 <-> fabric
                # boundary, so Infinity-Cache hits (the 97 MB item table is MALL-resident) are included: this is a
                # FABRIC-side rate, an upper bound on what HBM itself moved -- not to be read as an HBM utilisation
                roof["traffic_rate"] = {"fabric_side_GB/s": traffic / t_k / 1e9,
                                        "fabric_side_over_hbm_peak": traffic / t_k / HBM_PEAK,
                                        "over_algorithmic": traffic / b_launch,
                                        "counts": "L2<->fabric requests (includes Infinity-Cache hits); not HBM-only"}
            if sampled is not None:
                pref = [n for n in ("gpu_metrics_average_umc_activity", "amd_smi_umc_activity", "sysfs_mem_busy_percent")
                        if n in sampled["sources"]]
                hs = {"window": "the long confirmation pass (%d iterations, %.1f s), sampled on rank 0's GPU"
                                % (long_steps, long_ms * long_steps * 1e-3), **sampled}
                cal = (sampled.get("calibration") or {}).get("copy_2GiB")
                if pref:
                    pct = sampled["sources"][pref[0]]["mean"]
                    hs["source"] = pref[0]
                    hs["umc_activity_percent_mean"] = pct
                    cal_pct = cal["umc_activity"].get(pref[0]) if cal else None
                    if cal_pct:
                        # percent -> bytes through the copy whose DRAM traffic is known (read + written bytes per second)
                        per_pct = cal["bytes_read_plus_written_per_s"] / cal_pct
                        hs["dram_bytes_per_s_per_percent"] = per_pct
                        hs["implied_dram_GBps"] = pct * per_pct / 1e9
                        hs["implied_dram_over_hbm_peak"] = pct * per_pct / HBM_PEAK
                        hs["implied_dram_bytes_per_iteration"] = pct * per_pct * long_ms * 1e-3
                        hs["implied_dram_over_algorithmic_bytes"] = hs["implied_dram_bytes_per_iteration"] / b_iter
                        if traffic and world == 1:
                            fabric_iter = 2.0 * traffic         # two sweep launches carry an iteration's traffic
                            hs["fabric_side_bytes_per_iteration"] = fabric_iter
                            hs["implied_infinity_cache_hit_share"] = 1.0 - hs["implied_dram_bytes_per_iteration"] / fabric_iter
                        hs["reading"] = ("UMC activity = the driver's memory-controller busy percentage (gpu_metrics / "
                                         "mem_busy_percent, 1 % resolution); converted to bytes with the percentage the SAME "
                                         "counter shows under a 2 GiB device-to-device copy whose DRAM traffic is known "
                                         "(calibration.copy_2GiB; the 32 MiB copy shows what an Infinity-Cache-resident stream "
                                         "reads) -- linear in between is an assumption; implied hit share = 1 - DRAM bytes / "
                                         "fabric-side bytes (FETCH_SIZE + WRITE_SIZE of the committed PMC passes)")
                    else:
                        hs["reading"] = "UMC activity sampled, but the calibration copy gave no reading: percentage only"
                else:
                    hs["source"] = None
                    hs["reading"] = "no UMC / HBM activity source answered on this box (see `unavailable`)"
                roof["hbm_util_sampled"] = hs
            if roof["frac"] > 1.0:
                roof["note"] = ("algorithmic bytes exceed what HBM delivers: at this size the gathered tables stay in "
                                "L2 / Infinity Cache (each gather is still counted at face value, SURVEY.md section 8d)")
        watchdog_state["done"] = True
        line = {
            "metric": "full-batch CAVI iters/sec (48M nnz, k=50)" if args.workload == "c3" else
                      "full-batch CAVI iters/sec (%s)" % args.workload,
            "value": args.steps / dt, "unit": "iters/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": label, "users": nU, "items": nI, "nnz": nnz, "k": k, "ld": model.ld,
                       "parallelism": SCHEDULE_TEXT.get(getattr(model, "schedule", None), "%s") % (
                           world, len(model.item_chunks or [])) if sharded else "1 GPU",
                       "schedule": getattr(model, "schedule", None) if sharded else None,
                       "exchange_autotune": autotune,
                       "iteration_issued_by": ("one C call (hpf_hip_shard_iterate)" if getattr(model, "_plan", None)
                                               is not None else "python, call by call") if sharded else None,
                       "native_plan_error": getattr(model, "native_error", None) if sharded else None,
                       "first_iteration_check": getattr(model, "first_check", None) if sharded else None,
                       "first_iteration_checks_failed": getattr(model, "first_checks_failed", None) if sharded else None,
                       # per schedule, for the whole process (autotune candidates included): C-issued iterations that were
                       # compared with the call-by-call form on the same state and agreed / schedules struck by that check
                       "checked_iterations_passed": _shard_checks()[0] if sharded else None,
                       "schedules_struck_by_the_check": _shard_checks()[1] if sharded else None,
                       "seg_cap": cavi.layout.SEG_CAP, "fused_finalize": model.fused and world == 1,
                       "stores_all_state_tables": store, "state_finite": finite},
            "roofline": roof,
        }
        if sharded:
            line["collective"] = collective
        if long_ms is not None:
            line["ms_per_step_long"] = long_ms
            line["steps_long"] = long_steps
            line["ms_per_step_long_over_ms_per_step"] = long_ms / ms
        if lean_ms is not None:
            line["ms_per_step_without_output_table_stores"] = lean_ms
        if llk_ms is not None:
            line["llk_pass_ms"] = llk_ms
            line["iters_per_sec_incl_llk_every_10"] = 1e3 / (ms + llk_ms / 10.0)
            line["train_llk_after_run"] = llk_val
            if world == 1:      # the train-llk pass against the same roofline (nnz*(8+4k) + nU*4k bytes)
                b_llk = nnz * (8 + 4 * k) + nU * 4 * k
                line["llk_pass"] = {"ms": llk_ms, "algorithmic_bytes": b_llk, "frac_of_hbm_peak": b_llk / (llk_ms * 1e-3) / HBM_PEAK,
                                    "kernel": "llk_sweep_kernel (PXI:627-658 over the CSR layout) + the colsum dot"}
        if ref_ms is not None:
            line["reference_order_column_sums"] = {
                "switch": "HPF_COLSUM_ORDER=reference", "ms_per_step": ref_ms, "iters_per_s": 1e3 / ref_ms, "steps": ref_steps,
                "over_default_ms_per_step": ref_ms / ms,
                "what": "the same iteration with Theta.sum(axis=0) / Beta.sum(axis=0) formed in numpy's own order (float32, "
                        "row after row: PXI:236,255) on the device -- the mode that holds 1e-4 against the reference itself at "
                        "every size and horizon (tests/test_hip_parity.py::test_large_vs_golden)"}
        if want_workloads:
            try:
                line["workloads"] = other_workloads(device, host_triplets, nU, nI, with_oracle=not args.no_cpu_baseline)
            except Exception as e:   # noqa: BLE001  (the bench line must survive)
                line["workloads"] = {"error": repr(e)[:300]}
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(nU, nI, k, nnz, device=device, triplets=host_triplets)
            except Exception as e:  # the bench line must survive a broken host toolchain
                line["cpu_baseline"] = {"value": None, "error": repr(e)}
        else:
            line["cpu_baseline"] = None
    else:
        line = None
    if dist:
        dist.destroy_process_group()
    if line is not None:
        # the JSON line must be the LAST line of stdout: flush whatever RCCL/HIP left in C stdio first
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        emit(line)


def exchange_report(model, dist, world, device, ms_per_step, store, fence, reps=10):
    """N>1: the `collective` block of the line.  (1) The exchange ALONE, with nothing else on the GPU, HIP events around
    each group, slowest rank reported: RCCL schedules -- the iteration's reduce-scatters, all-gathers and k-float
    all-reduces back to back; direct schedule -- the kernels that carry it: the pulling shape halves, the pull of the
    finished rows, the granule all-reduce.  Bytes per rank, ms, algorithm and bus GB/s (nccl-tests' convention: bytes *
    (n-1)/n / t; for the pulls that IS the traffic on the links).  (2) The compute ALONE: the same iterations with the
    exchange emulated locally (the plan's dry run: this rank alone), barrier-bracketed -> exposed_ms = iteration -
    compute_only: what the schedule failed to hide.  Needs the C-issued plan; the state is garbage afterwards."""
    import ctypes
    from hpfrec_amd import p2p, shard_native as sn
    model._scatter_views()
    plan = getattr(model, "_plan", None)
    if plan is None:
        return {"skipped": "no C-issued plan on this configuration (%s)" % getattr(model, "native_error", None)}
    k, ld, W = model.k, model.ld, world
    views = model._chunk_views
    rows = sum(c["hi"] - c["lo"] for c in views)
    e_ld = int(model.e_own_all.shape[1])
    direct = model.schedule == "direct"
    if direct:
        ranks, ranks_src = model._region.world, "ranks of the peer-mapped region (every one of them mapped by this rank)"
    elif model.comm is not None:
        ranks, ranks_src = model.comm.count(), "ncclCommCount of the iteration's communicator"
    else:
        ranks, ranks_src = W, "torch.distributed world size (no RCCL communicator: gloo callback)"
    cur = lambda: torch.cuda.current_stream(device).cuda_stream   # noqa: E731
    fence()

    def timed(fn):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        fn()                                   # warm
        fence()
        ev[0].record()
        for _ in range(reps):
            fn()
        ev[1].record()
        torch.cuda.synchronize()
        t = torch.tensor([ev[0].elapsed_time(ev[1]) / reps], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    rs_ms = timed(lambda: [plan.exchange_only(sn.COLL_REDUCE_SCATTER, j, cur()) for j in range(len(views))])
    ag_ms = timed(lambda: [plan.exchange_only(sn.COLL_ALL_GATHER, j, cur()) for j in range(len(views))])
    ar_ms = timed(lambda: [plan.exchange_only(sn.COLL_ALL_REDUCE, -1, cur()) for _ in range(2)])
    if direct:
        plan.status()
    rs_bytes, ag_bytes = rows * k * 4, rows * e_ld * 4           # the whole buffer an exchange step spans, per rank
    bus = (W - 1) / W

    # compute only: a dry-run twin of the plan over the same tensors (direct: over a region connected to itself)
    d = sn.ShardDesc()
    ctypes.memmove(ctypes.byref(d), ctypes.byref(plan.desc), ctypes.sizeof(d))
    d.dry_run, d.comm, d.comm_small, d.coll_ctx = 1, None, None, None
    d.coll = sn.COLLECTIVE_FN()
    keep = list(plan.keep)
    if direct:
        twin = p2p.PeerRegion(device, model._region.data_bytes, ld, rank=model.rank, world=W, local=True)
        d.p2p_region = twin.handle.value
        d.acc_i = twin.data_ptr() + d.p2p_acc_offset
        d.e_own = twin.data_ptr() + d.p2p_send_offset
        keep.append(twin)
    dry = sn.ShardPlan(d, keep=keep)
    fence()
    its = 20
    stream = cur()

    def run(n):
        for _ in range(n):
            dry.iterate(model.eT, model.eT_next, store, stream)
            model.eT, model.eT_next = model.eT_next, model.eT
    run(3)
    dry.join(stream)
    fence()
    t0 = time.perf_counter()
    run(its)
    dry.join(stream)
    fence()
    comp = torch.tensor([(time.perf_counter() - t0) / its * 1e3], dtype=torch.float64, device=device)
    dist.all_reduce(comp, op=dist.ReduceOp.MAX)
    comp_ms = float(comp.item())
    dry.close()
    return {"ranks": ranks, "ranks_source": ranks_src, "ranges": len(views),
            "carried_by": "kernels of the iteration pulling peer-mapped memory (no collective library)" if direct
            else "RCCL collectives on a communicator of our own" if model.comm is not None else "gloo behind the plan's callback",
            "bytes_per_rank": {"reduce_scatter_buffer": rs_bytes, "all_gather_buffer": ag_bytes,
                               "sent_and_received_per_rank": (rs_bytes + ag_bytes) * bus, "small_all_reduces": 2 * ld * 4},
            "schedule": model.schedule,
            "rs_ms": rs_ms, "ag_ms": ag_ms, "small_allreduce_ms_each": ar_ms / 2,
            "algbw_GBps": {"reduce_scatter": rs_bytes / rs_ms / 1e6, "all_gather": ag_bytes / ag_ms / 1e6},
            "busbw_GBps": {"reduce_scatter": rs_bytes * bus / rs_ms / 1e6, "all_gather": ag_bytes * bus / ag_ms / 1e6},
            "exchange_alone_ms": rs_ms + ag_ms + ar_ms, "compute_only_ms": comp_ms, "iteration_ms": ms_per_step,
            "exposed_ms": ms_per_step - comp_ms,
            "note": "exchange alone: back to back, nothing else running (direct: rs = the pulling shape halves, ag = the "
                    "pull of the finished rows); compute only: the exchange emulated locally (state not meaningful afterwards)"}


def _shard_checks():
    from hpfrec_amd import shard
    return ({k_[0]: n for k_, n in shard._PASSED.items()}, sorted(k_[0] for k_ in shard._FAILED))


def library_collective_probe(dist, world, device, floats, reps=5):
    """torch.distributed's own all-reduce / reduce-scatter / all-gather (RCCL on a GPU job) over a buffer the size of the
    item statistics (nI*k floats), alone on the GPU: ms and bus GB/s (nccl-tests' convention) -- the yardstick the
    exchange kernels of this library are to be read against, and link data that does not depend on peer mapping."""
    n = (floats // (4 * world)) * 4 * world
    buf = torch.ones(n, dtype=torch.float32, device=device)
    part = torch.empty(n // world, dtype=torch.float32, device=device)
    out = {"backend": dist.get_backend(), "bytes": n * 4}

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        dist.barrier()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(reps):
            fn()
        ev[1].record()
        torch.cuda.synchronize()
        t = torch.tensor([ev[0].elapsed_time(ev[1]) / reps], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    bus = (world - 1) / world
    for name, fn, factor in (("all_reduce", lambda: dist.all_reduce(buf), 2 * bus),
                             ("reduce_scatter", lambda: dist.reduce_scatter_tensor(part, buf), bus),
                             ("all_gather", lambda: dist.all_gather_into_tensor(buf, part), bus)):
        ms = timed(fn)
        out[name] = {"ms": ms, "busbw_GBps": n * 4 * factor / ms / 1e6}
        buf.fill_(1.0)
    return out


def kernel_source_sha16():
    """Hash of the kernel source + C header the loaded library was built from (what a PMC summary is tied to)."""
    import hashlib
    from hpfrec_amd import _lib
    h = hashlib.sha256()
    for path in (_lib.SRC_PATH, os.path.join(_lib.INC_PATH, "hpf_hip.h")):
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


_PMC_EXTRA = {}      # hbm_bytes / mall_hit_rate of the committed PMC summary (filled by _pmc_traffic)


def _pmc_traffic(workload, world):
    """(HBM bytes per sweep launch, provenance) from the rocprofv3 --pmc passes committed under profiles/ (collected
    in separate runs, never in the timed one; tools/gpu_profile.sh + tools/pmc_json.py).  The summary names the
    kernel source it was measured on: when that is not the source of the library running now, the number is stale
    and `traffic` is null."""
    rel = os.path.join("profiles", "pmc_%s_n%d.json" % (workload, world))
    p = os.path.join(ROOT, rel)
    if not os.path.exists(p):
        return None, "no PMC summary for this workload (%s)" % rel
    try:
        d = json.load(open(p))
    except Exception as e:   # noqa: BLE001
        return None, "%s unreadable: %r" % (rel, e)
    have, want = d.get("kernel_source_sha16"), kernel_source_sha16()
    if have != want:
        return None, "%s was measured on kernel source %s, this run uses %s: stale" % (rel, have, want)
    _PMC_EXTRA.update({kk: d.get(kk) for kk in ("hbm_bytes", "hbm_bytes_note", "mall_hit_rate", "mall_hit_rate_note",
                                                "read_requests_per_launch", "write_requests_per_launch") if kk in d})
    return d.get("sweep_kernel_hbm_bytes_per_launch"), "%s (%s; kernel source %s)" % (rel, d.get("source"), have)


if __name__ == "__main__":
    main()
