"""Host-side logic without a GPU: layouts, the C-ABI library's exports, the driver loop (run on the
numpy stand-in ops of tests/cpu_ops.py) against the reference's golden vectors."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import datagen
from conftest import GOLDEN, ROOT
from hpfrec_amd import _lib, layout

NAMES = ("Theta", "Beta", "Gamma_shp", "Gamma_rte", "Lambda_shp", "Lambda_rte", "k_rte", "t_rte")


def test_ld_for_k():
    assert [_lib.ld_for_k(k) for k in (1, 30, 32, 33, 50, 64, 100, 200, 1024)] == [32, 32, 32, 64, 64, 64, 128, 256, 1024]
    with pytest.raises(ValueError):
        _lib.ld_for_k(1025)
    with pytest.raises(ValueError):
        _lib.ld_for_k(0)


def test_library_exports_every_declared_symbol():
    so = _lib.build()
    hdr = open(os.path.join(ROOT, "include", "hpf_hip.h")).read()
    declared = set(re.findall(r"\b(hpf_hip_\w+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS)
    L = ctypes.CDLL(so)
    for s in declared:
        assert hasattr(L, s), s
    assert L.hpf_hip_abi_version() == _lib.HPF_HIP_ABI_VERSION
    for k in (1, 30, 50, 100, 200, 1000):
        assert L.hpf_hip_ld_for_k(k) == _lib.ld_for_k(k)
    assert L.hpf_hip_ld_for_k(0) == -1 and L.hpf_hip_ld_for_k(5000) == -2


def test_c_abi_rejects_bad_arguments_before_touching_the_gpu():
    """Error behaviour of the boundary (int return codes, nothing thrown across the ABI):
    null pointers, a leading dimension that does not belong to k, k out of range -- all caught by the argument
    checks, which run before any HIP call (so this needs no GPU)."""
    L = ctypes.CDLL(_lib.build())
    vp, i64, ci, cf = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float
    L.hpf_hip_sweep_f32.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp, vp]
    L.hpf_hip_expect_f32.argtypes = [vp, vp, vp, vp, vp, i64, ci, ci, vp, vp, cf, vp, vp]
    L.hpf_hip_colsum_reduce_f32.argtypes = [vp, ci, vp, ci, vp]
    L.hpf_hip_pair_dot_f32.argtypes = [vp, vp, vp, vp, i64, vp, ci, ci, vp]
    EINVAL, EUNSUPPORTED = -1, -2
    assert L.hpf_hip_sweep_f32(None, 5, None, None, None, None, None, None, 0, 50, 64, 0, 8, None, None) == EINVAL
    assert L.hpf_hip_sweep_f32(None, 0, None, None, None, None, None, None, 0, 50, 64, 0, 8, None, None) == 0  # empty
    buf = ctypes.create_string_buffer(64)
    p = ctypes.cast(buf, vp)
    assert L.hpf_hip_sweep_f32(p, 1, p, p, p, p, p, None, 0, 50, 128, 0, 8, None, None) == EINVAL      # ld is not ld(k)
    assert L.hpf_hip_expect_f32(None, None, None, None, None, 3, 50, 64, None, None, 0.0, None, None) == EINVAL
    assert L.hpf_hip_expect_f32(p, p, p, None, None, 0, 50, 64, None, None, 0.0, None, None) == 0                                    # no rows
    assert L.hpf_hip_expect_f32(p, p, p, None, None, 3, 50, 64, None, None, 0.0, p, None) == EINVAL      # rte_out without a factored rate
    assert L.hpf_hip_colsum_reduce_f32(None, 4, None, 64, None) == EINVAL
    assert L.hpf_hip_pair_dot_f32(p, p, p, p, -1, p, 50, 64, None) == EINVAL
    assert L.hpf_hip_ld_for_k(2000) == EUNSUPPORTED
    # round 5 entries: the epoch-level batch preparation, the sweep with the other side's stochastic step fused in, the
    # whole-table pass's done_flag
    L.hpf_hip_svi_epoch_prepare.argtypes = [vp, vp]
    L.hpf_hip_sweep_svi_f32.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp] + [cf] * 7 + [ci] * 4 + [vp, vp]
    L.hpf_hip_svi_side_f32.argtypes = [i64] + [vp] * 9 + [cf] * 7 + [ci] * 5 + [vp, vp, vp, ci, vp]
    assert L.hpf_hip_svi_epoch_prepare(None, None) == EINVAL
    desc = ctypes.create_string_buffer(512)
    assert L.hpf_hip_svi_epoch_prepare(ctypes.cast(desc, vp), None) == EINVAL                    # all-null descriptor
    w = (0.3, 1.0, 0.0, 15.3, 0.3, 0.5, 0.5)
    assert L.hpf_hip_sweep_svi_f32(None, 5, p, p, p, p, p, None, p, p, None, p, p, p, *w, 50, 64, 0, 8, None, None) == EINVAL
    assert L.hpf_hip_sweep_svi_f32(p, 5, p, p, p, p, p, None, p, p, None, p, p, p, *w, 50, 128, 0, 8, None, None) == EINVAL   # ld
    q = ctypes.cast(ctypes.create_string_buffer(64), vp)
    assert L.hpf_hip_sweep_svi_f32(p, 5, p, p, p, p, p, q, p, p, None, p, p, p, *w, 50, 64, 0, 8, None, None) == EINVAL   # e_new must be tab_self
    side = lambda flag, rs_mode, done: L.hpf_hip_svi_side_f32(4, flag, p, p, p, p, None, p, p, p, *w, 1, rs_mode, 50, 64, 2, None,  # noqa: E731
                                                              None, None, done, None)
    assert side(None, 1, 1) == EINVAL            # done_flag without flags
    assert side(p, 1, 300) == EINVAL
    assert L.hpf_hip_svi_side_f32(4, p, p, p, p, p, None, p, p, p, *w, 1, 1, 50, 64, 2, p, None, None, 1, None) == EINVAL  # done_flag + rs_rate
    # round 6: the sweep with the BATCH side's stochastic step fused in
    L.hpf_hip_sweep_svi_batch_f32.argtypes = [vp, i64] + [vp] * 13 + [cf, vp, vp] + [cf] * 7 + [ci] * 4 + [vp, vp]
    bat = lambda segs, rte_in, rate_rs, rate_cs, ld: L.hpf_hip_sweep_svi_batch_f32(      # noqa: E731
        segs, 5, p, p, p, p, p, p, rte_in, None, None, p, None, rate_rs, rate_cs, 15.3, p, p, *w, 50, ld, 0, 8, None, None)
    assert bat(None, p, None, None, 64) == EINVAL         # no segments
    assert bat(p, None, None, None, 64) == EINVAL         # neither a factored rate nor a rate table for the prologue
    assert bat(p, None, p, None, 64) == EINVAL            # factored rate without its column sums
    assert bat(p, p, None, None, 128) == EINVAL           # ld is not ld(k)


def test_integration_doc_stub_matches_the_abi():
    """The ctypes stub shown in INTEGRATION.md must stay in step with include/hpf_hip.h."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"lib\.hpf_hip_sweep_f32\.argtypes = \[(.*?)\]", doc)
    hdr = open(os.path.join(ROOT, "include", "hpf_hip.h")).read()
    decl = re.search(r"int hpf_hip_sweep_f32\((.*?)\);", hdr, re.S).group(1)
    assert len(m.group(1).split(",")) == len(decl.split(","))
    L = _lib.lib() if torch.cuda.is_available() else None
    if L is not None:
        assert len(L.hpf_hip_sweep_f32.argtypes) == len(decl.split(","))


def test_missing_gpu_fails_loudly():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from hpfrec_amd import cython_loops_float as be
    from hpfrec_amd.ops_hip import HipOps
    assert be.HipOps is HipOps
    with pytest.raises(_lib.HpfHipError):
        be.predict_arr(np.ones((2, 3), np.float32), np.ones((2, 3), np.float32), np.zeros(1, np.uint64),
                       np.zeros(1, np.uint64), 1)


def _segments_cover(side):
    s = side.segs.numpy()
    begin, length, row = s[:, 0], s[:, 1] & layout.SEG_LEN_MASK, s[:, 1] >> 32
    whole = (s[:, 1] & layout.SEG_WHOLE_ROW) != 0
    indptr = side.indptr.numpy()
    cover = np.zeros(side.nnz, dtype=np.int64)
    for b, l, r in zip(begin, length, row):
        assert l > 0 and indptr[r] <= b and b + l <= indptr[r + 1]
        cover[b:b + l] += 1
    assert (cover == 1).all()
    rsp = side.row_seg_ptr.numpy()
    for r in range(side.nrows):
        assert (row[rsp[r]:rsp[r + 1]] == r).all()
        assert whole[rsp[r]:rsp[r + 1]].all() == (rsp[r + 1] - rsp[r] == 1) or rsp[r + 1] == rsp[r]
    nseg_row = rsp[1:] - rsp[:-1]
    assert np.array_equal(side.multi_rows.numpy(), np.nonzero(nseg_row != 1)[0])


@pytest.mark.parametrize("seg_cap", [4, 256])
def test_layout_keeps_duplicates_and_covers(seg_cap):
    rs = np.random.RandomState(0)
    nU, nI, n = 37, 23, 900
    iu = torch.from_numpy(rs.randint(nU, size=n))
    ii = torch.from_numpy(rs.randint(nI, size=n))
    iu[:5] = 36  # ragged: some rows heavy, some empty
    y = torch.from_numpy(rs.gamma(1, 1, size=n).astype(np.float32) + 1)
    users, items, u_sorted = layout.build_sides(iu, ii, y, nU, nI, seg_cap)
    assert users.nnz == n and items.nnz == n  # duplicates stay separate observations
    for side in (users, items):
        _segments_cover(side)
        assert int(side.segs[:, 1].bitwise_and(layout.SEG_LEN_MASK).max()) <= seg_cap
    # same multiset of (u,i,y) on both sides
    a = sorted(zip(u_sorted.tolist(), users.idx.tolist(), users.y.tolist()))
    rows_i = torch.repeat_interleave(torch.arange(nI), items.indptr[1:] - items.indptr[:-1])
    b = sorted(zip(items.idx.tolist(), rows_i.tolist(), items.y.tolist()))
    c = sorted(zip(iu.tolist(), ii.tolist(), y.tolist()))
    assert a == b == c


def test_layout_empty_and_out_of_range():
    e = torch.zeros(0, dtype=torch.int64)
    users, items, _ = layout.build_sides(e, e, torch.zeros(0), 3, 4)
    assert users.nseg == 0 and items.nseg == 0 and users.row_seg_ptr.tolist() == [0, 0, 0, 0]
    with pytest.raises(ValueError):
        layout.build_sides(torch.tensor([3]), torch.tensor([0]), torch.ones(1), 3, 4)


def test_nnz_balanced_ranges():
    indptr = torch.tensor([0, 10, 10, 11, 50, 51, 100])
    r = layout.nnz_balanced_ranges(indptr, 2)
    assert r[0][0] == 0 and r[-1][1] == 6 and r[0][1] == r[1][0]
    r8 = layout.nnz_balanced_ranges(indptr, 8)
    assert len(r8) == 8 and all(a <= b for a, b in r8) and r8[-1][1] == 6
    assert all(r8[i][1] == r8[i + 1][0] for i in range(7))


def _fit(be, Y, iu, ii, nU, nI, k, its, seed=123, verbose=0, check_every=0, stop_crit="maxiter", **hyper):
    h = dict(a=0.3, a_prime=0.3, b_prime=1.0, c=0.3, c_prime=0.3, d_prime=1.0)
    h.update(hyper)
    Theta = np.empty((nU, k), np.float32)
    Beta = np.empty((nI, k), np.float32)
    i, temp, llk = be.fit_hpf(h["a"], h["a_prime"], h["b_prime"], h["c"], h["c_prime"], h["d_prime"], Y, iu, ii, Theta,
                              Beta, its, stop_crit, check_every, 1e-3, 0, 0, None, 0, np.zeros(1, np.uint64), "", seed,
                              verbose, 1, 0, 0, np.empty(0, np.float32), np.empty(0, np.uint64), np.empty(0, np.uint64),
                              0, 1, 0)
    return i, dict(zip(NAMES, (Theta, Beta) + tuple(temp))), llk


def _maxrel(a, b):
    return float(np.max(np.abs(a - b) / np.abs(b)))


def test_driver_on_standin_matches_golden(cpu_ops_backend, capsys):
    df, nU, nI = datagen.readme_counts()
    Y, iu, ii = datagen.triplets(df)
    g = np.load(os.path.join(GOLDEN, "c1_full.npz"))
    for its, tol in ((1, 5e-6), (5, 2e-5), (10, 5e-5)):
        i, arrs, _ = _fit(cpu_ops_backend, Y, iu, ii, nU, nI, 30, its)
        assert i == its - 1  # PXI:418: 0-based index of the last iteration
        for n in NAMES:
            assert arrs[n].shape == g["it%d_%s" % (its, n)].shape
            assert _maxrel(arrs[n], g["it%d_%s" % (its, n)]) < tol, (its, n)
    i, arrs, llk = _fit(cpu_ops_backend, Y, iu, ii, nU, nI, 30, 10, verbose=1, check_every=10)
    out = capsys.readouterr().out
    assert "Iteration 10 | train llk:" in out and "Optimization finished" in out
    assert abs(float(llk) / g["train_llk_it10"] - 1) < 1e-5


def test_driver_return_contract(cpu_ops_backend):
    df, nU, nI = datagen.readme_counts()
    Y, iu, ii = datagen.triplets(df)
    # verbose=0 + maxiter: no llk at all (PXI:99-113)
    i, arrs, llk = _fit(cpu_ops_backend, Y, iu, ii, nU, nI, 30, 3)
    assert llk is None and i == 2
    assert arrs["Gamma_rte"].shape == (nU, 30) and arrs["k_rte"].shape == (nU, 1) and arrs["t_rte"].shape == (nI, 1)
    g = np.load(os.path.join(GOLDEN, "c1_full.npz"))
    # stopping rules stop at the iteration the reference stops at
    i, _, _ = _fit(cpu_ops_backend, Y, iu, ii, nU, nI, 30, 200, stop_crit="train-llk", check_every=5)
    assert i == int(g["trainllk_stop_niter"])


def test_ctpfrec_exports_vs_reference(any_backend, capsys):
    """The module-level helpers the reference exports "for ctpfrec" (PXI:20-113, 830-847) with the reference's
    arity, against values the real extension module returned (tests/golden/c1_boundary.npz): train / validation
    llk and RMSE (errs[0], errs[1]; PXI:66-79), the val-set expression of eval_after_term (PXI:105), the stopping
    rule, the diff-norm branch, get_csc_data with duplicate pairs, get_unique_items_batch."""
    import inspect
    import pandas as pd
    from hpfrec_amd import HPF
    be = any_backend
    g = np.load(os.path.join(GOLDEN, "c1_boundary.npz"))
    g10 = np.load(os.path.join(GOLDEN, "c1_full.npz"))
    for name, arity in (("assess_convergence", 22), ("eval_after_term", 17), ("get_csc_data", 5),
                        ("get_unique_items_batch", 5), ("print_norm_diff", 3), ("print_llk_iter", 4),
                        ("print_final_msg", 4), ("save_parameters", 4)):
        assert len(inspect.signature(getattr(be, name)).parameters) == arity, name
    df, nU, nI = datagen.readme_counts()
    Y, iu, ii = datagen.triplets(df)
    Yv, iuv, iiv = datagen.boundary_valset(nU, nI)
    Theta, Beta, k = g10["it10_Theta"], g10["it10_Beta"], 30
    for tag, full in (("", 0), ("_full", 1)):
        for has_val, which in ((0, "train"), (1, "val")):
            errs = np.zeros(2, dtype=np.longdouble)
            conv, crit = be.assess_convergence(9, 10, "train-llk", -1e300, 1e-3, Theta, Theta.copy(), Beta, Y.shape[0], Y,
                                               iu, ii, Yv.shape[0], Yv, iuv, iiv, errs, k, 1, 1, full, has_val)
            assert not conv and float(crit) == float(errs[0])
            assert abs(float(errs[0]) / g["assess_%s_llk%s" % (which, tag)] - 1) < 2e-6
            assert abs(float(errs[1]) / g["assess_%s_rmse%s" % (which, tag)] - 1) < 2e-6          # errs[1], PXI:73,79
            errs2 = np.zeros(2, dtype=np.longdouble)
            last = be.eval_after_term("maxiter", 1, 1, full, k, Y.shape[0], Yv.shape[0], has_val, Theta, Beta, errs2,
                                      Y, iu, ii, Yv, iuv, iiv)
            assert abs(float(last) / g["after_term_%s_llk%s" % (which, tag)] - 1) < 2e-6          # PXI:105 for "val"
            assert abs(float(errs2[1]) / g["after_term_%s_rmse%s" % (which, tag)] - 1) < 2e-6
    assert be.eval_after_term("train-llk", 1, 1, 0, k, Y.shape[0], 0, 0, Theta, Beta, np.zeros(2, np.longdouble), Y, iu,
                              ii, Yv, iuv, iiv) is None
    out = capsys.readouterr().out
    assert "Iteration 10 | train llk: -9871 | train rmse: 1.0955" in out and "val llk: -408 | val rmse: 1.2016" in out
    # not verbose: no squared-error accumulation (add_mse = verbose), nothing printed
    errs = np.zeros(2, dtype=np.longdouble)
    conv, _ = be.assess_convergence(19, 10, "train-llk", float(g["assess_train_llk"]) * (1 - 5e-4), 1e-3, Theta,
                                    Theta.copy(), Beta, Y.shape[0], Y, iu, ii, Yv.shape[0], Yv, iuv, iiv, errs, k, 1, 0, 0, 0)
    assert bool(conv) == bool(g["assess_second_check_converged"]) and errs[1] == 0 and capsys.readouterr().out == ""
    Tp = (Theta * np.float32(1.01)).astype(np.float32)
    conv, crit = be.assess_convergence(9, 10, "diff-norm", -1e300, 1e-9, Theta, Tp, Beta, Y.shape[0], Y, iu, ii, 0, Yv,
                                       iuv, iiv, np.zeros(2, np.longdouble), k, 1, 0, 0, 0)
    assert not conv and abs(crit / g["assess_diffnorm"] - 1) < 1e-5 and np.array_equal(Tp, Theta)
    conv, _ = be.assess_convergence(9, 10, "diff-norm", -1e300, 1e-3, Theta, Theta.copy(), Beta, Y.shape[0], Y, iu, ii, 0,
                                    Yv, iuv, iiv, np.zeros(2, np.longdouble), k, 1, 0, 0, 0)
    assert conv
    # CSC conversion: duplicates merged, rows ascending; the batch helper
    rs = np.random.RandomState(1)
    du, di = rs.randint(nU, size=3000).astype(np.uint64), rs.randint(40, size=3000).astype(np.uint64)
    dy = (rs.gamma(1, 1, size=3000) + 1).astype(np.float32)
    ptr, ind, dat = be.get_csc_data(du, di, dy, nU, nI)
    for got, want in ((ptr, g["csc_indptr"]), (ind, g["csc_indices"])):
        assert got.dtype == want.dtype and np.array_equal(got, want)
    assert dat.dtype == g["csc_data"].dtype and np.allclose(dat, g["csc_data"], rtol=1e-6, atol=0)   # (sum order of triples)
    Ys, ius, iis, st = datagen.sorted_by_user(Y, iu, ii, nU, nI)
    users_b = np.array([5, 17, 3, 99, 42], dtype=np.uint64)
    items, st_pos = be.get_unique_items_batch(users_b, st, iis, 1, True)
    assert np.array_equal(items, g["batch_items"]) and np.array_equal(st_pos, g["batch_st_pos"])
    assert items.dtype == g["batch_items"].dtype and st_pos.dtype == g["batch_st_pos"].dtype
    assert np.array_equal(be.get_unique_items_batch(users_b, st, iis, 1, False), g["batch_items_only"])
    # the class with a validation set: where 'val-llk' stops, and the final expression of a verbose 'maxiter' run
    vdf = pd.DataFrame({"UserId": iuv.astype(np.int64), "ItemId": iiv.astype(np.int64), "Count": Yv})
    mv = HPF(k=30, maxiter=200, random_seed=123, reindex=False, verbose=False, stop_crit="val-llk", check_every=5,
             stop_thr=1e-3).fit(df.copy(), val_set=vdf.copy())
    assert mv.niter == int(g["valllk_stop_niter"])
    mv = HPF(k=30, maxiter=10, random_seed=123, reindex=False, verbose=True, stop_crit="maxiter", check_every=10)
    mv.fit(df.copy(), val_set=vdf.copy())
    assert abs(float(mv.train_llk) / g["maxiter_valset_last_llk"] - 1) < 1e-4
    out = capsys.readouterr().out
    assert "Final RMSE: %.4f" % g["after_term_val_rmse"] in out


@pytest.mark.parametrize("dtype,bad", [(np.uint64, 100), (np.uint64, 2 ** 63 + 5), (np.int64, -1), (np.int32, 100)])
def test_fit_rejects_ids_outside_the_tables(cpu_ops_backend, dtype, bad):
    """The reference indexes without bounds checks (PXI:547-550: UB); here an id outside [0, n) -- also one that only
    looks valid after the 64-bit reinterpretation of size_t ids -- is an error before anything is launched, for every
    id dtype, and valid ids of every dtype give the same fit."""
    df, nU, nI = datagen.readme_counts()
    Y, iu, ii = datagen.triplets(df)
    base = _fit(cpu_ops_backend, Y, iu, ii, nU, nI, 8, 2)[1]
    same = _fit(cpu_ops_backend, Y, iu.astype(np.int64).astype(dtype), ii.astype(np.int64).astype(dtype), nU, nI, 8, 2)[1]
    assert all(np.array_equal(base[n], same[n]) for n in NAMES)
    for side in (0, 1):
        ids = [iu.astype(np.int64).astype(dtype), ii.astype(np.int64).astype(dtype)]
        ids[side] = ids[side].copy()
        ids[side][17] = np.array(bad).astype(dtype)
        with pytest.raises(ValueError):
            _fit(cpu_ops_backend, Y, ids[0], ids[1], nU, nI, 8, 2)


def test_rccl_binding_and_unique_id_bytes():
    """libhpf_hip.so binds RCCL's entry points at run time from the librccl.so PyTorch ships (dlsym, nothing linked),
    rejects bad arguments before touching RCCL, and the ncclUniqueId rank 0 hands to the others is carried as 128 BINARY
    bytes: zeros inside it survive the trip through a byte tensor and into the C array the init call reads."""
    import ctypes
    from hpfrec_amd import rccl
    L = rccl.open_rccl()
    assert L.hpf_hip_rccl_open(None) == 0                         # idempotent once bound
    assert L.hpf_hip_rccl_comm_init(None, 1, 0, None) == -1       # HPF_EINVAL
    assert L.hpf_hip_rccl_comm_count(None, None) == -1
    assert L.hpf_hip_rccl_all_reduce_f32(None, None, 4, None) == -1
    raw = bytes([7, 0, 0, 9] + [0] * 60 + list(range(64)))
    buf = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
    assert buf.numel() == rccl.UID_BYTES and buf.numpy().tobytes() == raw
    uid = (ctypes.c_uint8 * rccl.UID_BYTES).from_buffer_copy(buf.numpy().tobytes())
    assert bytes(uid) == raw


def _mt_key_after(bg, nwords):
    """numpy's MT19937 key after `nwords` more draws (nwords a multiple of 624, generator at a block boundary)."""
    left = nwords
    while left > 0:
        step = min(left, 624 * 4096)
        bg.random_raw(step)
        left -= step
    st = bg.state["state"]
    assert int(st["pos"]) == 624
    return st["key"].astype(np.uint32)


@pytest.mark.parametrize("q", [0, 3, 5, 9, 12])
def test_mt19937_jump_polynomials_against_numpy_stepping(q):
    """The jump-ahead of the parallel initial draw (hpfrec_amd/csrc/hpf_mt19937.hip): the state 624 * 2^q words ahead is
    the XOR, over the set coefficients i of x^(624 * 2^q) mod phi, of the stream window z[i .. i+624) -- checked against
    numpy's own generator stepped that far (which also pins the table of phi's 135 exponents: phi has degree 19937 and
    q >= 5 needs the reduction).  Host only: the polynomial comes from the library, the XOR is numpy."""
    import ctypes
    from hpfrec_amd import _lib
    L = _lib.lib()
    poly = np.zeros(624, dtype=np.uint32)
    assert L.hpf_hip_mt19937_jump_poly(q, poly.ctypes.data_as(ctypes.c_void_p)) == 0
    assert L.hpf_hip_mt19937_jump_poly(99, poly.ctypes.data_as(ctypes.c_void_p)) == -1
    bits = np.unpackbits(poly.view(np.uint8), bitorder="little")
    idx = np.nonzero(bits)[0]
    assert idx.max() < 19937
    if q <= 4:
        assert idx.tolist() == [624 << q]          # still a monomial below the degree
    else:
        assert idx.shape[0] >= 134                 # reduced at least once (q = 5: x^31 times phi's low terms)
    bg = np.random.MT19937(2024)
    bg.random_raw(624 - int(bg.state["state"]["pos"]))   # (a fresh numpy state has one word left: pos = 623)
    bg.random_raw(624)                             # one regeneration: every key word is a full stream word now
    base = bg.state["state"]["key"].astype(np.uint32)
    # the stream window z[0 .. 19937 + 624) from that state: the key itself, then the following keys
    walker = np.random.MT19937()
    walker.state = bg.state
    win = [base]
    while sum(w.shape[0] for w in win) < 19937 + 624:
        win.append(_mt_key_after(walker, 624))
    z = np.concatenate(win)
    jumped = np.zeros(624, dtype=np.uint32)
    for i in idx:
        jumped ^= z[i: i + 624]
    want = _mt_key_after(bg, 624 << q)
    assert np.array_equal(jumped, want)


def _synth_call(fn_name, params, sig, rs):
    """Arguments for one recorded call of the reference class into the backend module (tests/golden/swap_calls.json):
    arrays of the recorded dtype / shape / layout with valid contents chosen by the parameter's name."""
    by_name = dict(zip(params, sig))
    tab = by_name.get("Theta") or by_name.get("M1")
    tabB = by_name.get("Beta") or by_name.get("M2")
    nU = tab["shape"][0] if tab["ndim"] == 2 else None
    nI = tabB["shape"][0]
    k = tabB["shape"][1]
    ints = {"maxiter": 2, "check_every": 2, "users_per_batch": 0, "items_per_batch": 0, "sum_exp_trick": 0,
            "random_seed": 5, "verbose": 0, "nthreads": 1, "par_sh": 0, "has_valset": 0, "full_llk": 0, "keep_all_objs": 1,
            "alloc_full_phi": 0, "k": k, "return_all": 1}
    floats = {"a": 0.3, "a_prime": 0.3, "b_prime": 1.0, "c": 0.3, "c_prime": 0.3, "d_prime": 1.0, "stop_thr": 1e-3,
              "add_k_rte": 0.3, "add_t_rte": 0.3, "k_shp": 0.3 + k * 0.3, "t_shp": 0.3 + k * 0.3, "step_size_batch": 0.5,
              "multiplier_batch": 4.0}
    batch_users = None
    if fn_name == "partial_fit":
        batch_users = np.sort(rs.choice(nU, size=by_name["users_this_batch"]["shape"][0], replace=False))
        batch_items = np.sort(rs.choice(nI, size=by_name["items_this_batch"]["shape"][0], replace=False))
    args = []
    for name, d in zip(params, sig):
        kind = d["kind"]
        if kind == "ndarray":
            shape, dt = tuple(d["shape"]), np.dtype(d["dtype"])
            n = int(np.prod(shape))
            if name in ("users_this_batch",):
                v = batch_users.astype(dt)
            elif name in ("items_this_batch",):
                v = batch_items.astype(dt)
            elif name.startswith("ix_u"):
                v = (rs.choice(batch_users, size=n) if batch_users is not None else rs.randint(0, nU, size=n)).astype(dt)
            elif name.startswith("ix_i"):
                pool = batch_items if batch_users is not None else np.arange(nI)
                v = (rs.choice(pool, size=n, replace=(fn_name != "calc_user_factors"))).astype(dt)
            elif name == "st_ix_u":
                v = np.zeros(shape, dtype=dt)
            elif name.startswith("Y"):
                v = (1 + rs.poisson(1.0, size=n)).astype(dt)
            else:                                   # a table or a scalar-rate column: positive
                v = (0.3 + rs.random_sample(size=shape)).astype(dt)
            v = np.ascontiguousarray(v.reshape(shape))
            assert d["c_contiguous"], (fn_name, name)          # the reference always passes C-contiguous arrays
            args.append(v)
        elif kind == "callable":
            args.append(lambda it: 1.0 / np.sqrt(it + 2))
        elif kind == "str":
            args.append(d["value"])
        elif kind == "bool":
            args.append(True)
        elif kind == "int":
            args.append(int(by_name and ints.get(name, 1)) if name != "nY" else by_name["Y"]["shape"][0])
        elif kind == "float":
            args.append(float(floats[name]))
        else:
            raise AssertionError((fn_name, name, d))
    return args


def test_backend_accepts_what_the_reference_class_passes(cpu_ops_backend):
    """tests/golden/swap_calls.json holds the argument kinds / dtypes / shapes the REAL hpfrec.HPF class passed to every
    function of this backend module when it was swapped in for the compiled extension (tests/golden/swap_check.py, build
    container; the two runs agreed to 3e-6).  Here, without the reference: every recorded call shape is replayed with
    synthetic contents -- same arity and order (INIT:650-669, 882, 914-927, 1038, 1145, 1284, 1433), same dtypes
    (float32 / uint64 / Python scalars), same layouts -- and must be accepted and honour the in-place contracts."""
    import inspect
    import json
    be = cpu_ops_backend
    rec = json.load(open(os.path.join(GOLDEN, "swap_calls.json")))
    assert max(rec["agreement_max_rel"].values()) < 1e-4
    calls = rec["calls"]
    assert {"fit_hpf", "partial_fit", "calc_user_factors", "calc_llk", "predict_arr", "initialize_parameters", "cast_real_t",
            "cast_int", "cast_ind_type"} <= set(calls)
    rs = np.random.RandomState(11)
    assert be.cast_real_t(0.25) == np.float32(0.25) and isinstance(be.cast_int(3), (int, np.integer))
    for fn_name in ("fit_hpf", "partial_fit", "calc_user_factors", "calc_llk", "predict_arr", "initialize_parameters"):
        fn = getattr(be, fn_name)
        params = [p for p in inspect.signature(fn).parameters if p not in ("device_triplets", "resident")]
        for sig in calls[fn_name]:
            assert not any("keyword" in d for d in sig), "the reference passes everything positionally"
            assert len(sig) == len(params), (fn_name, len(sig), len(params))
            args = _synth_call(fn_name, params, sig, rs)
            before = [a.copy() if isinstance(a, np.ndarray) else None for a in args]
            out = fn(*args)
            named = dict(zip(params, args))
            if fn_name == "fit_hpf":
                i, temp, llk = out
                assert i == named["maxiter"] - 1 and len(temp) == 6
                assert np.isfinite(named["Theta"]).all() and np.isfinite(named["Beta"]).all()
                assert temp[0].shape == named["Theta"].shape and temp[4].shape == (named["Theta"].shape[0], 1)
            elif fn_name == "partial_fit":
                assert out is None
                changed = [n for n, a, b in zip(params, args, before) if isinstance(a, np.ndarray) and not np.array_equal(a, b)]
                assert {"Theta", "Beta", "Gamma_shp", "Lambda_shp", "k_rte", "t_rte"} <= set(changed), changed
            elif fn_name == "calc_user_factors":
                assert len(out) == 3 and out[2].shape == (named["nY"], named["k"]) and np.isfinite(named["Theta"]).all()
            elif fn_name == "calc_llk":
                assert np.isfinite(float(out))
            elif fn_name == "predict_arr":
                assert out.shape == named["ix_u"].shape and out.dtype == np.float32
            else:
                assert len(out) == 6 and out[0].shape == named["Theta"].shape


def _traced_plan(world, rank, schedule, nranges=2, k=50, nI=1003):
    """A C-issued iteration plan in trace mode (hpf_shard_desc.dry_run = 2): nothing is dereferenced or issued, so the
    table pointers are arbitrary non-null values and no GPU is needed."""
    from hpfrec_amd import shard_native as sn
    ld = _lib.ld_for_k(k)
    d = sn.ShardDesc()
    d.world, d.rank, d.k, d.ld, d.nU, d.nI = world, rank, k, ld, 500, nI
    fake = iter(range(0x10000, 0x7fffffff, 0x10000))
    for n in ("u_segs", "u_idx", "u_y", "u_row_seg_ptr", "u_multi_rows", "i_segs", "i_idx", "i_y", "i_row_seg_ptr", "eB", "part_u",
              "part_i", "Gamma_shp", "Theta", "k_rte", "k_rte_prev", "Lambda_shp", "Beta", "t_rte", "t_rte_prev", "csT", "csB",
              "csB_used", "csT_part", "csB_part", "acc_i", "acc_own", "e_own", "ag_recv", "shp_own"):
        setattr(d, n, next(fake))
    d.u_nseg, d.u_nmulti = 700, 3
    # item ranges: multiples of the world size, the last one ends in pad rows; issue order NOT ascending
    rows = -(-nI // world) * world
    cut = (rows // (2 * world)) * world if nranges > 1 else rows
    bounds = [(cut, rows), (0, cut)][: nranges] if nranges == 2 else [(0, rows)]
    d.nranges = len(bounds)
    seg = 0
    for j, (lo, hi) in enumerate(bounds):
        r = d.ranges[j]
        r.lo, r.hi, r.seg_lo, r.nseg, r.nmulti, r.short_rows = lo, hi, seg, hi - lo, 2, j & 1
        r.multi_rows = next(fake)
        seg += hi - lo
    d.csT_part_rows, d.user_sweep_grid, d.user_multi_grid = 64, 48, 4
    d.csB_part_rows, d.item_sweep_grid = 8 * world, 256
    d.e_own_ld = ld if schedule == 0 else (k + 4) // 4 * 4
    d.a = d.k_shp = d.add_k_rte = d.c = d.t_shp = d.add_t_rte = 0.3
    d.xstream = 0xE0
    d.dry_run, d.schedule = 2, schedule
    d.direct_prefetch, d.direct_pull_grid, d.direct_gather_gx = 1, 256, 16      # (read by the direct schedule only)
    plan = sn.ShardPlan(d)
    return plan, sn


@pytest.mark.parametrize("schedule", [0, 1])
@pytest.mark.parametrize("world,nranges", [(8, 2), (3, 2), (2, 1)])
def test_c_issued_iteration_traces(schedule, world, nranges):
    """What a multi-rank run of the C-issued iteration depends on, checked WITHOUT a GPU on the operations each rank's plan
    issues (trace mode of hpf_hip_shard_iterate / _join): every rank issues the SAME sequence of collectives (kind and
    element count -- a mismatch is a hang on real links), every stream wait names an event recorded before it, each
    range's reduce-scatter follows its sweep and precedes its all-gather, the user side is issued once per iteration,
    and a join issues no kernel."""
    CS = 0xC0
    traces = []
    for rank in range(world):
        plan, sn = _traced_plan(world, rank, schedule, nranges)
        ops = []
        for it in range(3):
            plan.iterate_raw(0x100 + (it & 1), 0x101 - (it & 1), it == 2, CS)
            ops.append(plan.trace())
        plan.join(CS)
        ops.append(plan.trace())
        plan.close()
        traces.append(ops)
    KER = sn.TRACE_KERNELS
    coll = lambda tr: [(i, a) for kind, i, st, a in tr if kind == sn.TRACE_COLLECTIVE]    # noqa: E731
    for step in range(4):
        assert all(coll(traces[r][step]) == coll(traces[0][step]) for r in range(world)), (schedule, step)
    for rank in range(world):
        recorded = set()
        for step, tr in enumerate(traces[rank]):
            names = [KER[i] for kind, i, st, a in tr if kind == sn.TRACE_KERNEL]
            for kind, i, st, a in tr:
                if kind == sn.TRACE_RECORD:
                    recorded.add(a)
                elif kind == sn.TRACE_WAIT:
                    assert a in recorded, (schedule, rank, step, hex(a))
            if step == 3:       # the join
                assert "item_apply" not in names and "sweep" not in names
                continue
            assert names.count("sweep_finalize") == 1 and names.count("sweep") == nranges
            c = coll(tr)
            n_rs = sum(1 for i, a in c if i == sn.COLL_REDUCE_SCATTER)
            n_ag = sum(1 for i, a in c if i == sn.COLL_ALL_GATHER)
            n_small = sum(1 for i, a in c if i == (sn.COLL_ALL_REDUCE | 0x100) or i == sn.COLL_ALL_REDUCE)
            assert n_rs == nranges and n_ag == (1 if schedule == 1 else nranges)
            assert n_small == 2
            # a range's reduce-scatter after its sweep, its all-gather after its reduce-scatter
            kinds = [("sweep", None) if (kind == sn.TRACE_KERNEL and KER[i] == "sweep") else
                     ("rs", a) if (kind == sn.TRACE_COLLECTIVE and i == sn.COLL_REDUCE_SCATTER) else
                     ("ag", a) if (kind == sn.TRACE_COLLECTIVE and i == sn.COLL_ALL_GATHER) else None
                     for kind, i, st, a in tr]
            kinds = [x for x in kinds if x]
            seen_sweeps = seen_rs = 0
            for what, a in kinds:
                if what == "sweep":
                    seen_sweeps += 1
                elif what == "rs":
                    seen_rs += 1
                    assert seen_rs <= seen_sweeps
                else:
                    assert seen_rs >= 1
            if schedule == 1:                   # apply of all ranges after the user side
                assert names.index("item_apply") > names.index("sweep_finalize")


@pytest.mark.parametrize("schedule", [0, 1])
def test_c_issued_iteration_traces_with_fewer_items_than_ranks(schedule):
    """5 items over 8 ranks: ranks 5-7 own pad rows only -- they launch no item finalizer (or shape half), yet take part in
    every collective with the same element counts as everybody else."""
    seqs, finalizers = [], []
    for rank in range(8):
        plan, sn = _traced_plan(8, rank, schedule, 1, nI=5)
        for it in range(2):
            plan.iterate_raw(0x100 + (it & 1), 0x101 - (it & 1), True, 0xC0)
        plan.join(0xC0)
        tr = plan.trace()
        plan.close()
        seqs.append([(i, a) for kind, i, st, a in tr if kind == sn.TRACE_COLLECTIVE])
        finalizers.append(sum(1 for kind, i, st, a in tr if kind == sn.TRACE_KERNEL and
                              sn.TRACE_KERNELS[i] in ("item_shape", "row_finalize_ranges")))
    assert all(s == seqs[0] for s in seqs) and len(seqs[0]) > 0
    assert finalizers == [2] * 5 + [0] * 3


@pytest.mark.parametrize("prefetch", [1, 0])
@pytest.mark.parametrize("world,nranges", [(8, 2), (3, 2), (2, 1), (8, 1)])
def test_direct_schedule_traces_have_no_collective(world, nranges, prefetch):
    """The direct (peer-mapped) schedule, HPF_SCHEDULE_DIRECT, in trace mode: NO collective of any kind is issued -- the
    exchange is inside the kernels -- and the flag protocol is consistent on every rank: range j's SWEPT flag is raised on
    entry of the launch that FOLLOWS the range's sweep on the compute stream (never by the sweep itself); on the exchange
    stream a one-wave wait for exactly that kind precedes the range's pull-reduce; the shape half follows the last pull;
    SHAPED is raised once per iteration after it; the colsum(Theta) launch does the waiting for the apply (GATHERED of this
    rank with prefetch, every owner's SHAPED without).  In steady state NO stream event ties the two streams (the flags do).
    An empty user shard (u_nseg = 0) still raises the last SWEPT flag through a signal-only launch."""
    from hpfrec_amd import shard_native as sn
    CS, XS = 0xC0, 0xE0
    SWEPT = lambda j: j                   # noqa: E731  (HPF_P2P_FLAG_SWEPT)
    SHAPED, GATHERED = 8, 29              # HPF_P2P_FLAG_SHAPED(0), HPF_P2P_FLAG_GATHERED
    per_rank = []
    for rank in range(world):
        for empty_users in ((False, True) if rank == world - 1 else (False,)):
            plan, _ = _traced_plan(world, rank, 3, nranges)
            plan.close()
            # (rebuild with the direct fields set: _traced_plan leaves them at zero)
            d = plan.desc
            d.direct_prefetch, d.direct_pull_grid, d.direct_gather_gx = prefetch, 256, 16
            if empty_users:
                d.u_nseg = 0
            plan = sn.ShardPlan(d)
            seq = []
            for it in range(3):
                plan.iterate_raw(0x100 + (it & 1), 0x101 - (it & 1), it == 2, CS)
                tr = plan.trace()
                assert not [1 for kind, i, st, a in tr if kind == sn.TRACE_COLLECTIVE], "a collective in the direct schedule"
                ker = [(sn.TRACE_KERNELS[i], st, a) for kind, i, st, a in tr if kind == sn.TRACE_KERNEL]
                names = [n for n, _, _ in ker]
                assert names.count("sweep") == nranges and names.count("pull_reduce") == nranges
                assert names.count("colsum_allreduce") == 2 and names.count("item_apply") == 1
                assert names.count("item_shape") == 1 and names.count("sweep_finalize") == (0 if empty_users else 1)
                # flags raised on the compute stream, in order: SWEPT(0) .. SWEPT(last), each by the launch after its sweep
                raised = [a - 1 for n, st, a in ker if st == CS and n in ("sweep", "sweep_finalize") and a > 0] + \
                         [a for n, st, a in ker if st == CS and n == "signal"]
                assert raised == [SWEPT(j) for j in range(nranges)], raised
                assert [a for n, st, a in ker if n == "sweep"][0] == 0          # the first sweep raises nothing
                # exchange stream: wait(SWEPT j) . pull_reduce(j) per range, then the shape half, then SHAPED
                xs_ops = [(n, a) for n, st, a in ker if st == XS]
                want = []
                for j in range(nranges):
                    want += [("wait", 1 << SWEPT(j)), ("pull_reduce", j)]
                want += [("item_shape", 0)]
                want += [("gather_pull", SHAPED | (GATHERED << 8))] if prefetch else [("signal", SHAPED)]
                assert xs_ops == want, xs_ops
                # the colsum(Theta) launch waits for what the apply needs; the apply follows it and precedes colsum(Beta)
                cs_ar = [a for n, st, a in ker if n == "colsum_allreduce"]
                assert cs_ar[0] & 0xFF == 0 and cs_ar[1] == 1              # HPF_P2P_VEC_CST then _CSB (which waits for nothing)
                assert (cs_ar[0] >> 8) & 0xFF == 1 + (GATHERED if prefetch else SHAPED)
                assert (cs_ar[0] >> 16) == (0 if prefetch else 1 << SHAPED)
                ia = names.index("item_apply")
                cs_pos = [i for i, n in enumerate(names) if n == "colsum_allreduce"]
                assert cs_pos[0] < ia < cs_pos[1]
                # (prefetch: the apply reads local memory only and waits for nothing itself -- the colsum launch did)
                assert ker[ia][2] == (0 if prefetch else 1 + SHAPED)
                # steady state: nothing recorded on, or waited for by, the compute stream; every wait after its record
                rec = set()
                for kind, i, st, a in tr:
                    if kind == sn.TRACE_RECORD:
                        rec.add(a)
                        assert st == XS or it == 0
                    elif kind == sn.TRACE_WAIT:
                        assert a in rec and it == 0
                seq.append([(n, a) for n, st, a in ker if n in ("wait", "pull_reduce", "gather_pull", "colsum_allreduce", "item_apply")])
            plan.join(CS)
            assert not [1 for kind, i, st, a in plan.trace() if kind in (sn.TRACE_COLLECTIVE, sn.TRACE_KERNEL)]
            plan.close()
            per_rank.append(seq)
    # what every rank waits for / publishes is the same sequence on every rank (a mismatch is a time-out on real links)
    assert all(s == per_rank[0] for s in per_rank)


def test_reference_order_column_sums_mode_on_standin(cpu_ops_backend, monkeypatch):
    """HPF_COLSUM_ORDER=reference (cavi.FullBatchCavi: the two column sums of an iteration in numpy's float32 row-after-row
    order, as PXI:236,255 form them) runs the same fit as the default -- at 100 rows the orders differ in the last bits only
    -- and is refused for a sharded model."""
    df, nU, nI = datagen.readme_counts()
    Y, iu, ii = datagen.triplets(df)
    out = {}
    for mode in ("tree", "reference"):
        monkeypatch.setenv("HPF_COLSUM_ORDER", mode)
        _, arrs, llk = _fit(cpu_ops_backend, Y, iu, ii, nU, nI, 30, 5, verbose=1, check_every=5)
        out[mode] = dict(arrs, llk=np.float64(llk))
    for n in out["tree"]:
        assert _maxrel(np.asarray(out["reference"][n]), np.asarray(out["tree"][n])) < 5e-6, n
