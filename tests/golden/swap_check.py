"""The drop-in claim of INTEGRATION.md section 1, checked against the REAL reference class.

Runs only in the build container (same rules as make_golden.py: the reference is compiled in a scratch copy OUTSIDE
this repo and imported from there; nothing of it travels):

    cp -r /root/reference /tmp/hpfrec_oracle && chmod -R u+w /tmp/hpfrec_oracle
    cd /tmp/hpfrec_oracle && python3 setup.py build_ext --inplace
    cd /root/repo && python tests/golden/swap_check.py

The reference's own `hpfrec.HPF` (hpfrec/__init__.py, "INIT") is driven twice through the same calls -- fit (full batch;
SVI with user batches, item batches, both; with a validation set), partial_fit x4, add_user, predict, topN, eval_llk,
predict_factors -- once with its compiled extension `hpfrec.cython_loops_float` and once with
`hpfrec_amd.cython_loops_float` assigned to that module global (the kernels replaced by the numpy stand-in of
tests/cpu_ops.py: there is no GPU in the build container).  The results must agree; the arguments the class passed to
every function of the swapped-in module (INIT:650-669, 882, 914-927, 1038, 1145, 1284-1291, 1433) are recorded -- kind,
dtype, shape, flags -- into tests/golden/swap_calls.json, so that tests/test_host_logic.py can assert, on any machine,
that the module still accepts exactly what the reference class passes.
"""
import contextlib
import io
import json
import os
import sys
import warnings

import numpy as np
import pandas as pd

REF = os.environ.get("HPFREC_REF_BUILD", "/tmp/hpfrec_oracle")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import hpfrec  # noqa: E402  (the reference, from the scratch build)
import cpu_ops  # noqa: E402
import datagen  # noqa: E402
from hpfrec_amd import cython_loops_float as ours  # noqa: E402

ours.HipOps = lambda device=None: cpu_ops.CpuOps()       # no GPU here: host logic + numpy kernels
REAL = hpfrec.cython_loops_float
NAMES = ("Theta", "Beta", "Gamma_shp", "Gamma_rte", "Lambda_shp", "Lambda_rte", "k_rte", "t_rte")


def describe(v):
    if isinstance(v, np.ndarray):
        return {"kind": "ndarray", "dtype": str(v.dtype), "ndim": v.ndim, "shape": list(v.shape),
                "c_contiguous": bool(v.flags.c_contiguous), "writeable": bool(v.flags.writeable)}
    if callable(v):
        return {"kind": "callable"}
    if v is None:
        return {"kind": "None"}
    if isinstance(v, (bool, np.bool_)):
        return {"kind": "bool"}
    if isinstance(v, (int, np.integer)):
        return {"kind": "int", "type": type(v).__name__}
    if isinstance(v, (float, np.floating)):
        return {"kind": "float", "type": type(v).__name__}
    if isinstance(v, str):
        return {"kind": "str", "value": v}
    return {"kind": type(v).__name__}


class Recorder:
    """hpfrec_amd.cython_loops_float with every call of a function noted (first call of each distinct signature)."""

    def __init__(self, mod):
        self._mod, self.calls = mod, {}

    def __getattr__(self, name):
        obj = getattr(self._mod, name)
        if not callable(obj) or isinstance(obj, type):
            return obj

        def wrapped(*a, **kw):
            sig = [describe(v) for v in a] + [dict(describe(v), keyword=k) for k, v in sorted(kw.items())]
            self.calls.setdefault(name, [])
            if sig not in self.calls[name]:
                self.calls[name].append(sig)
            return obj(*a, **kw)
        return wrapped


def quiet(fn, *a, **kw):
    with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return fn(*a, **kw)


def flows():
    """Every method of the class that reaches the extension module -> {label: array or scalar}."""
    out = {}
    df, nU, nI = datagen.readme_counts()
    base = dict(k=12, random_seed=123, ncores=1, reindex=True, verbose=False, stop_crit="maxiter", check_every=None,
                keep_all_objs=True, allow_inconsistent_math=False, use_float=True)

    def grab(tag, m):
        for n in NAMES:
            out["%s/%s" % (tag, n)] = np.array(getattr(m, n))

    m = hpfrec.HPF(maxiter=10, **base)
    quiet(m.fit, df.copy())
    grab("full", m)
    u0, i0 = df.UserId.iloc[0], df.ItemId.iloc[0]
    out["full/predict"] = np.array(quiet(m.predict, user=[u0, df.UserId.iloc[5]], item=[i0, df.ItemId.iloc[7]]))
    out["full/predict_scalar"] = np.array(quiet(m.predict, user=u0, item=i0))
    out["full/topN"] = np.array(quiet(m.topN, u0, n=7, exclude_seen=True))
    out["full/eval_llk"] = np.float64(quiet(m.eval_llk, df.copy())["llk"])
    out["full/eval_llk_full"] = np.float64(quiet(m.eval_llk, df.copy(), full_llk=True)["llk"])
    new = pd.DataFrame({"ItemId": df.ItemId.iloc[:9].to_numpy(), "Count": df.Count.iloc[:9].to_numpy()})
    out["full/predict_factors"] = np.array(quiet(m.predict_factors, new.copy(), random_seed=3))
    quiet(m.add_user, user_id=10 ** 6, counts_df=new.copy(), random_seed=3)
    out["full/add_user_Theta_last"] = np.array(m.Theta[-1])
    # verbose checks + stopping rule + validation set
    val = df.sample(200, random_state=2)
    m = hpfrec.HPF(maxiter=40, **dict(base, stop_crit="val-llk", check_every=5, stop_thr=1e-3, verbose=True))
    quiet(m.fit, df.copy(), val_set=val.copy())
    grab("valset", m)
    out["valset/niter"] = np.int64(m.niter)
    m = hpfrec.HPF(maxiter=10, **dict(base, verbose=True, check_every=5))
    quiet(m.fit, df.copy())
    out["verbose/train_llk"] = np.float64(m.train_llk)
    # stochastic fits: the three batch modes
    for tag, upb, ipb in (("svi_users", 30, None), ("svi_items", None, 40), ("svi_both", 20, 25)):
        m = hpfrec.HPF(maxiter=4, users_per_batch=upb, items_per_batch=ipb, **base)
        quiet(m.fit, df.copy())
        grab(tag, m)
    # partial_fit: the README sequence (keep_data=False), user and item batches
    m = hpfrec.HPF(**dict(base, reindex=False, keep_data=False))
    for j in range(4):
        sub = df.loc[df.UserId.isin(np.arange(25 * j, 25 * (j + 1)))]
        quiet(m.partial_fit, sub.copy(), batch_type="users", nusers=nU, nitems=nI)
    sub = df.loc[df.ItemId.isin(np.arange(0, 30))]
    quiet(m.partial_fit, sub.copy(), batch_type="items", nusers=nU, nitems=nI)
    grab("partial_fit", m)
    return out


def main():
    hpfrec.cython_loops_float = REAL
    want = flows()
    rec = Recorder(ours)
    hpfrec.cython_loops_float = rec
    try:
        got = flows()
    finally:
        hpfrec.cython_loops_float = REAL
    worst = {}
    for key in want:
        a, b = np.asarray(want[key], dtype=np.float64), np.asarray(got[key], dtype=np.float64)
        assert a.shape == b.shape, key
        if key.endswith("topN") or key.endswith("niter"):
            assert np.array_equal(a, b), key
            continue
        tag = key.split("/")[0]
        err = float(np.max(np.abs(a - b) / np.maximum(np.abs(a), 1e-30))) if a.size else 0.0
        worst[tag] = max(worst.get(tag, 0.0), err)
    print(json.dumps(worst, indent=1))
    tol = {"full": 2e-5, "valset": 5e-4, "verbose": 2e-5, "svi_users": 1e-4, "svi_items": 1e-4, "svi_both": 1e-4,
           "partial_fit": 5e-5}
    for tag, err in worst.items():
        assert err < tol[tag], (tag, err)
    # the functions INIT calls must all have been reached through the swapped-in module
    for fn in ("fit_hpf", "partial_fit", "calc_user_factors", "calc_llk", "predict_arr", "cast_real_t", "cast_int",
               "cast_ind_type"):
        assert fn in rec.calls, fn
    path = os.path.join(HERE, "swap_calls.json")
    json.dump({"made_by": "tests/golden/swap_check.py (reference class driving hpfrec_amd.cython_loops_float)",
               "agreement_max_rel": worst, "calls": rec.calls}, open(path, "w"), indent=1, sort_keys=True)
    print("wrote", path, {k: len(v) for k, v in rec.calls.items()})


if __name__ == "__main__":
    main()
