"""hpfrec_amd -- Hierarchical Poisson Factorization on AMD MI355X.

`HPF` keeps the Python surface of david-cortes/hpfrec (class HPF,
/root/reference/hpfrec/__init__.py:11-1458, "INIT" below): same constructor keywords and defaults,
same methods (fit, partial_fit, predict_factors, add_user, predict, topN, eval_llk), same public
attributes after fitting (Theta, Beta, user_mapping_, item_mapping_, user_dict_, item_dict_,
is_fitted, niter, train_llk, and Gamma_shp ... t_rte with keep_all_objs).  What changes is the
engine: every numeric loop runs on the GPU through libhpf_hip.so (module cython_loops_float, the
drop-in for the reference's compiled extension of that name).

Only float32 (`use_float=True`, the reference default) is built; `ncores`,
`allow_inconsistent_math` and `alloc_full_phi` are accepted for compatibility and have no effect
(the device path is parallel, reproducible, and never materialises phi).
"""
import ctypes
import inspect
import multiprocessing
import os
import types
import warnings

import numpy as np
import pandas as pd
from scipy.sparse import coo_array, issparse

from . import cython_loops_float, ingest, resident

__all__ = ["HPF"]

_COLS = ["UserId", "ItemId", "Count"]


def _to_float(x, name):
    if isinstance(x, (int, np.integer)) and not isinstance(x, bool):
        x = float(x)
    if not isinstance(x, float):
        raise AssertionError("'%s' must be a float" % name)
    if not x > 0:
        raise AssertionError("'%s' must be positive" % name)
    return x


def _resolve_ncores(ncores):
    if ncores is None:
        return 1
    if ncores < 1:
        ncores = multiprocessing.cpu_count()
    assert isinstance(ncores, int) and ncores > 0
    return ncores


def _frame_from(obj, what):
    """DataFrame / ndarray -> a private 3-column frame (INIT:437-447, 526-536)."""
    if isinstance(obj, np.ndarray):
        assert obj.ndim > 1 and obj.shape[1] >= 3
        return pd.DataFrame(obj[:, :3], copy=True, columns=_COLS)
    if isinstance(obj, pd.DataFrame):
        assert obj.shape[0] > 0
        for c in _COLS:
            assert c in obj.columns
        return obj[_COLS].copy()
    return None


def _codes(values, mapping):
    """position of each value in `mapping`, -1 when absent (pd.Categorical(...).codes semantics)."""
    return np.require(pd.Categorical(values, mapping).codes, requirements=["ENSUREARRAY"])


_BIG_LOOKUP = 200_000    # from this many ids on, numeric id lookups run on the device (hpfrec_amd.ingest.IdLookup)


class IdTable(dict):
    """`user_dict_` / `item_dict_`: external id -> internal position (INIT:529-531), as the plain dict the reference
    builds -- but filled from the mapping only when something needs the whole table (iteration, len, ==, repr).  A
    single lookup, which is all the package itself does, is a binary search in the sorted ids the device renumbering
    produced anyway; entries added later (add_user) live in the dict part.  Building the 1.4 million dict entries of a
    C3-sized model took a quarter of a one-second fit.  Without sorted ids (string ids: the pandas path) the dict is
    filled at once, as before."""

    def __init__(self, mapping, sorted_ids=None, codes=None):
        super().__init__()
        self._mapping, self._sorted, self._codes, self._filled = mapping, sorted_ids, codes, False
        if sorted_ids is None:
            self._fill()

    def _fill(self):
        if not self._filled:
            later = dict(dict.items(self))                         # entries assigned since (they win)
            dict.update(self, zip(self._mapping.tolist(), range(self._mapping.shape[0])))
            dict.update(self, later)
            self._filled = True

    def _search(self, key):
        try:
            pos = int(np.searchsorted(self._sorted, key))
            if pos < self._sorted.shape[0] and self._sorted[pos] == key:
                return int(self._codes[pos])
        except (TypeError, ValueError):
            pass
        return None

    def __missing__(self, key):
        if not self._filled:
            pos = self._search(key)
            if pos is not None:
                return pos
        raise KeyError(key)

    def __contains__(self, key):
        return dict.__contains__(self, key) or (not self._filled and self._search(key) is not None)

    def get(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            return default

    __hash__ = None


def _whole_table(name):
    def method(self, *args, **kwargs):
        self._fill()
        return getattr(dict, name)(self, *args, **kwargs)
    method.__name__ = name
    return method


# every dict method that reads or edits the table as a whole, or whose result depends on whether a key is present in the
# dict part (setdefault would shadow a mapped id with its default; clear would leave the mapping answering lookups)
for _n in ("__len__", "__iter__", "__eq__", "__ne__", "__repr__", "keys", "values", "items", "copy", "pop", "popitem",
           "__delitem__", "__reversed__", "__or__", "__ror__", "__ior__", "setdefault", "clear", "update", "__sizeof__",
           "__reduce_ex__", "__reduce__"):
    setattr(IdTable, _n, _whole_table(_n))


def _default_step_size(x):
    """The reference's default SVI schedule (INIT:207: `lambda x: 1/np.sqrt(x+2)`), as a named function so that a model
    with default settings can be pickled."""
    return 1 / np.sqrt(x + 2)


class HPF:
    """Hierarchical Poisson Factorization (Gopalan, Hofman & Blei 2015) fitted by mean-field
    coordinate-ascent variational inference, full-batch or stochastic.

    Parameters (identical to hpfrec.HPF, INIT:87-178)
    ----------
    k : int -- number of latent factors.
    a, a_prime, b_prime, c, c_prime, d_prime : float -- Gamma prior hyper-parameters.
    ncores : int -- accepted, unused on the device path.
    stop_crit : 'maxiter' | 'train-llk' | 'val-llk' | 'diff-norm'.
    check_every : int or None -- evaluate the stopping criterion every N iterations.
    stop_thr : float -- threshold on 1 - llk/llk_prev, or on ||Theta - Theta_prev|| for 'diff-norm'.
    users_per_batch, items_per_batch : int or None -- mini-batch sizes for stochastic VI.
    step_size : function(int) -> float in (0,1) -- SVI step size schedule.
    maxiter : int or None.
    use_float : bool -- must be True here (float32 tables).
    reindex : bool -- renumber user/item ids internally by first appearance.
    verbose, random_seed, allow_inconsistent_math, full_llk, alloc_full_phi, keep_data,
    save_folder, produce_dicts, keep_all_objs, sum_exp_trick : as in the reference.

    Attributes: Theta (nusers,k), Beta (nitems,k), user_mapping_, item_mapping_, user_dict_,
    item_dict_, is_fitted, niter, train_llk.
    """

    # the eight state arrays live in a ResidentState (host copy + device copy with explicit validity, see
    # hpfrec_amd/resident.py); as attributes they behave like the reference's plain numpy arrays
    Theta = resident.StateArray("Theta")
    Beta = resident.StateArray("Beta")
    Gamma_shp = resident.StateArray("Gamma_shp")
    Gamma_rte = resident.StateArray("Gamma_rte")
    Lambda_shp = resident.StateArray("Lambda_shp")
    Lambda_rte = resident.StateArray("Lambda_rte")
    k_rte = resident.StateArray("k_rte")
    t_rte = resident.StateArray("t_rte")
    seen = resident.DeviceBackedArray("seen")          # (48M ids at C3: produced on the device, downloaded when read)

    def __init__(self, k=30, a=0.3, a_prime=0.3, b_prime=1.0, c=0.3, c_prime=0.3, d_prime=1.0, ncores=-1,
                 stop_crit='maxiter', check_every=10, stop_thr=1e-3, users_per_batch=None, items_per_batch=None,
                 step_size=_default_step_size, maxiter=100, use_float=True, reindex=True, verbose=True,
                 random_seed=None, allow_inconsistent_math=False, full_llk=False, alloc_full_phi=False,
                 keep_data=True, save_folder=None, produce_dicts=True, keep_all_objs=True, sum_exp_trick=False):
        assert isinstance(k, int) and k > 0
        self.k = k
        self.a = _to_float(a, "a")
        self.a_prime = _to_float(a_prime, "a_prime")
        self.b_prime = _to_float(b_prime, "b_prime")
        self.c = _to_float(c, "c")
        self.c_prime = _to_float(c_prime, "c_prime")
        self.d_prime = _to_float(d_prime, "d_prime")
        self.ncores = _resolve_ncores(ncores)

        if random_seed is not None:
            assert isinstance(random_seed, int)
        assert stop_crit in ('maxiter', 'train-llk', 'val-llk', 'diff-norm')

        if maxiter is None:
            if stop_crit == 'maxiter':
                raise ValueError("If 'stop_crit' is set to 'maxiter', must provide a maximum number of iterations.")
            maxiter = 10 ** 10
        else:
            assert isinstance(maxiter, int) and maxiter > 0

        if check_every is None:
            if stop_crit != 'maxiter':
                raise ValueError("If 'stop_crit' is not 'maxiter', must input after how many iterations to calculate it.")
            check_every = 0
        else:
            assert isinstance(check_every, int) and 0 < check_every <= maxiter

        if isinstance(stop_thr, int):
            stop_thr = float(stop_thr)
        if stop_thr is not None:
            assert isinstance(stop_thr, float) and stop_thr > 0

        if save_folder is not None:
            save_folder = os.path.expanduser(save_folder)
            assert os.path.exists(save_folder)

        verbose = bool(verbose)
        if stop_crit == 'maxiter' and not verbose:
            check_every = 0  # INIT:291-292: nothing to print, nothing to evaluate

        if not isinstance(step_size, types.FunctionType):
            raise ValueError("'step_size' must be a function.")
        if len(inspect.getfullargspec(step_size).args) < 1:
            raise ValueError("'step_size' must be able to take the iteration number as input.")
        for probe in (0, 1):
            assert 0 <= step_size(probe) <= 1

        def batch_size(v):
            if v is None:
                return 0
            if isinstance(v, float):
                v = int(v)
            assert isinstance(v, int) and v > 0
            return v

        self.users_per_batch = batch_size(users_per_batch)
        self.items_per_batch = batch_size(items_per_batch)

        self.allow_inconsistent_math = bool(allow_inconsistent_math)
        self.use_float = bool(use_float)
        self.random_seed = random_seed
        self.stop_crit = stop_crit
        self.reindex = bool(reindex)
        self.keep_data = bool(keep_data)
        self.maxiter = maxiter
        self.check_every = check_every
        self.stop_thr = stop_thr
        self.save_folder = save_folder
        self.verbose = verbose
        self.produce_dicts = bool(produce_dicts) and self.reindex
        self.full_llk = bool(full_llk)
        self.alloc_full_phi = bool(alloc_full_phi)
        self.keep_all_objs = bool(keep_all_objs)
        self.sum_exp_trick = bool(sum_exp_trick)
        self.step_size = step_size

        self.Theta = None
        self.Beta = None
        self.user_mapping_ = None
        self.item_mapping_ = None
        self.user_dict_ = None
        self.item_dict_ = None
        self.is_fitted = False
        self.niter = None
        self.train_llk = None

    # ------------------------------------------------------------------------------------------
    def _backend(self):
        if not self.use_float:
            raise NotImplementedError("hpfrec_amd: only use_float=True (float32) is built for the HIP path")
        return cython_loops_float

    # ------------------------------------------------------------------------------------------
    def fit(self, counts_df, val_set=None):
        """Fit the model to triplets (UserId, ItemId, Count): a DataFrame with those columns, an
        array whose first three columns are those, or a scipy COO array (forces reindex=False).
        `val_set` (same formats) is only used with stop_crit='val-llk' / 'maxiter'.  Inputs may be
        modified in place, as in the reference (INIT:360-432).  Returns self."""
        if self.stop_crit == 'val-llk' and val_set is None:
            raise ValueError("If 'stop_crit' is set to 'val-llk', must provide a validation set.")
        if self.verbose:
            self._print_st_msg()
        self.__dict__.pop("_tick_t", None)
        self._tick("start")
        self._process_data(counts_df)
        if self.verbose:
            self._print_data_info()
        if (val_set is not None) and (self.stop_crit not in ("diff-norm", "train-llk")):
            self._process_valset(val_set)
        else:
            self.val_set = None

        self._cast_before_fit()
        self._tick("validation set, casts")
        self._fit()
        self._tick("fit_hpf (init, layout, iterations, outputs to host)")

        if self.keep_data:
            if self.users_per_batch == 0:
                self._store_metadata()
            else:
                self._st_ix_user = self._st_ix_user[:-1]
        self._dev_triplets = None
        if self.produce_dicts and self.reindex:
            srt = self.__dict__.pop("_sorted_ids", None) or {}
            self.user_dict_ = IdTable(self.user_mapping_, *srt.get("user", ()))
            self.item_dict_ = IdTable(self.item_mapping_, *srt.get("item", ()))
        self.is_fitted = True
        del self.input_df
        del self.val_set
        return self

    def _process_data(self, input_df):
        be = self._backend()
        known_shape = False
        frame = _frame_from(input_df, "counts_df")
        if frame is None:
            if issparse(input_df) and input_df.format == "coo":
                self.nusers, self.nitems = input_df.shape
                frame = pd.DataFrame({"UserId": input_df.row, "ItemId": input_df.col, "Count": input_df.data},
                                     copy=False)
                self.reindex = False
                known_shape = True
            else:
                raise ValueError("'input_df' must be a pandas data frame, numpy array, or scipy sparse coo_array.")

        # llk-based criteria need counts >= 1 (log of the rate); the others only positive ones (INIT:462-475)
        thr = 0 if self.stop_crit in ('maxiter', 'diff-norm') else 0.9
        zero_msg = ("'counts_df' contains observations with a count value less than 1, these will be ignored."
                    " Any user or item associated exclusively with zero-value observations will be excluded."
                    " If using 'reindex=False', make sure that your data still meets the necessary criteria."
                    " If you still want to use these observations, set 'stop_crit' to 'diff-norm' or 'maxiter'.")
        self._dev_triplets = None
        tick = self._tick
        # numeric ids: filter, renumbering and (later) the seen-items index run on the device (hpfrec_amd/ingest.py);
        # the triplets stay there for the fit.  Other id types (strings, objects): pandas, as the reference does.
        import torch
        dev = be._make_ops().device
        du = ingest.to_device_ids(frame["UserId"].to_numpy(copy=False), dev)
        di = ingest.to_device_ids(frame["ItemId"].to_numpy(copy=False), dev) if du is not None else None
        on_device = di is not None and frame["Count"].dtype.kind in "iuf"
        tick("ids to device")
        if on_device:
            cnt = torch.from_numpy(np.ascontiguousarray(frame["Count"].to_numpy(copy=False))).to(dev)
            drop = cnt <= thr
            if bool(drop.any()):
                warnings.warn(zero_msg)
                keep = ~drop
                du, di, cnt = du[keep], di[keep], cnt[keep]
                frame = frame.loc[keep.cpu().numpy()]
            dy = cnt.to(torch.float32)
            tick("count filter")
        else:
            drop = frame["Count"] <= thr
            if drop.any():
                warnings.warn(zero_msg)
                frame = frame.loc[~drop]
        self.input_df = frame

        if self.reindex:
            # first-appearance numbering, exactly pd.factorize (INIT:478-479)
            if on_device:
                du, umap_d, us, uc = ingest.factorize(du, with_sorted=True)
                di, imap_d, is_, ic = ingest.factorize(di, with_sorted=True)
                umap = umap_d.cpu().numpy().astype(frame["UserId"].dtype, copy=False)
                imap = imap_d.cpu().numpy().astype(frame["ItemId"].dtype, copy=False)
                # (id -> position tables for user_dict_ / item_dict_: the sort of the renumbering is one already)
                self._sorted_ids = {"user": (us.cpu().numpy().astype(umap.dtype, copy=False), uc.cpu().numpy()),
                                    "item": (is_.cpu().numpy().astype(imap.dtype, copy=False), ic.cpu().numpy())}
                tick("factorize")
                # The renumbered ids stay on the device (self._dev_triplets): the fit, the seen-items index and the
                # batches read them there, and input_df is deleted when fit() ends (INIT:688), so the frame's id
                # columns -- still the caller's raw ids -- are dropped instead of being overwritten with 2 x nnz codes
                # brought back over PCIe (0.2 s at 48M rows).
                frame = self.input_df = frame[["Count"]]
            else:
                ucodes, umap = pd.factorize(frame["UserId"])
                icodes, imap = pd.factorize(frame["ItemId"])
                frame["UserId"], frame["ItemId"] = ucodes, icodes
            self.user_mapping_ = np.require(umap, requirements=["ENSUREARRAY"]).reshape(-1)
            self.item_mapping_ = np.require(imap, requirements=["ENSUREARRAY"]).reshape(-1)
            self.nusers = self.user_mapping_.shape[0]
            self.nitems = self.item_mapping_.shape[0]
            if self.save_folder is not None:
                if self.verbose:
                    print("\nSaving user and item mappings...\n")
                pd.Series(self.user_mapping_).to_csv(os.path.join(self.save_folder, 'users.csv'), index=False)
                pd.Series(self.item_mapping_).to_csv(os.path.join(self.save_folder, 'items.csv'), index=False)
        elif not known_shape:
            self.nusers = frame["UserId"].max() + 1
            self.nitems = frame["ItemId"].max() + 1

        if self.save_folder is not None:
            with open(os.path.join(self.save_folder, "hyperparameters.txt"), "w") as fh:
                for name in ("a", "a_prime", "b_prime", "c", "c_prime", "d_prime"):
                    fh.write("%s: %.3f\n" % (name, getattr(self, name)))
                fh.write("k: %d\n" % self.k)
                fh.write("random seed: %s\n" % ("None" if self.random_seed is None else "%d" % self.random_seed))

        self._cast_frame(self.input_df, be)
        if on_device:
            if du.numel() and (int(du.min()) < 0 or int(di.min()) < 0):
                raise ValueError("user/item ids must be non-negative")
            self._dev_triplets = (du, di, dy)
        tick("cast")

        if self.users_per_batch != 0:
            if self.nusers < self.users_per_batch:
                warnings.warn("Batch size passed is larger than number of users. Will set it to nusers/10.")
                self.users_per_batch = int(np.ceil(self.nusers / 10))
            if not on_device:      # (the engine groups the triplets by user itself, on the device: no host sort)
                self.input_df.sort_values('UserId', inplace=True)
            self._store_metadata(for_partial_fit=True)

    def _tick(self, phase):
        """HPF_TIMING=1: wall time per phase of fit() in self.timings_ (device work synchronised at phase ends)."""
        if os.environ.get("HPF_TIMING") != "1":
            return
        import time
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        now = time.perf_counter()
        t = self.__dict__.setdefault("timings_", {})
        t[phase] = t.get(phase, 0.0) + now - self.__dict__.get("_tick_t", now)
        self._tick_t = now

    @staticmethod
    def _cast_frame(frame, be):
        if frame['Count'].dtype != be.c_real_t:
            frame['Count'] = frame["Count"].astype(be.c_real_t)
        for col in ("UserId", "ItemId"):
            if col in frame.columns and frame[col].dtype != be.obj_ind_type:   # (absent: the ids live on the device)
                frame[col] = frame[col].astype(be.obj_ind_type)

    def _process_valset(self, val_set, valset=True):
        be = self._backend()
        frame = _frame_from(val_set, "val_set")
        if frame is None:
            if issparse(val_set) and val_set.format == "coo":
                assert val_set.shape[0] <= self.nusers and val_set.shape[1] <= self.nitems
                frame = pd.DataFrame({"UserId": val_set.row, "ItemId": val_set.col, "Count": val_set.data}, copy=False)
            else:
                raise ValueError("'val_set' must be a pandas data frame, numpy array, or sparse coo_array.")
        thr = 0 if self.stop_crit == 'val-llk' else 0.9
        drop = frame["Count"] <= thr
        if drop.any():
            warnings.warn("'val_set' contains observations with a count value less than 1, these will be ignored.")
            frame = frame.loc[~drop]
        self.val_set = frame

        if self.reindex:
            frame['UserId'] = self._codes_of(frame["UserId"], "user")
            frame['ItemId'] = self._codes_of(frame["ItemId"], "item")
            frame = frame.loc[(frame["UserId"] != -1) & (frame["ItemId"] != -1)]
            self.val_set = frame
            if frame.shape[0] == 0:
                if not valset:
                    raise ValueError("'input_df' has no combinations of users and items"
                                     "in common with the training set.")
                warnings.warn("Validation set has no combinations of users and items"
                              " in common with training set. If 'stop_crit' was set"
                              " to 'val-llk', will now be switched to 'train-llk'.")
                if self.stop_crit == 'val-llk':
                    self.stop_crit = 'train-llk'
                self.val_set = None
                return
            frame.reset_index(drop=True, inplace=True)
        self._cast_frame(self.val_set, be)

    def _codes_of(self, values, which):
        """Internal numbering of external ids (-1: not in the training data).  Large numeric queries go through a
        device lookup table built once per mapping (sorted mapping + searchsorted); everything else through pandas."""
        mapping = self.user_mapping_ if which == "user" else self.item_mapping_
        vals = values.to_numpy(copy=False) if hasattr(values, "to_numpy") else np.asarray(values)
        if vals.shape[0] >= _BIG_LOOKUP and vals.dtype.kind in "iuf" and mapping.dtype.kind in "iuf":
            cache = self.__dict__.setdefault("_id_lookup", {})
            hit = cache.get(which)
            if hit is None or hit[0] is not mapping:
                dev = self._backend()._make_ops().device
                hit = cache[which] = (mapping, ingest.IdLookup(mapping, dev))
            codes = hit[1](vals)
            if codes is not None:
                return codes.cpu().numpy()
        return _codes(vals, mapping)

    def _store_metadata(self, for_partial_fit=False):
        """CSR bookkeeping of who saw what, for topN(exclude_seen=True) and for the SVI user batches
        (INIT:587-606).  scipy's tocsr() merges duplicate pairs here, as in the reference -- that only
        affects the `seen` lists, never the fit."""
        be = self._backend()
        if self.verbose and for_partial_fit:
            print("Creating user indices for stochastic optimization...")
        trip = getattr(self, "_dev_triplets", None)
        if trip is not None:
            # the same three arrays as scipy's coo -> csr below, from one device sort of the pair keys
            n_seen, indptr, seen = ingest.seen_metadata(trip[0], trip[1], self.nusers, self.nitems)
            idt = ingest.SEEN_INDEX_DTYPE
            indptr = indptr.cpu().numpy().astype(idt)
            self._n_seen_by_user = n_seen.cpu().numpy().astype(idt)
            assert seen.dtype.itemsize == np.dtype(idt).itemsize       # (int64, as scipy's indices for 64-bit ids)
            type(self).seen.set_device(self, seen)       # stays on the device until somebody reads `model.seen`
            if for_partial_fit:
                self._st_ix_user = np.require(indptr, dtype=be.obj_ind_type, requirements=["ENSUREARRAY", "C_CONTIGUOUS"])
            else:
                self._st_ix_user = indptr[:-1]
            self._tick("seen-items index")
            return
        X = coo_array((self.input_df["Count"].to_numpy(copy=False),
                       (self.input_df["UserId"].to_numpy(copy=False), self.input_df["ItemId"].to_numpy(copy=False))),
                      shape=(self.nusers, self.nitems), dtype=ctypes.c_float).tocsr()
        self._n_seen_by_user = X.indptr[1:] - X.indptr[:-1]
        if for_partial_fit:
            self._st_ix_user = np.require(X.indptr, dtype=be.obj_ind_type, requirements=["ENSUREARRAY", "C_CONTIGUOUS"])
            self.input_df.sort_values('UserId', inplace=True)
        else:
            self._st_ix_user = X.indptr[:-1]
        self.seen = X.indices

    def _cast_before_fit(self):
        be = self._backend()
        self._state.set_host("Theta", np.empty((self.nusers, self.k), dtype=be.c_real_t), private=True)
        self._state.set_host("Beta", np.empty((self.nitems, self.k), dtype=be.c_real_t), private=True)
        self.k = be.cast_ind_type(self.k)
        self.nusers = be.cast_ind_type(self.nusers)
        self.nitems = be.cast_ind_type(self.nitems)
        self.ncores = be.cast_int(self.ncores)
        self.maxiter = be.cast_int(self.maxiter) if self.maxiter < 2 ** 31 else 2 ** 31 - 1
        self.verbose = be.cast_int(self.verbose)
        self.random_seed = be.cast_int(0 if self.random_seed is None else self.random_seed)
        self.check_every = be.cast_int(self.check_every)
        for name in ("stop_thr", "a", "a_prime", "b_prime", "c", "c_prime", "d_prime"):
            setattr(self, name, be.cast_real_t(getattr(self, name)))
        if self.save_folder is None:
            self.save_folder = ""

    @staticmethod
    def _col(frame, col, dtype):
        return np.require(frame[col].to_numpy(copy=False), dtype=dtype, requirements=["ENSUREARRAY", "C_CONTIGUOUS"])

    def _ids_for_fit(self, col, be):
        """The host id column handed to fit_hpf; empty when the (renumbered) ids are on the device only -- fit_hpf
        does not read the host ids when it gets `device_triplets`."""
        if col not in self.input_df.columns:
            assert getattr(self, "_dev_triplets", None) is not None
            return np.empty(0, dtype=be.obj_ind_type)
        return self._col(self.input_df, col, be.obj_ind_type)

    def _fit(self):
        be = self._backend()
        if self.val_set is None:
            use_valset = 0
            empty = np.empty(0)
            self.val_set = pd.DataFrame({"UserId": empty.astype(be.obj_ind_type), "ItemId": empty.astype(be.obj_ind_type),
                                         "Count": empty.astype(be.c_real_t)})
        else:
            use_valset = 1
        if self.users_per_batch == 0:
            self._st_ix_user = np.arange(1).astype(be.obj_ind_type)

        self.niter, temp, self.train_llk = be.fit_hpf(
            self.a, self.a_prime, self.b_prime, self.c, self.c_prime, self.d_prime,
            self._col(self.input_df, "Count", be.c_real_t),
            self._ids_for_fit("UserId", be), self._ids_for_fit("ItemId", be),
            self._state.peek_host("Theta"), self._state.peek_host("Beta"),
            self.maxiter, self.stop_crit, self.check_every, self.stop_thr,
            self.users_per_batch, self.items_per_batch, self.step_size, be.cast_int(self.sum_exp_trick),
            self._st_ix_user.astype(be.obj_ind_type),
            self.save_folder, self.random_seed, self.verbose,
            self.ncores, be.cast_int(self.allow_inconsistent_math), use_valset,
            self._col(self.val_set, "Count", be.c_real_t),
            self._col(self.val_set, "UserId", be.obj_ind_type),
            self._col(self.val_set, "ItemId", be.obj_ind_type),
            be.cast_int(self.full_llk), be.cast_int(self.keep_all_objs), be.cast_int(self.alloc_full_phi),
            device_triplets=getattr(self, "_dev_triplets", None), resident=self._state)

        if self.users_per_batch == 0:
            del self._st_ix_user
        if self.keep_all_objs and temp is None:
            return          # the fit ended on the device and handed its tables to self._state (fit_hpf: `resident`)
        for name in ("Theta", "Beta"):          # filled in place by fit_hpf: the device copies (if any) are old
            self._state.set_host(name, self._state.host[name], private=not self._state.handed.get(name, True))
        if self.keep_all_objs:
            for name, arr in zip(("Gamma_shp", "Gamma_rte", "Lambda_shp", "Lambda_rte", "k_rte", "t_rte"), temp):
                self._state.set_host(name, arr, private=True)

    # ------------------------------------------------------------------------------------------
    def _process_data_single(self, counts_df):
        """(ItemId, Count) rows of ONE user -> internal item numbering (INIT:682-712)."""
        assert self.is_fitted and self.keep_all_objs
        be = self._backend()
        if isinstance(counts_df, np.ndarray):
            assert counts_df.ndim > 1 and counts_df.shape[1] >= 2
            counts_df = pd.DataFrame(counts_df[:, :2], columns=["ItemId", "Count"], copy=True)
        elif isinstance(counts_df, pd.DataFrame):
            assert counts_df.shape[0] > 0 and "ItemId" in counts_df.columns and "Count" in counts_df.columns
            counts_df = counts_df[["ItemId", "Count"]].copy()
        else:
            raise ValueError("'counts_df' must be a pandas data frame or a numpy array")
        if self.reindex:
            msg = "Can only make calculations for items that were in the training set."
            if self.produce_dicts:
                try:
                    counts_df["ItemId"] = counts_df["ItemId"].map(lambda x: self.item_dict_[x])
                except Exception:
                    raise ValueError(msg)
            else:
                counts_df["ItemId"] = _codes(counts_df["ItemId"].to_numpy(copy=False), self.item_mapping_)
                if (counts_df["ItemId"] == -1).any():
                    raise ValueError(msg)
        counts_df["ItemId"] = np.require(counts_df["ItemId"], dtype=be.obj_ind_type)
        counts_df["Count"] = np.require(counts_df["Count"], dtype=be.c_real_t)
        return counts_df

    def partial_fit(self, counts_df, batch_type='users', step_size=None, nusers=None, nitems=None,
                    users_in_batch=None, items_in_batch=None, new_users=False, new_items=False, random_seed=None):
        """One stochastic update from ALL the interactions of a subset of users (batch_type='users')
        or of items ('items').  Requires reindex=False and keep_all_objs=True; the first call on an
        unfitted object needs nusers/nitems (INIT:714-931).  Returns self."""
        if self.reindex:
            raise ValueError("'partial_fit' can only be called when using reindex=False.")
        if not self.keep_all_objs:
            raise ValueError("'partial_fit' can only be called when using keep_all_objs=True.")
        if self.keep_data:
            if hasattr(self, "seen"):
                warnings.warn("When using 'partial_fit', the list of items seen by each user is not updated "
                              "with the data passed here.")
            else:
                # the reference raises NameError here (INIT:799, undefined `msg`); the evident intent:
                warnings.warn("When fitting the model through 'partial_fit' without calling 'fit' beforehand, "
                              "'keep_data' will be forced to False.")
                self.keep_data = False

        assert batch_type in ('users', 'items')
        user_batch = batch_type == 'users'

        if nusers is None:
            nusers = getattr(self, "nusers", None)
            if nusers is None:
                raise ValueError("Must specify total number of users when calling 'partial_fit' for the first time.")
        if nitems is None:
            nitems = getattr(self, "nitems", None)
            if nitems is None:
                raise ValueError("Must specify total number of items when calling 'partial_fit' for the first time.")
        if getattr(self, "nusers", None) is None:
            self.nusers = nusers
        if getattr(self, "nitems", None) is None:
            self.nitems = nitems

        if step_size is None:
            # INIT:834-847: schedule(niter) once the model has an iteration count, else 1.0
            try:
                self.step_size(0)
                schedule = self.step_size
            except Exception:
                schedule = lambda it: 1 / np.sqrt(it + 2)
            try:
                step_size = schedule(self.niter)
            except Exception:
                self.niter = 0
                step_size = 1.0
        assert 0 <= step_size <= 1

        if random_seed is not None:
            if isinstance(random_seed, float):
                random_seed = int(random_seed)
            assert isinstance(random_seed, int)

        if isinstance(counts_df, np.ndarray):
            counts_df = pd.DataFrame(counts_df[:, :3], copy=False, columns=_COLS)
        assert isinstance(counts_df, pd.DataFrame) and counts_df.shape[0] > 0
        for ccol in _COLS:
            assert ccol in counts_df.columns

        be = self._backend()
        req = ["ENSUREARRAY", "C_CONTIGUOUS"]
        Y_batch = np.require(counts_df["Count"], dtype=be.c_real_t, requirements=req)

        def ids_of(col):
            # (a signed 64-bit id column IS the reference's size_t column bit for bit for every valid id: no conversion
            #  pass over the batch -- a negative id fails the device's range check instead of wrapping around)
            v = counts_df[col].to_numpy(copy=False)
            if v.dtype == np.int64 and v.flags.c_contiguous:
                return v.view(np.uint64)
            return np.require(counts_df[col], dtype=be.obj_ind_type, requirements=req)
        ix_u_batch, ix_i_batch = ids_of("UserId"), ids_of("ItemId")
        # (INIT:864-871 takes np.unique of the batch's ids when the lists are not given: here they fall out of the
        #  grouping the step builds on the device anyway -- svi.partial_fit_device)
        if users_in_batch is not None:
            users_in_batch = np.require(users_in_batch, dtype=be.obj_ind_type, requirements=req)
        if items_in_batch is not None:
            items_in_batch = np.require(items_in_batch, dtype=be.obj_ind_type, requirements=req)

        if self._state.host.get("Theta") is None or self._state.host.get("Beta") is None:
            self._cast_before_fit()
            st = self._state
            temp = be.initialize_parameters(st.host["Theta"], st.host["Beta"], self.random_seed, self.a, self.a_prime,
                                            self.b_prime, self.c, self.c_prime, self.d_prime)
            for name, arr in zip(("Gamma_shp", "Gamma_rte", "Lambda_shp", "Lambda_rte", "k_rte", "t_rte"), temp):
                st.set_host(name, arr, private=True)
            st.set_host("Theta", temp[0] / temp[1], private=True)
            st.set_host("Beta", temp[2] / temp[3], private=True)

        if new_users:
            n_add = self.nusers - (ix_u_batch.max() + 1)
            if n_add < 1:
                raise ValueError("There are no new users in the data passed to 'partial_fit'.")
            self._initialize_extra_users(n_add, random_seed)
            self.nusers += n_add
        if new_items:
            n_add = self.nitems - (ix_i_batch.max() + 1)
            if n_add < 1:
                raise ValueError("There are no new items in the data passed to 'partial_fit'.")
            self._initialize_extra_items(n_add, random_seed)
            self.nitems += n_add

        k_shp = be.cast_real_t(self.a_prime + self.k * self.a)
        t_shp = be.cast_real_t(self.c_prime + self.k * self.c)
        add_k_rte = be.cast_real_t(self.a_prime / self.b_prime)
        add_t_rte = be.cast_real_t(self.c_prime / self.d_prime)
        # sic (INIT:912): the multiplier uses the user counts for item batches too (None: the device counts the users)
        multiplier_batch = None if users_in_batch is None else be.cast_real_t(float(nusers) / users_in_batch.shape[0])

        # the same step as the extension's partial_fit (be.partial_fit, PXI:423-473), on the state that stays on the
        # device between calls: only the batch crosses PCIe; host copies are refreshed when somebody reads them
        from . import svi
        m = self._state.ensure_model(be._make_ops(), lazy_ok=True)
        svi.partial_fit_device(m, Y_batch, ix_u_batch, ix_i_batch, add_k_rte, add_t_rte, self.a, self.c, k_shp, t_shp,
                               users_in_batch, items_in_batch, be.cast_real_t(step_size), multiplier_batch, user_batch,
                               nusers_total=nusers)
        self._state.touched()
        self.niter += 1
        self.is_fitted = True
        return self

    def _fresh_rows(self, n, seed, prime, scalar_rate):
        """shape/rate/factor/scalar-rate rows for n late-coming users or items (INIT:933-963):
        default_rng stream, shape drawn first."""
        rng = np.random.default_rng(seed=seed if seed > 0 else None)
        shp = prime + 0.01 * rng.random(size=(n, self.k), dtype=np.float32)
        rte = prime + 0.01 * rng.random(size=(n, self.k), dtype=np.float32)
        sc = np.full((n, 1), scalar_rate, dtype=np.float32)
        return shp, rte, shp / rte, sc

    def _initialize_extra_users(self, n, seed):
        shp, rte, fac, sc = self._fresh_rows(n, seed, self.a_prime, self.b_prime)
        self.k_rte = np.r_[self.k_rte, sc]
        self.Theta = np.r_[self.Theta, fac]
        self.Gamma_rte = np.r_[self.Gamma_rte, rte]
        self.Gamma_shp = np.r_[self.Gamma_shp, shp]

    def _initialize_extra_items(self, n, seed):
        shp, rte, fac, sc = self._fresh_rows(n, seed, self.c_prime, self.d_prime)
        self.t_rte = np.r_[self.t_rte, sc]
        self.Beta = np.r_[self.Beta, fac]
        self.Lambda_rte = np.r_[self.Lambda_rte, rte]
        self.Lambda_shp = np.r_[self.Lambda_shp, shp]

    # ------------------------------------------------------------------------------------------
    @staticmethod
    def _check_input_predict_factors(ncores, random_seed, stop_thr, maxiter):
        ncores = _resolve_ncores(ncores)
        assert isinstance(random_seed, int) and random_seed > 0
        if isinstance(stop_thr, int):
            stop_thr = float(stop_thr)
        assert isinstance(stop_thr, float) and stop_thr > 0
        if isinstance(maxiter, float):
            maxiter = int(maxiter)
        assert isinstance(maxiter, int) and maxiter > 0
        return ncores, random_seed, stop_thr, maxiter

    def _fold_in(self, counts_df, maxiter, ncores, random_seed, stop_thr, return_all):
        be = self._backend()
        Theta = np.empty(self.k, dtype=be.c_real_t)
        # item tables: the resident device copies (uploaded once, re-used by topN and the next fold-in)
        m = self._state.ensure_model(be._make_ops(), ("Beta", "Lambda_shp", "Lambda_rte"))
        temp = be.calc_user_factors(
            self.a, self.a_prime, self.b_prime, self.c, self.c_prime, self.d_prime,
            self._col(counts_df, "Count", be.c_real_t), self._col(counts_df, "ItemId", be.obj_ind_type),
            Theta, None, None, None,
            be.cast_ind_type(counts_df.shape[0]), be.cast_ind_type(self.k), be.cast_int(int(maxiter)),
            be.cast_int(ncores), be.cast_int(int(random_seed)), stop_thr, be.cast_int(bool(return_all)), resident=m)
        if np.isnan(Theta).any():
            raise ValueError("NaNs encountered in the result. Failed to produce latent factors.")
        return Theta, temp

    def predict_factors(self, counts_df, maxiter=10, ncores=1, random_seed=1, stop_thr=1e-3, return_all=False):
        """Latent factors of one (new) user from her (ItemId, Count) rows, item parameters fixed
        (INIT:989-1058).  With return_all=True returns (Theta, Gamma_shp, Gamma_rte, Phi)."""
        ncores, random_seed, stop_thr, maxiter = self._check_input_predict_factors(ncores, random_seed, stop_thr, maxiter)
        counts_df = self._process_data_single(counts_df)
        be = self._backend()
        Theta, temp = self._fold_in(counts_df, maxiter, ncores, random_seed, be.cast_real_t(stop_thr), return_all)
        if return_all:
            return (Theta, temp[0], temp[1], temp[2])
        return Theta

    def add_user(self, user_id, counts_df, update_existing=False, maxiter=10, ncores=1, random_seed=1, stop_thr=1e-3,
                 update_all_params=None):
        """Add one user (or refresh an existing one with update_existing=True) without refitting
        (INIT:1060-1196).  Returns True."""
        ncores, random_seed, stop_thr, maxiter = self._check_input_predict_factors(ncores, random_seed, stop_thr, maxiter)
        if update_existing:
            if self.produce_dicts and self.reindex:
                user_id = self.user_dict_[user_id]
            elif self.reindex:
                user_id = _codes(np.array([user_id]), self.user_mapping_)[0]
                if user_id == -1:
                    raise ValueError("User was not present in the training data.")
        counts_df = self._process_data_single(counts_df)
        be = self._backend()

        if update_all_params:
            counts_df['UserId'] = user_id
            counts_df['UserId'] = np.require(counts_df["UserId"], dtype=be.obj_ind_type)
            self.partial_fit(counts_df, new_users=(not update_existing))
            Theta_prev = self._state.rows("Theta", -1)[0].copy()       # (one row: the state stays on the device)
            for _ in range(maxiter - 1):
                self.partial_fit(counts_df)
                last = self._state.rows("Theta", -1)[0]
                if np.linalg.norm(last - Theta_prev) <= stop_thr:
                    break
                Theta_prev = last.copy()
        else:
            # sic (INIT:1155): the reference passes cast_int(stop_thr) == 0, i.e. never stops early
            Theta, temp = self._fold_in(counts_df, maxiter, ncores, random_seed, float(be.cast_int(stop_thr)),
                                        self.keep_all_objs)
            if self.keep_all_objs:
                g_shp, g_rte = temp[0].reshape((1, -1)), temp[1].reshape((1, -1))
                new_k_rte = self.a_prime / self.b_prime + (g_shp / g_rte).sum(axis=1, keepdims=True)
            if update_existing:
                self.Theta[user_id] = Theta
                if self.keep_all_objs:
                    self.Gamma_shp[user_id] = temp[0]
                    self.Gamma_rte[user_id] = temp[1]
                    self.k_rte[user_id] = new_k_rte
            else:
                if self.reindex:
                    new_pos = self.user_mapping_.shape[0]
                    self.user_mapping_ = np.r_[self.user_mapping_, np.array(user_id)]
                    if self.produce_dicts:
                        self.user_dict_[user_id] = new_pos
                self.Theta = np.r_[self.Theta, Theta.reshape((1, self.k))]
                if self.keep_all_objs:
                    self.Gamma_shp = np.r_[self.Gamma_shp, g_shp]
                    self.Gamma_rte = np.r_[self.Gamma_rte, g_rte]
                    self.k_rte = np.r_[self.k_rte, new_k_rte]
                self.nusers += 1

        if self.keep_data:
            items_now = counts_df["ItemId"].to_numpy(copy=False)
            if update_existing:
                before = self._n_seen_by_user[user_id]
                self._n_seen_by_user[user_id] = counts_df.shape[0]
                # sic (INIT:1189): splices at position user_id of the flat `seen` array
                self.seen = np.r_[self.seen[:user_id], items_now, self.seen[(user_id + 1):]]
                self._st_ix_user[(user_id + 1):] += self._n_seen_by_user[user_id] - before
            else:
                self._n_seen_by_user = np.r_[self._n_seen_by_user, np.array(counts_df.shape[0])]
                self._st_ix_user = np.r_[self._st_ix_user, self.seen.shape[0]]
                self.seen = np.r_[self.seen, items_now]
        return True

    # ------------------------------------------------------------------------------------------
    def _lookup(self, ids, mapping, table):
        """external id(s) -> internal positions as an int array; -1 for unknown ids (INIT:1221-1269)."""
        if not np.isscalar(ids):
            ids = np.require(ids, requirements=["ENSUREARRAY"]).reshape(-1)
            assert ids.shape[0] > 0
            if not self.reindex:
                return ids
            if ids.shape[0] > 1:
                return self._codes_of(ids, "user" if mapping is self.user_mapping_ else "item")
            ids = ids[0]
        if self.reindex:
            if table is not None:
                try:
                    ids = table[ids]
                except Exception:
                    ids = -1
            else:
                ids = _codes(np.array([ids]), mapping)[0]
        return np.array([ids])

    def predict(self, user, item):
        """Predicted count(s) Theta_u . Beta_i for one pair or for aligned arrays of pairs; NaN for
        ids not seen in training (INIT:1198-1293)."""
        assert self.is_fitted
        user = self._lookup(user, self.user_mapping_, self.user_dict_)
        item = self._lookup(item, self.item_mapping_, self.item_dict_)
        assert user.shape[0] == item.shape[0]
        st = self._state
        if user.shape[0] == 1:
            if user[0] == -1 or item[0] == -1:
                return np.nan
            return st.rows("Theta", user).dot(st.rows("Beta", item).T).reshape(-1)[0]
        unknown = (user == -1) | (item == -1)
        if not unknown.any():
            return self._predict_pairs(user, item)
        out = np.full(user.shape[0], np.nan, dtype=np.float32)
        if (~unknown).any():
            out[~unknown] = self._predict_pairs(user[~unknown], item[~unknown])
        return out

    def __getstate__(self):
        """Pickles carry host data only (loadable on a machine without this GPU): the device-backed `seen` list is
        brought down, the device-side id lookups and triplets are dropped (they are caches / fit-time scratch), the
        state tables pickle through ResidentState's own host-array form."""
        d = dict(self.__dict__)
        if d.get("_devbacked_seen") is not None:
            d["_devbacked_seen"] = [type(self).seen.__get__(self), None]
        for n in ("_id_lookup", "_dev_triplets", "_tick_t"):
            d.pop(n, None)
        return d

    def _pair_tables(self, n_pairs):
        """(Theta, Beta) operands for n_pairs listed pairs: the resident device tables when they are current or the
        pairs are many, else the host arrays (the backend then ships only the rows the pairs touch)."""
        st = self._state
        be = self._backend()
        rows = int(st.host["Theta"].shape[0]) + int(st.host["Beta"].shape[0])
        if (st.on_device("Theta") and st.on_device("Beta")) or 2 * n_pairs >= rows:
            ops = be._make_ops()
            return st.table(ops, "Theta"), st.table(ops, "Beta"), True
        return st.peek_host("Theta"), st.peek_host("Beta"), False

    def _predict_pairs(self, user, item):
        be = self._backend()
        req = ["ENSUREARRAY", "C_CONTIGUOUS"]
        user = np.require(user, dtype=be.obj_ind_type, requirements=req)
        item = np.require(item, dtype=be.obj_ind_type, requirements=req)
        T, B, on_dev = self._pair_tables(user.shape[0])
        if on_dev:
            return be.pair_dots_device(T, B, user, item, self.k)
        return be.predict_arr(T, B, user, item, self.ncores)

    def _seen_by(self, user, device=False):
        """The items `user` had in the training data; device=True: as a device tensor when the list lives there."""
        # int(): after an SVI fit the start index is a size_t array, and uint64 + int32 is float64 in numpy
        st = int(self._st_ix_user[user])
        en = st + int(self._n_seen_by_user[user])
        if device:
            on_dev = type(self).seen.device_of(self)
            if on_dev is not None:
                return on_dev[st:en]
        return self.seen[st:en]

    def topN(self, user, n=10, exclude_seen=True, items_pool=None):
        """The n items with the highest predicted count for `user`, best first; optionally without
        the items she had in the training data, optionally restricted to `items_pool`
        (INIT:1296-1396)."""
        if isinstance(n, float):
            n = int(n)
        assert isinstance(n, int)
        if self.reindex:
            unknown = "Can only predict for users who were in the training set."
            if self.produce_dicts:
                try:
                    user = self.user_dict_[user]
                except Exception:
                    raise ValueError(unknown)
            else:
                user = _codes(np.array([user]), self.user_mapping_)[0]
                if user == -1:
                    raise ValueError(unknown)
        if exclude_seen and not self.keep_data:
            raise Exception("Can only exclude seen items when passing 'keep_data=True' to .fit")

        def back(ids):
            return self.item_mapping_[ids] if self.reindex else ids

        if items_pool is None:
            # device path: GEMV over the (cached) item table + mask + top-k; same ids as the reference's
            # argpartition/setdiff1d/argsort sequence (INIT:1337-1356), ties aside
            be = self._backend()
            st = self._state
            # (the user's row and her seen items are taken on the device when they are there: no round trip)
            row = st.device_row("Theta", user)
            rec = be.top_items(st.rows("Theta", user)[0] if row is None else row, st.table(be._make_ops(), "Beta"), n,
                               self._seen_by(user, device=True) if exclude_seen else None)
            return back(rec)

        items_pool = np.require(items_pool, requirements=["ENSUREARRAY"]).reshape(-1)
        pool = items_pool
        if self.reindex:
            pool = _codes(items_pool, self.item_mapping_)
            missing = pool == -1
            if missing.any():
                pool = pool[~missing]
                warnings.warn("There were %d entries from 'item_pool'"
                              "that were not in the training data and will be exluded." % int(missing.sum()))
            if pool.shape[0] == 0:
                raise ValueError("No items to recommend.")
            if pool.shape[0] == 1:
                raise ValueError("Only 1 item to recommend.")
        theta_u = self._state.rows("Theta", user)[0]
        neg = -theta_u.dot(self._state.rows("Beta", pool).T)
        n = int(min(n, items_pool.shape[0]))
        if exclude_seen:
            n_ext = int(min(n + self._n_seen_by_user[user], items_pool.shape[0]))
            cand = np.argpartition(neg, n_ext - 1)[:n_ext]
            cand = np.setdiff1d(pool[cand], self._seen_by(user))
            neg = -theta_u.dot(self._state.rows("Beta", cand).T)
            return back(cand[np.argsort(neg)[:n]])
        cand = np.argpartition(neg, n - 1)[:n]
        return items_pool[cand[np.argsort(neg[cand])]]

    def eval_llk(self, input_df, full_llk=False):
        """Poisson log-likelihood (plus a data-only constant unless full_llk) of the listed
        observations, over the pairs whose user and item were in the training data
        (INIT:1399-1446).  Returns {'llk': ..., 'nobs': ...}."""
        assert self.is_fitted
        self._process_valset(input_df, valset=False)
        be = self._backend()
        self.ncores = be.cast_int(self.ncores)
        yv = self._col(self.val_set, "Count", be.c_real_t)
        uv = self._col(self.val_set, "UserId", be.obj_ind_type)
        iv = self._col(self.val_set, "ItemId", be.obj_ind_type)
        T, B, on_dev = self._pair_tables(yv.shape[0])
        if on_dev:
            llk = be.calc_llk_device(yv, uv, iv, T, B, self.k, bool(full_llk))
        else:
            llk = be.calc_llk(yv, uv, iv, T, B, self.k, self.ncores, be.cast_int(bool(full_llk)))
        out = {'llk': llk, 'nobs': self.val_set.shape[0]}
        del self.val_set
        return out

    @staticmethod
    def _print_st_msg():
        print("**********************************")
        print("Hierarchical Poisson Factorization")
        print("**********************************")
        print("")

    def _print_data_info(self):
        print("Number of users: %d" % self.nusers)
        print("Number of items: %d" % self.nitems)
        print("Latent factors to use: %d" % self.k)
        print("")
