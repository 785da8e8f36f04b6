"""Id handling and interaction metadata on the device -- what `HPF.fit` does with pandas/scipy before and after the
optimisation in the reference (/root/reference/hpfrec/__init__.py, "INIT"):

  * the `Count <= thr` filter                                              INIT:462-475
  * first-appearance renumbering of user / item ids (`pd.factorize`)      INIT:478-479
  * who-saw-what CSR for `topN(exclude_seen=True)` and the SVI user index  INIT:587-606 (scipy coo -> csr: duplicate
    pairs merged, item ids ascending inside a row)

All of it is sorting / run-length / prefix-sum work on torch tensors (any device; the tests also run it on CPU
tensors against pandas and scipy).  Ids that are not numbers (strings, objects) cannot live on the device: the
caller falls back to pandas for those.
"""
import numpy as np
import torch


def to_device_ids(values, device):
    """Host id column -> device tensor, or None when the dtype cannot be factorized on the device (non-numeric ids,
    floats with NaNs -- pandas gives those the sentinel -1, a case left to pandas)."""
    a = np.asarray(values)
    if a.dtype.kind in "iu":
        if a.dtype == np.uint64:
            if a.size and a.max() > np.iinfo(np.int64).max:
                return None
            a = a.astype(np.int64)
        return torch.from_numpy(np.ascontiguousarray(a).astype(np.int64, copy=False)).to(device)
    if a.dtype.kind == "f":
        if np.isnan(a).any():
            return None
        return torch.from_numpy(np.ascontiguousarray(a)).to(device)
    return None


def factorize(ids, with_sorted=False):
    """pd.factorize(ids) for a numeric device vector: (codes int64, uniques) with uniques in order of first
    appearance and codes[n] = position of ids[n] in uniques.  One stable sort + run detection + a sort of the runs'
    first positions.  with_sorted: also (the unique ids ascending, the code of each) -- a ready-made lookup table."""
    n = ids.shape[0]
    dev = ids.device
    if n == 0:
        e = torch.empty(0, dtype=torch.int64, device=dev)
        return (e, ids.clone(), ids.clone(), e) if with_sorted else (e, ids.clone())
    sval, order = torch.sort(ids, stable=True)                   # equal ids keep their input order
    uniq, inverse, counts = torch.unique_consecutive(sval, return_inverse=True, return_counts=True)
    run_start = torch.cumsum(counts, 0) - counts
    first_pos = order[run_start]                                 # stable sort: the run's first entry is the earliest
    by_appearance = torch.argsort(first_pos)                     # run index, ordered by first appearance
    code_of_run = torch.empty_like(by_appearance)
    code_of_run[by_appearance] = torch.arange(by_appearance.shape[0], device=dev)
    codes = torch.empty(n, dtype=torch.int64, device=dev)
    codes[order] = code_of_run[inverse]
    if with_sorted:
        return codes, uniq[by_appearance], uniq, code_of_run
    return codes, uniq[by_appearance]


class IdLookup:
    """pd.Categorical(values, mapping).codes for numeric ids on the device: position of each value in `mapping`
    (the first-appearance numbering of the fit), -1 for values that are not in it (INIT:561-563, 1221-1269).
    The mapping is sorted once; a query is a searchsorted + an equality check."""

    def __init__(self, mapping, device):
        m = to_device_ids(mapping, device)
        if m is None:
            raise ValueError("mapping is not numeric")
        self.sorted_vals, order = torch.sort(m)
        self.codes = order                       # position in the original (first-appearance) order
        self.device = device

    def __call__(self, values):
        v = to_device_ids(values, self.device)
        if v is None:
            return None
        if v.dtype != self.sorted_vals.dtype:
            v = v.to(self.sorted_vals.dtype) if self.sorted_vals.dtype.is_floating_point else None
            if v is None:
                return None
        n = self.sorted_vals.shape[0]
        if n == 0:
            return torch.full(v.shape, -1, dtype=torch.int64, device=self.device)
        pos = torch.searchsorted(self.sorted_vals, v).clamp_(max=n - 1)
        hit = self.sorted_vals[pos] == v
        return torch.where(hit, self.codes[pos], torch.full_like(pos, -1))


def seen_metadata(ix_u, ix_i, nU, nI):
    """(n_seen_by_user [nU], st_ix_user = indptr [nU+1], seen) of scipy's coo_array((.., (ix_u, ix_i))).tocsr():
    duplicate pairs counted once, item ids ascending inside a user's slice (INIT:589-605).  int64 device tensors."""
    dev = ix_u.device
    nU, nI = int(nU), int(nI)
    key = torch.unique(ix_u.to(torch.int64) * nI + ix_i.to(torch.int64))      # sorted, duplicates merged
    u = torch.div(key, nI, rounding_mode="floor")
    seen = key - u * nI
    n_seen = torch.bincount(u, minlength=nU)
    indptr = torch.zeros(nU + 1, dtype=torch.int64, device=dev)
    torch.cumsum(n_seen, 0, out=indptr[1:])
    return n_seen, indptr, seen


#: index dtype of the arrays scipy's coo -> csr returns for the reference's size_t id columns (INIT:589-599)
SEEN_INDEX_DTYPE = np.int64
