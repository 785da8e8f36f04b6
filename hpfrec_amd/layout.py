"""COO triplets -> the sparse layouts the sweep kernel consumes.

The reference keeps the data as unsorted COO triplets (fit_hpf signature, cython_loops.pxi:147-151)
and, for SVI only, a CSR start-index vector (hpfrec/__init__.py:587-606) and a scipy CSC copy
(cython_loops.pxi:22-25).  The HIP path wants, per side, the nonzeros grouped by row and cut into
bounded *segments* (include/hpf_hip.h: hpf_segment) so that one wavefront never owns more than
SEG_CAP nonzeros.  Duplicate (user,item) pairs stay separate observations, exactly as the
reference treats them in full-batch mode (scipy's tocsr() would merge them; we never call it).

Everything here is index plumbing on torch tensors (any device, incl. CPU for the tests).
"""
import os

import numpy as np
import torch

# nonzeros per segment (256 steps of 4 nonzeros at ld=64).  Measured at C3, ms/iteration: 64: 4.09, 128: 3.87,
# 256: 3.77, 512: 3.74, 1024: 3.70-3.72, 2048-4096: 3.73-3.74, 16384: 3.84 (tail imbalance).
SEG_CAP = int(os.environ.get("HPF_SEG_CAP", "1024"))
SHORT_ROW_NNZ = int(os.environ.get("HPF_SHORT_ROW_NNZ", "24"))      # average nonzeros per segment below which the plain sweep is launched with its short-row hint
SEG_LEN_MASK = 0x00FFFFFF    # include/hpf_hip.h: HPF_SEG_LEN_MASK
SEG_WHOLE_ROW = 0x40000000   # include/hpf_hip.h: HPF_SEG_WHOLE_ROW


class SparseSide:
    """Nonzeros grouped by the rows of one side (users -> CSR, items -> CSC)."""

    def __init__(self, nrows, indptr, idx, y, seg_cap=None):
        self.nrows = int(nrows)
        self.indptr = indptr            # int64 [nrows+1]
        self.idx = idx                  # int32 [nnz] row ids of the *other* side
        self.y = y                      # float32 [nnz]
        self.nnz = int(idx.shape[0])
        self.segs, self.row_seg_ptr = build_segments(indptr, seg_cap)
        self.nseg = int(self.segs.shape[0])
        # rows the fused sweep cannot finish on its own: split rows and rows without any nonzero
        nseg_row = self.row_seg_ptr[1:] - self.row_seg_ptr[:-1]
        self.multi_rows = torch.nonzero(nseg_row != 1).reshape(-1).contiguous()
        self.nmulti = int(self.multi_rows.shape[0])


def build_segments(indptr, seg_cap=None):
    """Cut rows into segments of at most seg_cap nonzeros.

    Returns (segs, row_seg_ptr): segs is an int64 [nseg,2] tensor whose memory image is an array
    of hpf_segment {int64 begin; int32 len|flags; int32 row} (little endian: len | flags | row<<32), and
    row_seg_ptr [nrows+1] lists each row's segment range (empty rows have no segment).
    """
    seg_cap = SEG_CAP if seg_cap is None else int(seg_cap)     # (the module's value at CALL time)
    dev = indptr.device
    deg = indptr[1:] - indptr[:-1]
    nseg_row = (deg + (seg_cap - 1)) // seg_cap
    row_seg_ptr = torch.zeros(deg.shape[0] + 1, dtype=torch.int64, device=dev)
    torch.cumsum(nseg_row, 0, out=row_seg_ptr[1:])
    nseg = int(row_seg_ptr[-1].item()) if deg.shape[0] > 0 else 0
    rows = torch.repeat_interleave(torch.arange(deg.shape[0], dtype=torch.int64, device=dev), nseg_row,
                                   output_size=nseg)
    within = torch.arange(nseg, dtype=torch.int64, device=dev) - row_seg_ptr[rows]
    begin = indptr[rows] + within * seg_cap
    length = torch.clamp(deg[rows] - within * seg_cap, max=seg_cap)
    flags = torch.where(nseg_row[rows] == 1, SEG_WHOLE_ROW, 0)
    segs = torch.stack([begin, length | flags | (rows << 32)], dim=1).contiguous()
    return segs, row_seg_ptr


def _indptr(sorted_rows, nrows):
    """Row pointers of ids that are SORTED: indptr[r] = how many ids are < r -- one binary search per row.  (A histogram +
    scan does the same with one atomic add per nonzero; on sorted ids neighbouring threads all hit the same counter, and
    with several ranks sharing a GPU torch.bincount stalled for minutes in bench.py's exchange autotune.)"""
    bounds = torch.arange(nrows + 1, dtype=sorted_rows.dtype, device=sorted_rows.device)
    return torch.searchsorted(sorted_rows.contiguous(), bounds, right=False).to(torch.int64)


def ids_to_device(a, dev):
    """Host id array -> int64 device tensor, without a host-side conversion pass when the ids are 64-bit already:
    the reference's size_t ids are reinterpreted (an id >= 2^63 comes out negative and fails the range checks)."""
    a = np.ascontiguousarray(a)
    if a.dtype == np.uint64:
        a = a.view(np.int64)
    elif a.dtype != np.int64:
        a = a.astype(np.int64)
    if not a.flags.writeable:
        a = a.copy()            # (torch.from_numpy wants a writable buffer; never written here)
    return torch.from_numpy(a).to(dev)


def build_sides(ix_u, ix_i, y, nU, nI, seg_cap=None):
    """(user side, item side, ix_u sorted) from COO triplets (int64 ids, float32 counts).

    User side: sorted by (user, item).  Item side: a stable re-sort of that order by item, so users
    ascend within every item column (gathers walk the user table in address order).
    """
    ix_u = ix_u.to(torch.int64)
    ix_i = ix_i.to(torch.int64)
    y = y.to(torch.float32)
    if ix_u.numel() > 0:
        if int(ix_u.max()) >= nU or int(ix_i.max()) >= nI or int(ix_u.min()) < 0 or int(ix_i.min()) < 0:
            raise ValueError("user/item index out of range")
    key = ix_u * int(nI) + ix_i
    order = torch.argsort(key, stable=True)
    u_s, i_s, y_s = ix_u[order], ix_i[order], y[order]
    users = SparseSide(nU, _indptr(u_s, nU), i_s.to(torch.int32), y_s, seg_cap)
    order2 = torch.argsort(i_s, stable=True)
    items = SparseSide(nI, _indptr(i_s[order2], nI), u_s[order2].to(torch.int32), y_s[order2], seg_cap)
    return users, items, u_s.to(torch.int32)


def nnz_balanced_ranges(indptr, nparts):
    """Contiguous row ranges with (nearly) equal nonzero counts: list of (row_begin, row_end).
    Used to shard users over GPUs (balance nnz, not user counts)."""
    nrows = indptr.shape[0] - 1
    total = int(indptr[-1].item())
    bounds = [0]
    for p in range(1, nparts):
        target = (total * p) // nparts
        r = int(torch.searchsorted(indptr, torch.tensor([target], dtype=indptr.dtype, device=indptr.device),
                                   right=False).item())
        r = max(bounds[-1], min(r, nrows))
        bounds.append(r)
    bounds.append(nrows)
    return [(bounds[p], bounds[p + 1]) for p in range(nparts)]
