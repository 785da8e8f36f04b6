"""Stochastic (mini-batch) paths on the device: the SVI epochs of fit_hpf
(/root/reference/hpfrec/cython_loops.pxi:262-377, "PXI"), the Cython partial_fit (PXI:423-473) and the
single-user fold-in calc_user_factors (PXI:476-520) -- the "next" rows f1/f4 of SURVEY.md section 8.

Division of labour:
  * index structures of a batch (its rows' segment list, the same nonzeros grouped by the other side, flags of the rows both
    touch): built on the device, once per EPOCH (EpochWorkspace), per batch (BatchWorkspace) or from the caller's COO
    triplets (CooBatch, partial_fit) -- csrc/hpf_svi_prep.hip; nothing but a handful of sizes is ever read back;
  * one stochastic step (_svi_step, the reference's statement order): BOTH sides' statements are fused into the sweeps that
    form their phi-sums (sweep_kernel MODE 3: the batch side, E row in the prologue, row finished in the epilogue; MODE 2:
    the other side), a whole-table pass per side covers what the sweeps cannot finish (split rows; for the batch side every
    row outside the batch, whose rates and means the reference recomputes every batch, PXI:300,318 / 352,370);
  * the batch side's rate stays FACTORED (row scalar + column sums) and no mean table is stored between checks
    (DeviceModel.materialize expands them on demand, bit-identically).
torch is used for device memory, the two radix sorts of a COO batch and stream plumbing.  No numerics run on the host;
shuffles and seeds use numpy's generators so batches are the reference's.
"""
import os

import numpy as np
import torch

from . import _lib, _streams, layout

_NAMES = ("Gamma_shp", "Gamma_rte", "Lambda_shp", "Lambda_rte", "Theta", "Beta")
PF_TIMINGS = {}       # HPF_TIMING=1: {phase: [seconds per partial_fit_device call]}, device-synchronised
SVI_TIMINGS = {}      # HPF_TIMING=1: {"epochs": n, "seconds": wall time of the last fit's epoch loop}


def _reference_sums():
    """HPF_COLSUM_ORDER=reference: every Theta.sum(axis=0) / Beta.sum(axis=0) of a stochastic step (PXI:300,320 / 352,372;
    443-470) is formed in numpy's own order from the STORED mean table (hpf_hip_colsum_sequential_f32), as in the full-batch
    driver (cavi.py): the mode that holds the stochastic path against the reference itself without the summation-order noise
    of 1e5..1e6-row float32 sums.  Implies the stored form of the tables (HPF_SVI_LAZY=0)."""
    return os.environ.get("HPF_COLSUM_ORDER", "tree") == "reference"


class BatchSide:
    """Segments over the rows touched by a batch (same fields the sweep launcher reads from SparseSide), built with
    tensor-library sorts and host-side sizes.  No driver builds its batches this way any more (epochs: EpochWorkspace /
    BatchWorkspace; partial_fit: CooBatch -- all on the device, nothing read back): this is the plain construction the tests
    hold those against, and what the op-level tests feed the sweeps with."""

    def __init__(self, rows, cols, y, seg_cap=None, grouped=False):
        """rows/cols: int64 device tensors of a COO batch.  grouped=True: the triplets already come grouped by
        ascending `rows` (a sorted row list's nonzeros gathered out of a CSR) -- no sort needed; otherwise they are
        grouped here with a stable sort."""
        seg_cap = layout.SEG_CAP if seg_cap is None else int(seg_cap)
        if grouped:
            r_s, c_s, y_s = rows, cols, y
        else:
            order = torch.argsort(rows, stable=True)
            r_s, c_s, y_s = rows[order], cols[order], y[order]
        self.rows, counts = torch.unique_consecutive(r_s, return_counts=True)   # rows present, ascending
        self.idx = c_s.to(torch.int32).contiguous()
        self.y = y_s.to(torch.float32).contiguous()
        indptr = torch.zeros(self.rows.shape[0] + 1, dtype=torch.int64, device=rows.device)
        torch.cumsum(counts, 0, out=indptr[1:])
        segs, self.row_seg_ptr = layout.build_segments(indptr, seg_cap)
        if segs.shape[0] > 0:
            local = segs[:, 1] >> 32
            meta = segs[:, 1] & 0xFFFFFFFF      # length | whole-row flag
            segs = torch.stack([segs[:, 0], meta | (self.rows[local] << 32)], dim=1).contiguous()
        self.segs = segs
        self.nseg = int(segs.shape[0])
        self.nrows = int(self.rows.shape[0])
        self.short_rows = 1 if (self.nseg > 0 and
                                                   int(self.y.shape[0]) / self.nseg < layout.SHORT_ROW_NNZ) else 0
        nseg_row = self.row_seg_ptr[1:] - self.row_seg_ptr[:-1]
        self.multi_local = torch.nonzero(nseg_row > 1).reshape(-1)              # rows cut into several segments
        self.nmulti = int(self.multi_local.shape[0])

    def tensors(self):
        return [self.rows, self.idx, self.y, self.row_seg_ptr, self.segs, self.multi_local]


class DevSide:
    """One grouping of a batch BUILT ON THE DEVICE (BatchWorkspace), as the sweep launcher sees a side: `nseg` is the
    capacity of the segment list, the live count stays on the device (`nseg_dev`); split rows come as {first segment,
    segments, row} descriptors (`multi`, `nmulti_dev` of them)."""

    def __init__(self, segs, nseg_cap, idx, y, nseg_dev, multi, nmulti_dev, multi_cap, short_rows):
        self.segs, self.nseg, self.idx, self.y, self.nseg_dev = segs, int(nseg_cap), idx, y, nseg_dev
        self.multi, self.nmulti_dev, self.multi_cap, self.short_rows = multi, nmulti_dev, int(multi_cap), int(short_rows)


class BatchWorkspace:
    """Device buffers for the index structures of ONE stochastic batch drawn from the rows of `own` (users' CSR for a
    user batch, items' CSC for an item batch), filled by ops.svi_batch_prepare (hpf_hip_svi_batch_prepare: one host
    call, nothing read back).  Sized once per fit from a bound -- the nonzeros of the `batch_rows` largest rows of the
    side -- so that no batch can overflow it.  The reference slices the same things on the host per batch
    (PXI:280-290, 332-342, 27-42)."""

    def __init__(self, ops, own, oth, acc_own, ld, batch_rows, seg_cap=None):
        dev = own.idx.device
        i64 = dict(dtype=torch.int64, device=dev)
        seg_cap = layout.SEG_CAP if seg_cap is None else int(seg_cap)
        self.own, self.oth, self.acc_own, self.ld, self.seg_cap = own, oth, acc_own, int(ld), int(seg_cap)
        B = int(min(batch_rows, own.nrows))
        deg = own.indptr[1:] - own.indptr[:-1]
        bound = int(torch.topk(deg, B).values.sum().item()) if B > 0 else 0       # (once per fit)
        self.nnz_bound = bound
        self.b_cap = B + bound // seg_cap + 1
        self.multi_cap = bound // seg_cap + 2
        self.o_cap = max(bound, 1)
        self.o_segs_cap = min(oth.nrows, bound) + bound // seg_cap + 1
        self.flag_own = torch.zeros(own.nrows, dtype=torch.uint8, device=dev)
        self.flag_oth = torch.zeros(oth.nrows, dtype=torch.uint8, device=dev)
        self.b_segs = torch.zeros((self.b_cap, 2), **i64)
        self.b_multi = torch.zeros((self.multi_cap, 3), **i64)
        self.o_multi = torch.zeros((self.multi_cap, 3), **i64)
        self.o_idx = torch.zeros(self.o_cap, dtype=torch.int32, device=dev)
        self.o_y = torch.zeros(self.o_cap, dtype=torch.float32, device=dev)
        self.o_segs = torch.zeros((self.o_segs_cap, 2), **i64)
        self.sizes = torch.zeros(8, **i64)
        # scratch of the other side's filter: a keep bit per nonzero of the side, a count and an offset per 1024 of them
        ntiles = (oth.nnz + 1023) // 1024
        self.mask = torch.zeros(max(1, 16 * ntiles), **i64)
        self.chunk_pre = torch.zeros(max(1, 16 * ntiles), dtype=torch.int16, device=dev)
        self.flag_bits = torch.zeros((own.nrows + 31) // 32, dtype=torch.int32, device=dev)
        self.tile_cnt = torch.zeros(ntiles + 1, dtype=torch.int32, device=dev)
        self.tile_off = torch.zeros(ntiles + 1, **i64)
        self.row_start = torch.zeros(oth.nrows, **i64)
        self.row_cnt = torch.zeros(oth.nrows, dtype=torch.int32, device=dev)
        self.tiles = torch.zeros(ops.svi_prep_scratch_words(), **i64)
        self.ids = self.prev_ids = None
        self.ready = self.free = None         # events: structures built (preparation stream) / consumed (compute stream)
        # own side: descriptors index the side's GLOBAL idx / y; rows of a batch average tens of nonzeros -> no hint.
        # other side: the batch's nonzeros spread over many rows, a handful each -> the short-row launch
        self.side_own = DevSide(self.b_segs, self.b_cap, own.idx, own.y, self.sizes[0:1], self.b_multi, self.sizes[1:2],
                                self.multi_cap, 0)
        self.side_oth = DevSide(self.o_segs, self.o_segs_cap, self.o_idx, self.o_y, self.sizes[2:3], self.o_multi,
                                self.sizes[3:4], self.multi_cap, 1)

    def prepare(self, ops, ids):
        """Builds the structures of the batch made of rows `ids` (device int64) on the current stream."""
        self.prev_ids, self.ids = self.ids, ids
        ops.svi_batch_prepare(self)

    def overflowed(self):
        return bool(int(self.sizes[7].item()) != 0)


class EpochWorkspace:
    """Device buffers for the index structures of EVERY batch of an epoch over the rows of `own`, filled once per epoch
    by ops.svi_epoch_prepare (hpf_hip_svi_epoch_prepare) from the epoch's shuffled order: an epoch's batches partition the
    side's rows, so the other side's nonzeros are partitioned by batch in one pass (e_idx / e_y) instead of being filtered
    once per batch, and a batch is a set of fixed-capacity slices -- `batch(j)` hands them out, no device work.  Capacities
    per batch come from the same bound as BatchWorkspace's (the nonzeros of the `batch_rows` largest rows)."""

    MAX_BATCHES = 255                       # (a batch id is a byte)

    @staticmethod
    def plan(own, oth, batch_rows, seg_cap=None):
        """(nb, per, bound, bytes of the per-batch slices) of an epoch over `own` in batches of `batch_rows` rows."""
        seg_cap = layout.SEG_CAP if seg_cap is None else int(seg_cap)
        per = int(min(batch_rows, own.nrows))
        nb = -(-own.nrows // per)
        # (the bound costs a top-k and a read-back: once per side and batch size -- fits() and the constructor share it)
        cache = own.__dict__.setdefault("_epoch_bounds", {})
        if per not in cache:
            deg = own.indptr[1:] - own.indptr[:-1]
            cache[per] = int(torch.topk(deg, per).values.sum().item()) if per > 0 else 0
        bound = cache[per]
        o_segs_cap = min(oth.nrows, bound) + bound // seg_cap + 1
        b_cap = per + bound // seg_cap + 1
        multi_cap = bound // seg_cap + 2
        # per batch: the two segment lists, the two flag rows, the split-row lists, the {batch, segment} counts and offsets;
        # per epoch: e_idx / e_y / key; + the sweeps' part[] scratch of the other side's segment list ([o_segs_cap][ld]: the
        # caller's, counted by fits())
        nbytes = nb * (16 * (o_segs_cap + b_cap) + own.nrows + oth.nrows + 12 * oth.nseg + 48 * multi_cap) + 9 * oth.nnz
        return nb, per, bound, nbytes

    @classmethod
    def fits(cls, own, oth, batch_rows, seg_cap=None):
        """An epoch-level preparation is used when a batch id fits a byte and the per-batch slices stay a modest share of
        the device (HPF_SVI_EPOCH_BYTES, default 16 GiB); otherwise the batches are prepared one by one (BatchWorkspace)."""
        if os.environ.get("HPF_SVI_EPOCH_PREP", "1") != "1" or batch_rows <= 0:
            return False
        nb, _, _, nbytes = cls.plan(own, oth, batch_rows, seg_cap)
        budget = int(os.environ.get("HPF_SVI_EPOCH_BYTES", str(16 << 30)))
        dev = own.idx.device
        if dev.type == "cuda" and "HPF_SVI_EPOCH_BYTES" not in os.environ:
            # (a smaller device: at most a quarter of what is free now -- an SVI over one side alternates TWO workspaces --
            #  otherwise the batches are prepared one by one)
            budget = min(budget, torch.cuda.mem_get_info(dev)[0] // 4)
        return nb <= cls.MAX_BATCHES and nbytes <= budget

    def __init__(self, ops, own, oth, acc_own, ld, batch_rows, seg_cap=None):
        dev = own.idx.device
        i64 = dict(dtype=torch.int64, device=dev)
        u8 = dict(dtype=torch.uint8, device=dev)
        seg_cap = layout.SEG_CAP if seg_cap is None else int(seg_cap)
        self.own, self.oth, self.acc_own, self.ld, self.seg_cap = own, oth, acc_own, int(ld), int(seg_cap)
        self.nb, self.per, bound, _ = self.plan(own, oth, batch_rows, seg_cap)
        assert 1 <= self.nb <= self.MAX_BATCHES
        nb = self.nb
        self.nnz_bound = bound
        self.b_cap = self.per + bound // seg_cap + 1
        self.multi_cap = bound // seg_cap + 2
        self.o_segs_cap = min(oth.nrows, bound) + bound // seg_cap + 1
        self.batch_of = torch.zeros(own.nrows, **u8)
        self.flag_own = torch.zeros((nb, own.nrows), **u8)
        self.flag_oth = torch.zeros((nb, oth.nrows), **u8)
        self.b_segs = torch.zeros((nb, self.b_cap, 2), **i64)
        self.b_multi = torch.zeros((nb, self.multi_cap, 3), **i64)
        self.o_multi = torch.zeros((nb, self.multi_cap, 3), **i64)
        self.o_segs = torch.zeros((nb, self.o_segs_cap, 2), **i64)
        self.e_idx = torch.zeros(max(oth.nnz, 1), dtype=torch.int32, device=dev)
        self.e_y = torch.zeros(max(oth.nnz, 1), dtype=torch.float32, device=dev)
        self.sizes = torch.zeros((nb, 8), **i64)
        self.key = torch.zeros(max(oth.nnz, 1), **u8)
        self.seg_cnt = torch.zeros(nb * oth.nseg + 1, dtype=torch.int32, device=dev)
        self.seg_pos = torch.zeros(nb * oth.nseg + 1, **i64)
        self.tiles = torch.zeros(ops.svi_epoch_scratch_words(nb), **i64)
        self.order = None
        self.ready = self.free = None         # events: structures built (preparation stream) / consumed (compute stream)
        # own side: descriptors index the side's GLOBAL idx / y (no hint); other side: a handful of nonzeros per row
        self._sides = [(DevSide(self.b_segs[j], self.b_cap, own.idx, own.y, self.sizes[j, 0:1], self.b_multi[j],
                                self.sizes[j, 1:2], self.multi_cap, 0),
                        DevSide(self.o_segs[j], self.o_segs_cap, self.e_idx, self.e_y, self.sizes[j, 2:3], self.o_multi[j],
                                self.sizes[j, 3:4], self.multi_cap, 1)) for j in range(nb)]

    def prepare(self, ops, order):
        """Builds the structures of all batches of the epoch whose shuffled rows are `order` (device int64, a permutation
        of the side's rows) on the current stream."""
        self.order = order
        ops.svi_epoch_prepare(self)

    def batch(self, j):
        """(own side, other side, flags of the batch's rows, flags of the other side's rows it touches) of batch j."""
        return self._sides[j][0], self._sides[j][1], self.flag_own[j], self.flag_oth[j]

    def overflowed(self):
        return bool(int(self.sizes[:, 7].sum().item()) != 0)


class DeviceModel:
    """The variational state as padded device tables; `v(name)` is the [:, :k] view used by the dense algebra."""

    def __init__(self, ops, k, nU, nI, tables=None):
        """`tables`: padded device tensors ([n, ld] tables, [n] scalar rates; zero pad columns) to take over instead of
        allocating -- the state a full-batch fit leaves on the device (the arrays it names must all be given)."""
        self.ops, self.k, self.ld = ops, int(k), _lib.ld_for_k(int(k))
        self.nU, self.nI = int(nU), int(nI)
        dev = ops.device
        f32 = dict(dtype=torch.float32, device=dev)
        tables = tables or {}
        for names, rows in ((("Gamma_shp", "Gamma_rte", "Theta", "eT"), self.nU),
                            (("Lambda_shp", "Lambda_rte", "Beta", "eB"), self.nI)):
            for n in names:
                t = tables.get(n)
                if t is None:
                    t = torch.zeros((rows, self.ld), **f32)
                assert tuple(t.shape) == (rows, self.ld) and t.dtype == torch.float32 and t.is_contiguous(), n
                setattr(self, n, t)
        self.k_rte = tables["k_rte"] if "k_rte" in tables else torch.zeros(self.nU, **f32)
        self.t_rte = tables["t_rte"] if "t_rte" in tables else torch.zeros(self.nI, **f32)
        assert tuple(self.k_rte.shape) == (self.nU,) and tuple(self.t_rte.shape) == (self.nI,)
        self._cs_part = torch.zeros((ops.refresh_grid(max(self.nU, self.nI)), self.ld), **f32)
        # full-height accumulator tables: the batch sweeps write a row's phi-sums straight to acc[row]
        self.acc_u = torch.zeros((self.nU, self.ld), **f32)
        self.acc_i = torch.zeros((self.nI, self.ld), **f32)
        self.flag_u = torch.zeros(self.nU, dtype=torch.uint8, device=dev)   # rows of the current step, per side
        self.flag_i = torch.zeros(self.nI, dtype=torch.uint8, device=dev)     # (partial_fit; epochs use the workspace's)
        self._part = None                    # scratch for the split rows' partial sums of a device-built batch
        self._cs_fused = None
        # Lazy epochs (fit_hpf_svi): the rate of a BATCH side is rank-1 -- Gamma_rte = k_shp/k_rte + colsum(Beta) for every
        # row, recomputed every batch (PXI:300 / 352) -- and the means are read only through their column sums until a
        # check or the end of the fit.  A side in "factored" form keeps rs_prev (the scalar each row's rate was formed
        # with) + the column sums used, not the [rows, ld] rate table; `stale` means tables are recomputed on demand
        # (materialize).  Same float32 operations as the stored form: bit-identical results.
        self.factored = {"u": None, "i": None}       # None: rate table current; else (rs_prev, cs_used, top)
        self.means_stale = {"u": False, "i": False}
        # the E table (eT / eB) of a side is current for ALL of its rows: a lazy epoch keeps the OTHER side's so -- the
        # whole-table pass of a step writes the new E rows of the rows it changed -- and skips that side's expectation pass
        self.e_valid = {"u": False, "i": False}
        self._rs_prev = {"u": torch.zeros(self.nU, **f32), "i": torch.zeros(self.nI, **f32)}
        self.csT = torch.zeros(self.ld, **f32)      # Theta.sum(axis=0) / Beta.sum(axis=0): set by put(), kept
        self.csB = torch.zeros(self.ld, **f32)      # current by every step
        if "Theta" in tables:
            self.csT = self.colsum("Theta")
        if "Beta" in tables:
            self.csB = self.colsum("Beta")

    def v(self, name):
        return getattr(self, name)[:, : self.k]

    def coo_scratch(self):
        """Per-side scratch of CooBatch (row starts / counts of a grouping) + the scan tiles, allocated on first use."""
        if getattr(self, "_coo", None) is None:
            dev = self.ops.device
            self._coo = {w: (torch.empty(n, dtype=torch.int64, device=dev), torch.empty(n, dtype=torch.int32, device=dev))
                         for w, n in (("u", self.nU), ("i", self.nI))}
            self._coo["tiles"] = torch.zeros(self.ops.svi_prep_scratch_words(), dtype=torch.int64, device=dev)
        return self._coo

    def put(self, name, host):
        """Upload one state array ([n,k] table or [n,1] scalar-rate vector) from the host."""
        self.materialize()      # (a table a lazy step left factored / stale is brought up to date before any is replaced)
        dev = self.ops.device
        a = torch.from_numpy(np.ascontiguousarray(host, dtype=np.float32))
        self.e_valid = {"u": False, "i": False}       # (whatever is uploaded, the E tables no longer describe it)
        if name in ("k_rte", "t_rte"):
            getattr(self, name).copy_(a.reshape(-1))
        else:
            self.v(name).copy_(a.to(dev))
            if name == "Theta":           # Theta.sum(axis=0) / Beta.sum(axis=0) are kept current by every step
                self.csT = self.colsum("Theta")
            elif name == "Beta":
                self.csB = self.colsum("Beta")

    def get(self, name, out=None):
        """Download one state array; into `out` (in place) when given."""
        self.materialize()
        t = getattr(self, name).reshape(-1, 1) if name in ("k_rte", "t_rte") else self.v(name)
        if out is not None and isinstance(out, np.ndarray) and out.flags.c_contiguous and out.flags.writeable \
                and out.dtype == np.float32 and tuple(out.shape) == tuple(t.shape):
            torch.from_numpy(out).copy_(t)       # one device-to-host copy, no second pass on the host
            return out
        a = t.contiguous().cpu().numpy()
        if out is None:
            return a
        out[...] = a
        return out

    def load(self, Gamma_shp, Gamma_rte, Lambda_shp, Lambda_rte, k_rte, t_rte, Theta, Beta):
        for n, a in zip(_NAMES + ("k_rte", "t_rte"),
                        (Gamma_shp, Gamma_rte, Lambda_shp, Lambda_rte, Theta, Beta, k_rte, t_rte)):
            self.put(n, a)

    def init_state(self, raw, hy):
        """initialize_parameters (PXI:127-141) from the MT19937 stream's first 2*(nU + nI)*k words (cavi.draw_init_words):
        the four tables in the reference's draw order, bit-identical to its numpy draws; see cavi.init_state."""
        ops, k, ld, nU, nI = self.ops, self.k, self.ld, self.nU, self.nI
        assert raw.numel() == 2 * (nU + nI) * k
        for n in _NAMES:
            getattr(self, n).zero_()
        ops.uniform_rows(raw[: nU * k], self.Gamma_rte, nU, k, ld, hy.a_prime, 0.01)
        ops.uniform_rows(raw[nU * k: (nU + nI) * k], self.Lambda_rte, nI, k, ld, hy.c_prime, 0.01)
        ops.uniform_rows(raw[(nU + nI) * k: (2 * nU + nI) * k], self.Gamma_shp, nU, k, ld, hy.a_prime, 0.01,
                         den=self.Gamma_rte, ratio=self.Theta)
        ops.uniform_rows(raw[(2 * nU + nI) * k:], self.Lambda_shp, nI, k, ld, hy.c_prime, 0.01, den=self.Lambda_rte,
                         ratio=self.Beta)
        self.k_rte.fill_(float(hy.b_prime))
        self.t_rte.fill_(float(hy.d_prime))
        self.e_valid = {"u": False, "i": False}
        self.csT = self.colsum("Theta")
        self.csB = self.colsum("Beta")

    def store(self, Gamma_shp, Gamma_rte, Lambda_shp, Lambda_rte, k_rte, t_rte, Theta, Beta):
        for n, a in zip(_NAMES + ("k_rte", "t_rte"),
                        (Gamma_shp, Gamma_rte, Lambda_shp, Lambda_rte, Theta, Beta, k_rte, t_rte)):
            self.get(n, out=a)

    def colsum(self, name):
        """tab.sum(axis=0) -> [ld] (HIP colsum kernels; PXI:300,320,352,372).  HPF_COLSUM_ORDER=reference: in numpy's own
        order (float32, row after row), bit for bit the reference's sum."""
        tab = getattr(self, name)
        if _reference_sums():
            out = torch.zeros(self.ld, dtype=torch.float32, device=self.ops.device)
            self.ops.colsum_sequential(tab, tab.shape[0], self.ld, out)
            return out
        part = torch.zeros((self.ops.finalize_grid(tab.shape[0]), self.ld), dtype=torch.float32, device=self.ops.device)
        self.ops.colsum(tab, tab.shape[0], self.ld, part)
        out = torch.zeros(self.ld, dtype=torch.float32, device=self.ops.device)
        self.ops.colsum_reduce(part, out, self.ld)
        return out

    # ------------------------------------------------------------------------------------------
    def _side(self, which):
        if which == "u":
            return dict(n=self.nU, shp=self.Gamma_shp, rte=self.Gamma_rte, fac=self.Theta, rs=self.k_rte)
        return dict(n=self.nI, shp=self.Lambda_shp, rte=self.Lambda_rte, fac=self.Beta, rs=self.t_rte)

    def materialize(self, which=("u", "i"), means=True):
        """Bring the rate table (a factored side) and, with `means`, the mean table of the listed sides up to date: one
        whole-side pass each, the very statements the stored form executes every batch (rte = top/rs + cs; fac = shp/rte)."""
        ops, k, ld = self.ops, self.k, self.ld
        for w in which:
            S, fr = self._side(w), self.factored[w]
            if fr is not None:
                rs_prev, cs_used, top = fr
                ops.svi_side(S["n"], None, None, None, S["shp"], S["rte"], S["fac"] if means else None, S["rs"], cs_used,
                             self._cs_part, 0.0, 1.0, 0.0, top, 0.0, 1.0, 0.0, 0, 0, k, ld, rs_rate=rs_prev)
                self.factored[w] = None
                if means:
                    self.means_stale[w] = False
            elif means and self.means_stale[w]:
                ops.svi_side(S["n"], None, None, None, S["shp"], S["rte"], S["fac"], S["rs"], self.csT, self._cs_part,
                             0.0, 1.0, 0.0, 1.0, 0.0, 1.0, 0.0, 1, 0, k, ld)
                self.means_stale[w] = False

    def _part_scratch(self, rows):
        if self._part is None or self._part.shape[0] < rows:
            self._part = torch.empty((rows, self.ld), dtype=torch.float32, device=self.ops.device)
        return self._part

    def batch_phi_sums(self, su, si, flag_u, flag_i, e_current=(), sides=("u", "i")):
        """Per touched row, sum_n w_n * (other side's E row) over the batch's nonzeros (update_phi[_csr] +
        update_G_n_L_sh[_csr] restricted to the batch; sum phi = E_row (*) this), from the CURRENT shapes/rates,
        left in acc_u[user] / acc_i[item] for the rows present in the batch.  su / si: the batch grouped by user / by
        item -- BatchSides (partial_fit: sizes known on the host) or DevSides (epochs: sizes on the device); flag_u /
        flag_i: one byte per row, the rows of the step.  sides: which groupings are swept here (the E rows of both are
        brought up to date either way: each sweep gathers the other side's)."""
        ops, k, ld = self.ops, self.k, self.ld
        if "u" not in e_current:         # (e_current: sides whose E table is up to date for every row already)
            ops.expect(self.Gamma_shp, self.Gamma_rte, self.eT, self.nU, k, ld, flag=flag_u, factored=self.factored["u"])
        if "i" not in e_current:
            ops.expect(self.Lambda_shp, self.Lambda_rte, self.eB, self.nI, k, ld, flag=flag_i, factored=self.factored["i"])
        for w, side, e_self, e_other, acc in (("u", su, self.eT, self.eB, self.acc_u), ("i", si, self.eB, self.eT, self.acc_i)):
            if side.nseg == 0 or w not in sides:
                continue
            # a row that is one segment long writes its sums straight to acc[row]; split rows go through part[]
            if isinstance(side, DevSide):
                part = self._part_scratch(side.nseg)
                ops.sweep(side, e_self, e_other, part, k, ld, acc_rows=acc, acc_ld=ld)
                ops.segsum_desc(part, side.multi, side.nmulti_dev, side.multi_cap, acc, ld)
                continue
            part = torch.empty((side.nseg, ld), dtype=torch.float32, device=ops.device)
            ops.sweep(side, e_self, e_other, part, k, ld, acc_rows=acc, acc_ld=ld)
            if side.nmulti > 0:
                tmp = torch.zeros((side.nmulti, ld), dtype=torch.float32, device=ops.device)
                ops.segsum(part, side.row_seg_ptr, side.nmulti, tmp, ld, row_list=side.multi_local)
                acc.index_copy_(0, side.rows[side.multi_local], tmp)

    def fused_cs_part(self, blocks, tail):
        """[blocks + tail, ld] partial column sums: the fused sweep's blocks, then the whole-table pass's."""
        if self._cs_fused is None or self._cs_fused.shape[0] != blocks + tail:
            self._cs_fused = torch.zeros((blocks + tail, self.ld), dtype=torch.float32, device=self.ops.device)
        return self._cs_fused


def _svi_step(m, hy, su, si, flag_u, flag_i, step, mult, user_batch, all_scalar_rows, lazy=False):
    """One stochastic update in the reference's statement order (user batch: PXI:292-325 / 438-473;
    item batch: PXI:344-377).  su / si: the batch grouped by user / by item; flag_u / flag_i (uint8 per row): the rows
    the updates run over -- supersets of the rows present in the batch; a listed row without a nonzero must hold a zero
    phi-sum in acc_u / acc_i (the callers see to that).  `hy` carries a, c, k_shp, t_shp, add_k_rte, add_t_rte as
    python floats.  lazy (the epochs of fit_hpf_svi): the batch side's rate stays factored and no mean table is
    stored -- DeviceModel.materialize() brings the tables up to date when somebody reads them."""
    ops, k, ld = m.ops, m.k, m.ld
    ref_sums = _reference_sums()
    lazy = lazy and not ref_sums          # (the reference-order sums walk the stored mean tables)
    step_prev = float(np.float32(1) - np.float32(step))
    step = float(np.float32(step))
    w_other = float(np.float32(step * float(np.float32(mult))))   # step*multiplier as one float32 scalar (PXI:316)
    bw, ow = ("u", "i") if user_batch else ("i", "u")
    e_fold = lazy and os.environ.get("HPF_SVI_E_FOLD", "1") == "1"
    if not lazy:
        m.materialize(means=False)            # (stored form: both rate tables are read and written in place)
    elif m.factored[ow] is not None:
        # the other side's rate is BLENDED row by row (PXI:320 / 372): it needs the table.  At an epoch boundary the side's E
        # table is refreshed for all rows as well: ONE pass reads the shapes, forms the rank-1 rate, stores it and the E row
        if e_fold and not m.e_valid[ow]:
            So = m._side(ow)
            ops.expect(So["shp"], None, m.eB if ow == "i" else m.eT, So["n"], k, ld, factored=m.factored[ow],
                       rte_out=So["rte"])
            m.factored[ow] = None
            m.e_valid[ow] = True
        else:
            m.materialize((ow,), means=False)
    e_current = ()
    if e_fold:
        # The other side's E rows change only where a step changes its shapes and rates -- the step's own rows, whose new E
        # rows the whole-table pass below writes while it has them in registers.  So its E table only has to be made current
        # once per epoch (the sides swap roles), not re-read for the touched rows (most of the side) every batch.
        So = m._side(ow)
        if not m.e_valid[ow]:
            ops.expect(So["shp"], So["rte"], m.eB if ow == "i" else m.eT, So["n"], k, ld)
            m.e_valid[ow] = True
        e_current = (ow,)
    m.e_valid[bw] = False                                          # (its rates change for every row below)
    if not lazy:
        m.e_valid[ow] = False
    U = dict(n=m.nU, flag=flag_u, shp=m.Gamma_shp, rte=m.Gamma_rte, fac=m.Theta, rs=m.k_rte, e=m.eT, acc=m.acc_u,
             prior=hy["a"], top=hy["k_shp"], add=hy["add_k_rte"], cs="csT")
    I = dict(n=m.nI, flag=flag_i, shp=m.Lambda_shp, rte=m.Lambda_rte, fac=m.Beta, rs=m.t_rte, e=m.eB, acc=m.acc_i,
             prior=hy["c"], top=hy["t_shp"], add=hy["add_t_rte"], cs="csB")
    B, O = (U, I) if user_batch else (I, U)     # batch side, other side
    s_oth = si if user_batch else su             # the batch grouped by the other side's rows
    # The other side's pass FUSED into its sweep (hpf_hip_sweep_svi_f32): the rows a batch touches there are short, the
    # wavefront that forms a row's phi-sum finishes the row.  Device-built batches only (their flags tell split rows apart;
    # partial_fit's blend of ALL scalar rates, PXI:472-473, is the whole-table pass's rs_mode 2 over the rows the sweep
    # did not finish); needs the batch side's NEW column sums, so that
    # sweep runs after the batch side's pass -- phi still comes from the OLD parameters: the batch side's pass writes
    # neither E table, and the other side's rows are rewritten by the very wavefront that has just used them.
    fused = (isinstance(s_oth, DevSide) and s_oth.nseg > 0 and os.environ.get("HPF_SVI_FUSED", "1") == "1")
    # The batch side's step FUSED into ITS sweep as well (hpf_hip_sweep_svi_batch_f32): the wavefront that sweeps a batch row
    # forms the row's E row first (from its current shape and rate: no expectation launch, no read-back of the E row) and,
    # for a row present in one segment, finishes the row -- shape, rate, mean, scalar rate, column-sum share -- while it holds
    # the phi-sum.  Its rate uses the OTHER side's column sums from before the step (PXI:300 / 352): nothing it needs is
    # produced by the step itself.  The whole-table pass then covers split rows, rows without nonzeros and every row outside
    # the batch (the reference recomputes the batch side's rates and means for ALL rows, PXI:300,318 / 352,370).
    s_bat = su if user_batch else si             # the batch grouped by its own side's rows
    fused_b = (isinstance(s_bat, DevSide) and s_bat.nseg > 0 and os.environ.get("HPF_SVI_FUSED_BATCH", "1") == "1")
    if fused_b:       # (only the other side's E rows are brought up to date here: the batch side's come out of its sweep)
        m.batch_phi_sums(su, si, flag_u, flag_i, tuple(e_current) + (bw,), sides=())
    else:
        m.batch_phi_sums(su, si, flag_u, flag_i, e_current, sides=(bw,) if fused else ("u", "i"))   # phi from the OLD parameters
    # SVI epochs blend the scalar rates of the step's rows only (PXI:324-325, 376-377), partial_fit of all (PXI:472-473)
    rs_mode = 2 if all_scalar_rows else 1
    # One pass per side (hpf_hip_svi_side_f32).  Batch side: shapes of its rows reset to prior + phi, the rate of
    # EVERY row from the other side's current column sums, means, scalar rates, column sums ...
    cs_for_batch = getattr(m, O["cs"])
    cs_part_b, done_b = m._cs_part, 0
    if fused_b:
        blocks, tail = ops.sweep_blocks, m._cs_part.shape[0]
        cs_fused = m.fused_cs_part(blocks, tail)
        part = m._part_scratch(s_bat.nseg)
        ops.sweep_svi_batch(s_bat, B["e"], O["e"], part, B["shp"], B["rte"], None if lazy else B["rte"],
                            None if lazy else B["fac"], B["rs"], m._rs_prev[bw] if lazy else None, m.factored[bw],
                            cs_for_batch, cs_fused[:blocks], B["prior"], 1.0, 0.0, B["top"], B["add"], step, step_prev, k, ld)
        ops.segsum_desc(part, s_bat.multi, s_bat.nmulti_dev, s_bat.multi_cap, B["acc"], ld)
        if not fused:         # the other side's phi-sums, from the E rows the sweep above has just written (OLD parameters)
            m.batch_phi_sums(su, si, flag_u, flag_i, ("u", "i"), sides=(ow,))
        cs_part_b, done_b = cs_fused[blocks:], 1
    if lazy:
        ops.svi_side(B["n"], B["flag"], B["acc"], B["e"], B["shp"], None, None, B["rs"], cs_for_batch,
                     cs_part_b, B["prior"], 1.0, 0.0, B["top"], B["add"], step, step_prev, 0, rs_mode, k, ld,
                     rs_prev_out=m._rs_prev[bw], done_flag=done_b)
        m.factored[bw] = (m._rs_prev[bw], cs_for_batch, B["top"])
        m.means_stale[bw] = m.means_stale[ow] = True
    else:
        ops.svi_side(B["n"], B["flag"], B["acc"], B["e"], B["shp"], B["rte"], B["fac"], B["rs"], cs_for_batch,
                     cs_part_b, B["prior"], 1.0, 0.0, B["top"], B["add"], step, step_prev, 0, rs_mode, k, ld, done_flag=done_b)
    cs_batch = torch.empty(ld, dtype=torch.float32, device=ops.device)      # (the reduction writes every column)
    if ref_sums:
        ops.colsum_sequential(B["fac"], B["n"], ld, cs_batch)
    else:
        ops.colsum_reduce(cs_fused if fused_b else m._cs_part, cs_batch, ld)
    setattr(m, B["cs"], cs_batch)
    # ... other side: shapes and rates of the touched rows blended towards the step's estimate (the rates with the
    # batch side's NEW column sums), means of every row, scalar rates, column sums
    e_out = O["e"] if e_current else None
    if fused:
        blocks, tail = ops.sweep_blocks, m._cs_part.shape[0]
        cs_part = m.fused_cs_part(blocks, tail)
        part = m._part_scratch(s_oth.nseg)
        ops.sweep_svi(s_oth, O["e"], B["e"], part, e_out, O["shp"], O["rte"], None if lazy else O["fac"], O["rs"], cs_batch,
                      cs_part[:blocks], O["prior"], w_other, step_prev, O["top"], O["add"], step, step_prev, k, ld)
        ops.segsum_desc(part, s_oth.multi, s_oth.nmulti_dev, s_oth.multi_cap, O["acc"], ld)
        # split rows and the rows the batch does not touch (their means still count in the column sums)
        ops.svi_side(O["n"], O["flag"], O["acc"], O["e"], O["shp"], O["rte"], None if lazy else O["fac"], O["rs"], cs_batch,
                     cs_part[blocks:], O["prior"], w_other, step_prev, O["top"], O["add"], step, step_prev, 1, rs_mode, k, ld,
                     e_out=e_out, done_flag=1)
    else:
        cs_part = m._cs_part
        ops.svi_side(O["n"], O["flag"], O["acc"], O["e"], O["shp"], O["rte"], None if lazy else O["fac"], O["rs"], cs_batch,
                     cs_part, O["prior"], w_other, step_prev, O["top"], O["add"], step, step_prev, 1, rs_mode, k, ld,
                     e_out=e_out)
    cs_o = torch.empty(ld, dtype=torch.float32, device=ops.device)
    if ref_sums:
        ops.colsum_sequential(O["fac"], O["n"], ld, cs_o)
    else:
        ops.colsum_reduce(cs_part, cs_o, ld)
    setattr(m, O["cs"], cs_o)


def _dev_ids(a, dev):
    return layout.ids_to_device(a, dev)


def _host_ids(a):
    """The reference's size_t ids as an int64 view (no conversion pass; an id >= 2^63 comes out negative and fails the
    device's range check)."""
    a = np.ascontiguousarray(a)
    if a.dtype == np.uint64:
        a = a.view(np.int64)
    elif a.dtype != np.int64:
        a = a.astype(np.int64)
    if not a.flags.writeable:
        a = a.copy()            # (torch.from_numpy wants a writable buffer; never written here)
    return a


class CooBatch:
    """Both groupings of a batch that arrives as COO triplets (partial_fit), built ON THE DEVICE without a read-back:
    range check + narrowing of the ids (hpf_hip_svi_coo_narrow), one stable device sort per grouping, the segment layout of
    each (hpf_hip_svi_coo_prepare: flags 0 / 1 / 2 of every row of the side, segments, split-row descriptors, sizes).  The
    sides come out as DevSides -- what the fused sweeps of an epoch step take -- and the row lists the reference derives
    with np.unique (INIT:864-871) are the rows whose flag is set."""

    def __init__(self, m, bu, bi, by, seg_cap=None):
        ops, dev = m.ops, m.ops.device
        self.seg_cap = seg_cap = layout.SEG_CAP if seg_cap is None else int(seg_cap)
        self.n = n = int(by.shape[0])
        i32 = dict(dtype=torch.int32, device=dev)
        i64 = dict(dtype=torch.int64, device=dev)
        self.err = torch.zeros(1, **i64)
        u32, it32 = torch.empty(n, **i32), torch.empty(n, **i32)
        ops.svi_coo_narrow(bu, m.nU, u32, self.err)
        ops.svi_coo_narrow(bi, m.nI, it32, self.err)
        self.sizes = torch.zeros((2, 8), **i64)
        scratch = m.coo_scratch()
        self.parts = {}
        for w, key, other, nrows, flag in (("u", u32, it32, m.nU, m.flag_u), ("i", it32, u32, m.nI, m.flag_i)):
            ks, perm = torch.sort(key, stable=True)
            idx, y = other[perm], by[perm]
            segs = torch.empty((min(nrows, n) + n // seg_cap + 1, 2), **i64)
            multi = torch.empty((n // seg_cap + 2, 3), **i64)
            sz = self.sizes[0 if w == "u" else 1]
            ops.svi_coo_prepare(ks, nrows, seg_cap, flag, scratch[w][0], scratch[w][1], segs, multi, sz, scratch["tiles"])
            self.parts[w] = (segs, idx, y, multi, sz)

    def read_sizes(self):
        """The ONE synchronisation of a partial_fit call: (range error, {side: (segments, split rows, rows present)})."""
        h = torch.cat([self.err, self.sizes.reshape(-1)]).cpu().numpy()
        self.host = {"u": (int(h[1 + 2]), int(h[1 + 3]), int(h[1 + 5])), "i": (int(h[9 + 2]), int(h[9 + 3]), int(h[9 + 5]))}
        return bool(h[0] != 0), bool(h[1 + 7] != 0 or h[9 + 7] != 0)

    def side(self, w):
        segs, idx, y, multi, sz = self.parts[w]
        nseg = self.host[w][0]
        short = 1 if (nseg > 0 and self.n / nseg < layout.SHORT_ROW_NNZ) else 0
        return DevSide(segs, segs.shape[0], idx, y, sz[2:3], multi, sz[3:4], multi.shape[0], short)


# -- PXI:423-473 ------------------------------------------------------------------------------------
def partial_fit_device(m, Y_batch, ix_u_batch, ix_i_batch, add_k_rte, add_t_rte, a, c, k_shp, t_shp,
                       users_this_batch, items_this_batch, step_size_batch, multiplier_batch, user_batch,
                       nusers_total=None):
    """One partial_fit step on a DeviceModel that already holds the current state: only the batch's triplets and
    row lists cross PCIe.  Mutates the state of `m` (as the reference mutates its eight arrays, PXI:443-473; rate and mean
    tables the step leaves factored / stale are brought up to date when somebody reads them: DeviceModel.materialize).
    users_this_batch / items_this_batch None: the users / items that occur in the batch (what the class computes with
    np.unique, INIT:864-871) -- they fall out of the grouping the step needs anyway, on the device; multiplier_batch None:
    nusers_total / (number of those users), INIT:912.
    The step is the epoch step of fit_hpf_svi: both groupings built on the device (CooBatch), both sides' statements fused
    into their sweeps; only k_rte / t_rte differ -- blended for ALL rows (PXI:472-473)."""
    ops = m.ops
    dev = ops.device
    if multiplier_batch is None and nusers_total is None:
        raise ValueError("partial_fit: either multiplier_batch or nusers_total must be given")
    hy = {"a": float(np.float32(a)), "c": float(np.float32(c)), "k_shp": float(np.float32(k_shp)),
          "t_shp": float(np.float32(t_shp)), "add_k_rte": float(np.float32(add_k_rte)),
          "add_t_rte": float(np.float32(add_t_rte))}
    timing = os.environ.get("HPF_TIMING") == "1" and dev.type == "cuda"      # PF_TIMINGS: device-synchronised phases of a call
    if timing:
        import time
        torch.cuda.synchronize(dev)
        t_ph = [time.perf_counter()]

        def phase(name):
            torch.cuda.synchronize(dev)
            t_ph.append(time.perf_counter())
            PF_TIMINGS.setdefault(name, []).append(t_ph[-1] - t_ph[-2])
    else:
        def phase(name):
            pass
    # (plain pageable copies: 62 MB of C5 triplets go up in 1.1-1.6 ms here; staging them through a page-locked buffer of
    #  our own was no faster and its host-side copy stalled for 90 ms every few calls: profiles/r06_partial_fit_upload.txt)
    bu, bi, by = (torch.from_numpy(x).to(dev) for x in (_host_ids(ix_u_batch), _host_ids(ix_i_batch),
                                                       np.ascontiguousarray(Y_batch, dtype=np.float32)))
    phase("upload")
    batch = CooBatch(m, bu, bi, by)
    phase("groupings")
    lists = {}
    for w, given in (("u", users_this_batch), ("i", items_this_batch)):
        lists[w] = None if given is None else _dev_ids(given, dev)
    bad_id, overflow = batch.read_sizes()
    if bad_id:
        raise ValueError("partial_fit: user/item id out of range")
    if overflow:
        raise _lib.HpfHipError("hpfrec_amd: a partial_fit batch outgrew its workspace")
    su, si = batch.side("u"), batch.side("i")
    count = {}
    for w, flag, acc, nrows in (("u", m.flag_u, m.acc_u, m.nU), ("i", m.flag_i, m.acc_i, m.nI)):
        tb = lists[w]
        if tb is None:
            count[w] = batch.host[w][2]
            continue
        if tb.numel() > 0 and (int(tb.min()) < 0 or int(tb.max()) >= nrows):
            raise ValueError("partial_fit: user/item id out of range")
        listed = torch.zeros(nrows, dtype=torch.bool, device=dev)
        listed[tb] = True
        if bool(((flag != 0) & ~listed).any()):
            raise ValueError("the batch contains users/items that are not in users_in_batch/items_in_batch")
        absent = tb[flag[tb] == 0]           # listed rows without any nonzero in the batch: a zero phi-sum, finished by the
        if absent.numel() > 0:                # whole-table pass (flag 2, as a split row)
            flag[absent] = 2
            acc.index_fill_(0, absent, 0.0)
        count[w] = int(tb.shape[0])
    if multiplier_batch is None:
        multiplier_batch = np.float32(float(nusers_total) / float(max(count["u"], 1)))
    phase("sizes, lists")
    _svi_step(m, hy, su, si, m.flag_u, m.flag_i, step_size_batch, multiplier_batch, user_batch, all_scalar_rows=True,
              lazy=os.environ.get("HPF_SVI_LAZY", "1") == "1")
    phase("step")


def partial_fit_step(ops, Y_batch, ix_u_batch, ix_i_batch, Theta, Beta, Gamma_shp, Gamma_rte, Lambda_shp,
                     Lambda_rte, k_rte, t_rte, add_k_rte, add_t_rte, a, c, k_shp, t_shp, k, users_this_batch,
                     items_this_batch, step_size_batch, multiplier_batch, user_batch):
    """The module-level form (drop-in for the extension function): the caller's host arrays are the state, so all
    eight go up and come back, in place.  hpfrec_amd.HPF keeps the state on the device between calls instead
    (hpfrec_amd.resident)."""
    m = DeviceModel(ops, k, Theta.shape[0], Beta.shape[0])
    m.load(Gamma_shp, Gamma_rte, Lambda_shp, Lambda_rte, k_rte, t_rte, Theta, Beta)
    partial_fit_device(m, Y_batch, ix_u_batch, ix_i_batch, add_k_rte, add_t_rte, a, c, k_shp, t_shp,
                       users_this_batch, items_this_batch, step_size_batch, multiplier_batch, user_batch)
    m.store(Gamma_shp, Gamma_rte, Lambda_shp, Lambda_rte, k_rte, t_rte, Theta, Beta)


# -- PXI:262-377 (+ the shared convergence tail PXI:380-418) ---------------------------------------------
def fit_hpf_svi(hy, Y, ix_u, ix_i, Theta, Beta, Gamma_shp, Gamma_rte, Lambda_shp, Lambda_rte, k_rte, t_rte, maxiter,
                stop_crit, check_every, stop_thr, users_per_batch, items_per_batch, step_size, save_folder,
                random_seed, verbose, has_valset, Yval, ix_u_val, ix_i_val, full_llk, keep_all_objs, make_ops,
                device_triplets=None, init_draw=None, resident=None, tick=None):
    """`init_draw`: the running device draw of the initial state (cython_loops_float.start_init_draw); the eight host
    arrays are then outputs only.  Without it they hold the initial state (initialize_parameters).
    `resident`: see cython_loops_float.fit_hpf."""
    from . import cython_loops_float as be   # printing helpers and save_parameters
    import time
    tick = tick or (lambda phase: None)
    ops = make_ops()
    dev = ops.device
    nU, k = Theta.shape
    nI = Beta.shape[0]
    m = DeviceModel(ops, k, nU, nI)
    tick("state tables allocated (the phase clock also waits for the MT19937 recurrence on its side stream here)")
    if init_draw is None:
        m.load(Gamma_shp, Gamma_rte, Lambda_shp, Lambda_rte, k_rte, t_rte, Theta, Beta)
    hyd = {"a": float(hy.a), "c": float(hy.c), "k_shp": float(hy.k_shp), "t_shp": float(hy.t_shp),
           "add_k_rte": float(hy.add_k_rte), "add_t_rte": float(hy.add_t_rte)}

    if device_triplets is not None:
        tu, ti, ty = (t.to(dev) for t in device_triplets)
    else:
        tu = _dev_ids(ix_u, dev)
        ti = _dev_ids(ix_i, dev)
        ty = torch.from_numpy(np.ascontiguousarray(Y, dtype=np.float32)).to(dev)
    users, items, u_sorted = layout.build_sides(tu, ti, ty, nU, nI)
    tick("triplets to the device, CSR/CSC layout")
    if init_draw is not None:        # (the MT19937 recurrence ran on its own stream under the uploads and sorts above)
        m.init_state(be.finish_init_draw(dev, init_draw), hy)
        tick("initial tables (incl. waiting for the MT19937 recurrence)")

    val = None
    if has_valset and Yval is not None and Yval.shape[0] > 0:
        val = (_dev_ids(ix_u_val, dev).to(torch.int32), _dev_ids(ix_i_val, dev).to(torch.int32),
               torch.from_numpy(np.ascontiguousarray(Yval, dtype=np.float32)).to(dev))

    users_numeration = np.arange(nU, dtype=np.uint64) if users_per_batch != 0 else None
    items_numeration = np.arange(nI, dtype=np.uint64) if items_per_batch > 0 else None
    nbatches_u = int(np.ceil(float(nU) / float(users_per_batch))) if users_per_batch != 0 else 0
    nbatches_i = int(np.ceil(float(nI) / float(items_per_batch))) if items_per_batch > 0 else 0
    rng = np.random.default_rng(seed=random_seed if random_seed > 0 else None)   # PXI:207

    # A batch's index structures (its rows' segment list, the same nonzeros grouped by the other side, the flags of the
    # rows both touch) are built ON THE DEVICE, on a second stream, ahead of the kernels that use them; nothing data-
    # dependent comes back.  The batches of an epoch partition its side's rows, so they are built per EPOCH
    # (EpochWorkspace / hpf_hip_svi_epoch_prepare: one labelled partition of the other side's nonzeros, queued while the
    # PREVIOUS epoch's batches run); when an epoch has more batches than a byte counts, or its slices would not fit, per
    # batch (BatchWorkspace / hpf_hip_svi_batch_prepare: one filter pass per batch, one batch ahead, two workspaces
    # alternating).  The host's share is the shuffle (numpy's, as in the reference) and one 8-byte-per-row upload per epoch.
    # (HPF_SVI_LAZY=0: every batch stores the rate and mean tables the reference rewrites -- 2-3 GB per C5 batch that nothing
    # reads before the next check; kept as a switch so that a test can hold the lazy form against it bit for bit)
    lazy = os.environ.get("HPF_SVI_LAZY", "1") == "1"
    # (one preparation stream per device for the life of the process, like the exchange streams of the sharded fit: a new
    # stream per fit walks through the hardware queues, and some of them collide with the compute stream's)
    # (a LOW-priority preparation stream -- hipStreamCreateWithPriority below the compute stream's -- was measured and
    #  changed nothing: 25.5-26.0 ms per C5 epoch either way, profiles/r06_svi_c5_ab.txt)
    prep_stream = _streams.side_stream(dev, "svi-prepare") if dev.type == "cuda" else None
    if prep_stream is not None:
        prep_stream.wait_stream(torch.cuda.current_stream(dev))      # the CSR / CSC built above
    workspaces = {}
    per_epoch = {}       # epoch type -> whether its batches are prepared per epoch

    def sides_of(user_epoch):
        return (users, items, m.acc_u, users_per_batch) if user_epoch else (items, users, m.acc_i, items_per_batch)

    def workspace(user_epoch, slot, epoch_level=False):
        key = (bool(user_epoch), slot, epoch_level)
        if key not in workspaces:
            own, oth, acc, per = sides_of(user_epoch)
            workspaces[key] = (EpochWorkspace if epoch_level else BatchWorkspace)(ops, own, oth, acc, m.ld, per)
            if prep_stream is not None:      # (its buffers were zero-filled on the compute stream)
                prep_stream.wait_stream(torch.cuda.current_stream(dev))
        return workspaces[key]

    order_host = {}      # page-locked staging for an epoch's order, per epoch type (re-used once its copy has completed)

    def upload_order(numeration, user_epoch):
        """The epoch's shuffled row order -> device int64: one asynchronous copy from page-locked memory, queued on the
        preparation stream (its only consumer), so that the compute stream never waits for PCIe."""
        n = numeration.shape[0]
        if dev.type != "cuda":
            return torch.from_numpy(numeration.astype(np.int64))
        slot = order_host.get(user_epoch)
        if slot is None:
            slot = order_host[user_epoch] = [torch.empty(n, dtype=torch.int64, pin_memory=True), None]
        if slot[1] is not None:
            slot[1].synchronize()
        slot[0].numpy()[:] = numeration.view(np.int64)
        with torch.cuda.stream(prep_stream):
            out = slot[0].to(dev, non_blocking=True)
            slot[1] = torch.cuda.Event()
            slot[1].record(prep_stream)
        return out

    def prepare(ws, arg):
        """Queue a preparation (a batch's: arg = its rows; an epoch's: arg = its order) on the preparation stream, after
        the step that last read `ws`."""
        if prep_stream is None:
            ws.prepare(ops, arg)
            return
        if ws.free is not None:
            prep_stream.wait_event(ws.free)
        with torch.cuda.stream(prep_stream):
            ws.prepare(ops, arg)
            ws.ready = torch.cuda.Event()
            ws.ready.record(prep_stream)

    def plan_epoch(i):
        """Epoch i's shuffle (PXI:277 / 329; called in epoch order: one generator), the upload of its order and, for an
        epoch prepared as a whole, the preparation itself -- queued while the epoch before it runs."""
        if users_per_batch > 0 and items_per_batch > 0:
            user_epoch = ((i + 1) % 2) == 0        # PXI:265-269: epoch 0 is an item epoch
        else:
            user_epoch = users_per_batch > 0
        numeration = users_numeration if user_epoch else items_numeration
        rng.shuffle(numeration)
        own, oth, _, per = sides_of(user_epoch)
        per = int(per)
        if user_epoch not in per_epoch:
            per_epoch[user_epoch] = EpochWorkspace.fits(own, oth, per)
        order = upload_order(numeration, user_epoch)
        ews = None
        if per_epoch[user_epoch]:
            # (epochs alternate sides: one workspace per side; an SVI over one side only alternates two of them)
            try:
                ews = workspace(user_epoch, i % 2, epoch_level=True)
            except torch.cuda.OutOfMemoryError:      # a device smaller than the plan assumed: one batch at a time instead
                workspaces.pop((bool(user_epoch), i % 2, True), None)
                torch.cuda.empty_cache()
                per_epoch[user_epoch], ews = False, None
            else:
                prepare(ews, order)
        return dict(user_epoch=user_epoch, order=order, per=per, n_side=own.nrows, ews=ews,
                    nb=nbatches_u if user_epoch else nbatches_i)

    errs = np.zeros(2, dtype=np.longdouble)
    last_crit = -np.inf
    Theta_prev = m.Theta.clone() if stop_crit == "diff-norm" else None

    def evaluate(final=False):
        m.materialize()                      # (lazy epochs: the mean tables are brought up to date for the check)
        if val is not None:
            t = ops.pair_llk(m.Theta, m.Beta, val[0], val[1], val[2], k, m.ld, full_llk).cpu().numpy()
            if final:
                sub = float(np.dot(m.v("Theta")[val[0].long()].sum(dim=0).cpu().numpy(),
                                   m.v("Beta")[val[1].long()].sum(dim=0).cpu().numpy()))
            else:
                sub = t[2]
            errs[0] = np.longdouble(t[0]) - np.longdouble(sub)
            errs[1] = np.sqrt(np.longdouble(t[1]) / val[0].shape[0])
        else:
            # the training nonzeros in the row-grouped layout: user rows read once per segment, item rows gathered
            t = ops.llk_sweep(users, m.Theta, m.Beta, k, m.ld, full_llk).cpu().numpy()
            sub = np.dot(m.csT[:k].cpu().numpy(), m.csB[:k].cpu().numpy())
            errs[0] = np.longdouble(t[0]) - np.longdouble(sub)
            errs[1] = np.sqrt(np.longdouble(t[1]) / users.nnz)

    if verbose > 0:
        print("Initializing optimization procedure...")
    st_time = time.time()
    timing = os.environ.get("HPF_TIMING") == "1"      # SVI_TIMINGS: device-synchronised wall time of the epoch loop
    if timing and dev.type == "cuda":
        torch.cuda.synchronize(dev)
    t_loop = time.perf_counter()
    host_s = [0.0, 0.0]       # host time spent preparing batches / issuing their kernels
    i = -1
    planned = None
    for i in range(maxiter):
        step = float(np.float32(step_size(i)))
        t_h = time.perf_counter()
        cur = planned if planned is not None else plan_epoch(i)
        # the next epoch's order and (epoch-level) preparation are queued BEFORE this epoch's kernels: they run beside
        # them.  (A stop at this epoch's check leaves one unused shuffle behind: the generator is the fit's own.)
        planned = plan_epoch(i + 1) if i + 1 < maxiter else None
        user_epoch, per, n_side, nb, order = cur["user_epoch"], cur["per"], cur["n_side"], cur["nb"], cur["order"]
        host_s[0] += time.perf_counter() - t_h
        if cur["ews"] is not None:
            ews = cur["ews"]
            t_h = time.perf_counter()
            if ews.ready is not None:
                torch.cuda.current_stream(dev).wait_event(ews.ready)
            for j in range(nb):
                own_side, oth_side, f_own, f_oth = ews.batch(j)
                su, si = (own_side, oth_side) if user_epoch else (oth_side, own_side)
                flag_u, flag_i = (f_own, f_oth) if user_epoch else (f_oth, f_own)
                rows_j = min(n_side, (j + 1) * per) - j * per
                _svi_step(m, hyd, su, si, flag_u, flag_i, step, float(n_side) / float(rows_j), user_epoch,
                          all_scalar_rows=False, lazy=lazy)
            if prep_stream is not None:
                ews.free = torch.cuda.Event()
                ews.free.record(torch.cuda.current_stream(dev))
            host_s[1] += time.perf_counter() - t_h
        else:
            t_h = time.perf_counter()
            chunks = [order[bt * per: min(n_side, (bt + 1) * per)] for bt in range(nb)]
            prepare(workspace(user_epoch, 0), chunks[0])
            host_s[0] += time.perf_counter() - t_h
            for j in range(nb):
                t_h = time.perf_counter()
                ws = workspace(user_epoch, j % 2)
                if ws.ready is not None:
                    torch.cuda.current_stream(dev).wait_event(ws.ready)
                su, si = (ws.side_own, ws.side_oth) if user_epoch else (ws.side_oth, ws.side_own)
                flag_u, flag_i = (ws.flag_own, ws.flag_oth) if user_epoch else (ws.flag_oth, ws.flag_own)
                _svi_step(m, hyd, su, si, flag_u, flag_i, step, float(n_side) / float(chunks[j].shape[0]), user_epoch,
                          all_scalar_rows=False, lazy=lazy)
                if prep_stream is not None:
                    ws.free = torch.cuda.Event()
                    ws.free.record(torch.cuda.current_stream(dev))
                host_s[1] += time.perf_counter() - t_h
                if j + 1 < nb:
                    # the next batch's structures are built while this batch's kernels run.  Its workspace was last used
                    # by batch j-1: the host waits for that step here (not a bubble -- batch j is queued already), which
                    # also keeps it at most a batch ahead of the device, so the two clocks time host WORK, not a full
                    # launch queue
                    nxt = workspace(user_epoch, (j + 1) % 2)
                    if nxt.free is not None:
                        nxt.free.synchronize()
                    t_h = time.perf_counter()
                    prepare(nxt, chunks[j + 1])
                    host_s[0] += time.perf_counter() - t_h

        if check_every > 0 and ((i + 1) % check_every) == 0:
            if stop_crit == "diff-norm":
                m.materialize(("u",))
                d = (m.Theta - Theta_prev).double()
                last_crit = float(torch.sqrt((d * d).sum()).item())
                if verbose:
                    be._print_norm_diff(i + 1, check_every, last_crit)
                if last_crit < stop_thr:
                    break
                Theta_prev.copy_(m.Theta)
            else:
                evaluate()
                if verbose:
                    be._print_llk_iter(i + 1, errs[0], float(errs[1]), has_valset)
                if stop_crit != "maxiter":
                    if (i + 1) == check_every:
                        last_crit = errs[0]
                    else:
                        if (1.0 - errs[0] / last_crit) <= stop_thr:
                            break
                        last_crit = errs[0]

    if prep_stream is not None:
        # an epoch planned ahead of a stop at this epoch's check (its order upload and hpf_hip_svi_epoch_prepare) may still be
        # running on the preparation stream: the compute stream joins it before anything below reads the workspaces' sizes
        # or the caching allocator can hand their buffers (allocated on the compute stream's pool) to somebody else
        torch.cuda.current_stream(dev).wait_stream(prep_stream)
    if any(ws.overflowed() for ws in workspaces.values()):        # (cannot happen: capacities come from the largest rows)
        raise _lib.HpfHipError("hpfrec_amd: a stochastic batch outgrew its workspace")
    tick("epochs and checks")
    if timing:
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        SVI_TIMINGS.clear()
        SVI_TIMINGS.update(epochs=i + 1, seconds=time.perf_counter() - t_loop, host_prepare_s=host_s[0],
                           host_issue_s=host_s[1])
    last_llk = None
    if stop_crit in ("diff-norm", "maxiter") and verbose > 0:
        evaluate(final=True)
        last_llk = errs[0]
    if verbose:
        be._print_final_msg(i + 1, errs[0], float(errs[1]), (time.time() - st_time) / 60.0)

    if resident is not None and keep_all_objs and save_folder == "":
        m.materialize()
        resident.adopt(m)              # the state stays on the device; host copies are made when somebody reads them
        return i, None, last_llk
    m.store(Gamma_shp, Gamma_rte, Lambda_shp, Lambda_rte, k_rte, t_rte, Theta, Beta)
    tick("outputs to the host")
    temp = (Gamma_shp, Gamma_rte, Lambda_shp, Lambda_rte, k_rte, t_rte)
    if save_folder != "":
        be.save_parameters(verbose, save_folder,
                           ["Theta", "Beta", "Gamma_shp", "Gamma_rte", "Lambda_shp", "Lambda_rte", "kappa_rte", "tau_rte"],
                           [Theta, Beta] + list(temp))
    return i, (temp if keep_all_objs else None), last_llk


# -- PXI:476-520 ------------------------------------------------------------------------------------
def calc_user_factors(ops, a, a_prime, b_prime, c, c_prime, d_prime, Y, ix_i, Theta, Beta, Lambda_shp, Lambda_rte,
                      nY, k, maxiter, random_seed, stop_thr, return_all, resident=None):
    """Local CAVI for ONE user with the item parameters fixed.  Fills `Theta` (k,) in place; returns
    (Gamma_shp, Gamma_rte, phi/Y) when return_all else None.

    resident: a DeviceModel holding the current Beta / Lambda_shp / Lambda_rte (then the host arguments of those
    names are not read at all: the user's item rows are gathered on the device and Beta.sum(axis=0) is the
    model's resident column sum)."""
    f = np.float32
    dev = ops.device
    ld = _lib.ld_for_k(k)
    a, a_prime, b_prime = f(a), f(a_prime), f(b_prime)
    k_shp = f(a_prime + f(k) * a)
    add_k_rte = f(a_prime / b_prime)
    # initialisation: numpy default_rng stream, in the reference's draw order (PXI:490-497)
    rng = np.random.default_rng(seed=random_seed if random_seed > 0 else None)
    Theta[:] = rng.gamma(a, 1 / b_prime, size=k).astype(np.float32)
    k_rte = f(b_prime + Theta.sum())
    if resident is not None:
        cs = resident.csB
    else:
        Beta_dev = torch.zeros((Beta.shape[0], ld), dtype=torch.float32, device=dev)
        Beta_dev[:, :k] = torch.from_numpy(np.ascontiguousarray(Beta, dtype=np.float32)).to(dev)
        csp = torch.zeros((ops.finalize_grid(Beta.shape[0]), ld), dtype=torch.float32, device=dev)
        cs = torch.zeros(ld, dtype=torch.float32, device=dev)
        ops.colsum(Beta_dev, Beta.shape[0], ld, csp)
        ops.colsum_reduce(csp, cs, ld)
    g1 = rng.gamma(a_prime, b_prime / a_prime, size=1).astype(np.float32)
    unif = rng.uniform(low=.85, high=1.15, size=k).astype(np.float32)
    ix = np.ascontiguousarray(ix_i).astype(np.int64)
    n = int(nY)
    # Gamma_rte = g + Beta.sum(axis=0); Gamma_shp = Gamma_rte * Theta * U(0.85, 1.15) (PXI:493-497), on the device
    init = np.zeros((3, ld), dtype=np.float32)
    init[0, :k], init[1, :k], init[2, 0] = Theta, unif, g1[0]
    init_d = torch.from_numpy(init).to(dev)
    Gr = torch.zeros(ld, dtype=torch.float32, device=dev)
    Gs = torch.zeros(ld, dtype=torch.float32, device=dev)
    Gr[:k] = init_d[2, 0] + cs[:k]
    Gs[:k] = Gr[:k] * init_d[0, :k] * init_d[1, :k]
    torch.nan_to_num_(Gs)
    torch.nan_to_num_(Gr)
    th = init_d[0].clone()
    # E rows of the user's items: in place in the resident model's scratch table, or of the uploaded rows
    y_d = torch.from_numpy(np.ascontiguousarray(Y, dtype=np.float32)).to(dev)
    if resident is not None:
        ixd = torch.from_numpy(ix).to(dev)
        ops.expect(resident.Lambda_shp, resident.Lambda_rte, resident.eB, n, k, ld, row_list=ixd)
        e_items, idx32 = resident.eB, ixd.to(torch.int32)
    else:
        Ls = torch.zeros((n, ld), dtype=torch.float32, device=dev)
        Lr = torch.zeros((n, ld), dtype=torch.float32, device=dev)
        Ls[:, :k] = torch.from_numpy(np.ascontiguousarray(Lambda_shp[ix], dtype=np.float32)).to(dev)
        Lr[:, :k] = torch.from_numpy(np.ascontiguousarray(Lambda_rte[ix], dtype=np.float32)).to(dev)
        e_items = torch.zeros((n, ld), dtype=torch.float32, device=dev)
        ops.expect(Ls, Lr, e_items, n, k, ld)
        idx32 = torch.arange(n, dtype=torch.int32, device=dev)
    e_last = torch.zeros(ld, dtype=torch.float32, device=dev)
    rounds = torch.zeros(1, dtype=torch.int32, device=dev)
    # the whole local coordinate ascent (PXI:505-517) is one launch
    ops.fold_in(idx32, y_d, e_items, cs, Gs, Gr, th, e_last, rounds, a, k_shp, add_k_rte, k_rte, stop_thr, maxiter, k, ld)
    Theta[:] = th[:k].cpu().numpy()
    if not return_all:
        return None
    # phi / Y: the multinomial probabilities of the LAST phi (computed from the Gamma before its final update)
    prob = e_last[None, :] * e_items[idx32.long()]
    prob = (prob / prob.sum(dim=1, keepdim=True))[:, :k]
    return Gs[:k].cpu().numpy(), Gr[:k].cpu().numpy(), prob.contiguous().cpu().numpy()
