"""Drop-in for the reference's compiled extension module `hpfrec.cython_loops_float`.

The reference's `hpfrec/__init__.py` talks to its Cython extension only through module-level
functions and three type attributes (cython_loops.pxi, "PXI": fit_hpf PXI:147, partial_fit
PXI:423, calc_user_factors PXI:476, calc_llk PXI:525, predict_arr PXI:538,
initialize_parameters PXI:117, cast_* PXI:11-18; c_real_t / obj_ind_type from
cython_float.pxi:9 and cython_float_nonwindows.pyx:10).  This module exports the same names with
the same arity, argument meaning, in-place semantics and return values, but runs the numerics on
an MI355X through libhpf_hip.so.  A maintainer of the reference can therefore swap
`from . import cython_loops_float` for `from hpfrec_amd import cython_loops_float`
(INTEGRATION.md shows the stub).

Host side here: argument checking, RNG initialisation (numpy, bit-identical to the reference),
upload/download, the outer loop and stopping rule.  No numerics run on the CPU.
"""
import ctypes
import os
import time

import numpy as np
import torch

from . import _streams, cavi, layout, svi
from .ops_hip import HipOps

c_real_t = ctypes.c_float          # hpfrec/cython_float.pxi:9
obj_ind_type = ctypes.c_size_t     # hpfrec/cython_float_nonwindows.pyx:10
obj_long_double_type = ctypes.c_longdouble

_DEVICE = None   # None = torch's current device; a multi-GPU launcher sets the device per rank beforehand


_OPS = {}


def _make_ops():
    """The HIP op set (one instance per device, created on first use).  There is no alternative implementation in
    this package: HipOps raises when the extension or the GPU is missing.  (tests/ substitute this module's
    `HipOps` name with a numpy stand-in to exercise the host logic on GPU-less machines.)"""
    dev = _DEVICE
    if dev is None and torch.cuda.is_available():
        dev = "cuda:%d" % torch.cuda.current_device()
    key = (HipOps, dev)
    ops = _OPS.get(key)
    if ops is None:
        _OPS.clear()
        ops = _OPS[key] = HipOps(dev)
    return ops


# -- helper functions (PXI:11-18) ---------------------------------------------------------
def cast_real_t(n):
    return float(np.float32(n))


def cast_int(n):
    return int(ctypes.c_int(int(n)).value)


def cast_ind_type(n):
    return int(ctypes.c_size_t(int(n)).value)


# -- PXI:117-143 --------------------------------------------------------------------------
def initialize_parameters(Theta, Beta, random_seed, a, a_prime, b_prime, c, c_prime, d_prime):
    """Random initialisation, on the host with numpy so that the MT19937 stream -- four
    `random(dtype=float32)` draws in the order user-rate, item-rate, user-shape, item-shape -- is
    the reference's.  All four use a'/c' (not a/c), as the reference does.  Fills Theta/Beta in
    place and returns (Gamma_shp, Gamma_rte, Lambda_shp, Lambda_rte, k_rte, t_rte)."""
    nU, k = Theta.shape
    nI = Beta.shape[0]
    rng = np.random.Generator(np.random.MT19937(seed=random_seed if random_seed > 0 else None))
    draws = [rng.random(size=(n, k), dtype=np.float32) for n in (nU, nI, nU, nI)]
    Gamma_rte = a_prime + 0.01 * draws[0]
    Lambda_rte = c_prime + 0.01 * draws[1]
    Gamma_shp = a_prime + 0.01 * draws[2]
    Lambda_shp = c_prime + 0.01 * draws[3]
    k_rte = np.full((nU, 1), b_prime, dtype=np.float32)
    t_rte = np.full((nI, 1), d_prime, dtype=np.float32)
    np.divide(Gamma_shp, Gamma_rte, out=Theta)
    np.divide(Lambda_shp, Lambda_rte, out=Beta)
    return Gamma_shp, Gamma_rte, Lambda_shp, Lambda_rte, k_rte, t_rte


# -- PXI:830-847 --------------------------------------------------------------------------
def print_norm_diff(it, check_every, normdiff):
    print("Iteration %d | Norm(Theta_{%d} - Theta_{%d}): %.5f" % (it, it, it - check_every, normdiff))


def print_llk_iter(it, llk, rmse, has_valset):
    tag = "val" if has_valset else "train"
    print(("Iteration %d | " + tag + " llk: %d | " + tag + " rmse: %.4f") % (it, int(llk), rmse))


def print_final_msg(it, llk, rmse, end_tm):
    print("\n\nOptimization finished")
    print("Final log-likelihood: %d" % int(llk))
    print("Final RMSE: %.4f" % rmse)
    print("Minutes taken (optimization part): %.1f" % end_tm)
    print("")


_print_norm_diff, _print_llk_iter, _print_final_msg = print_norm_diff, print_llk_iter, print_final_msg


# -- PXI:22-42: helpers the reference exports "for ctpfrec" ---------------------------------
def get_csc_data(ix_u, ix_i, Y, nU, nI):
    """COO triplets -> CSC (indptr [nI+1], row indices, values), rows ascending inside a column and duplicate
    (user, item) pairs summed -- what scipy's coo_array(...).tocsc() gives the reference (PXI:22-25) -- built on the
    device: one stable sort of the column-major keys, a segmented sum over runs of equal keys, a bincount."""
    ops = _make_ops()
    dev = ops.device
    nU, nI = int(nU), int(nI)
    u = _as_index_tensor(ix_u, nU, "UserId", dev)
    i = _as_index_tensor(ix_i, nI, "ItemId", dev)
    y = torch.from_numpy(np.ascontiguousarray(Y, dtype=np.float32)).to(dev)
    key = i * nU + u
    skey, order = torch.sort(key, stable=True)
    ukey, counts = torch.unique_consecutive(skey, return_counts=True)
    # sums over runs of equal keys as differences of a float64 running sum: fixed order (no atomics), exact for the
    # pairs and triples of small counts that occur
    run = torch.cumsum(y[order].double(), 0)
    ends = torch.cumsum(counts, 0) - 1
    data = run[ends]
    data[1:] -= run[ends[:-1]]
    data = data.float()
    cols = torch.div(ukey, nU, rounding_mode="floor")
    indptr = torch.zeros(nI + 1, dtype=torch.int64, device=dev)
    torch.cumsum(torch.bincount(cols, minlength=nI), 0, out=indptr[1:])
    return (indptr.cpu().numpy().astype(obj_ind_type), (ukey - cols * nU).cpu().numpy().astype(obj_ind_type),
            data.cpu().numpy().astype(c_real_t))


def get_unique_items_batch(users_this_batch, st_ix_u, ix_i, nthreads, return_ix=False):
    """Sorted unique other-side ids of the listed CSR rows (and, with return_ix, the start of every listed row's
    nonzeros inside the batch: [0, n_0, n_0+n_1, ...]) -- PXI:27-42 -- gathered on the device."""
    ops = _make_ops()
    dev = ops.device
    rows = torch.from_numpy(np.ascontiguousarray(users_this_batch).astype(np.int64)).to(dev)
    st = torch.from_numpy(np.ascontiguousarray(st_ix_u).astype(np.int64)).to(dev)
    idx = torch.from_numpy(np.ascontiguousarray(ix_i).astype(np.int64)).to(dev)
    beg = st[rows]
    deg = st[rows + 1] - beg
    total = int(deg.sum().item())
    offs = torch.cumsum(deg, 0) - deg
    pos = torch.repeat_interleave(beg - offs, deg, output_size=total) + torch.arange(total, device=dev)
    items = torch.unique(idx[pos]).cpu().numpy().astype(np.asarray(ix_i).dtype)
    if not return_ix:
        return items
    st_pos = np.zeros(rows.shape[0] + 1, dtype=np.asarray(users_this_batch).dtype)
    st_pos[1:] = torch.cumsum(deg, 0).cpu().numpy()
    return items, st_pos


def _llk_terms_host(Theta, Beta, Y, ix_u, ix_i, full_llk):
    """[sum y*log(yhat) (- lgamma(y+1)), sum (y-yhat)^2, sum yhat] over listed pairs of host tables (float64)."""
    ops = _make_ops()
    T, B, iu, ii, k, ld = _pair_operands(Theta, Beta, ix_u, ix_i, ops.device)
    y = torch.from_numpy(np.ascontiguousarray(Y, dtype=np.float32)).to(ops.device)
    return ops.pair_llk(T, B, iu, ii, y, k, ld, bool(full_llk)).cpu().numpy()


def _colsum_dot_host(Theta, Beta):
    """Theta.sum(axis=0).dot(Beta.sum(axis=0)) (PXI:78), the column sums taken on the device."""
    ops = _make_ops()
    k = int(Theta.shape[1])
    ld = cavi._lib.ld_for_k(k)
    sums = []
    for M in (Theta, Beta):
        tab = _padded(M, ld, ops.device)
        part = torch.zeros((ops.finalize_grid(tab.shape[0]), ld), dtype=torch.float32, device=ops.device)
        out = torch.zeros(ld, dtype=torch.float32, device=ops.device)
        ops.colsum(tab, tab.shape[0], ld, part)
        ops.colsum_reduce(part, out, ld)
        sums.append(out[:k].cpu().numpy())
    return np.dot(sums[0], sums[1])


def _eval_llk_rmse(errs, Theta, Beta, Y, ix_u, ix_i, n, full_llk, verbose, subtract):
    """llk_plus_rmse (PXI:627-658) + the subtrahend + the square root, into errs[0:2]."""
    t = _llk_terms_host(Theta, Beta, Y, ix_u, ix_i, full_llk)
    if subtract == "yhat":                                  # sum_prediction over the listed pairs (PXI:72)
        sub = t[2]
    elif subtract == "colsums":                             # all pairs (PXI:78)
        sub = _colsum_dot_host(Theta, Beta)
    else:                                                   # PXI:105 (sic): sums over the listed rows only
        iu = np.ascontiguousarray(ix_u).astype(np.int64)
        ii = np.ascontiguousarray(ix_i).astype(np.int64)
        sub = _colsum_dot_host(Theta[iu], Beta[ii])
    errs[0] = np.longdouble(t[0]) - np.longdouble(sub)
    # the reference only accumulates the squared error when verbose (add_mse = verbose, PXI:71,77)
    errs[1] = np.sqrt(np.longdouble(t[1] if verbose else 0.0) / np.longdouble(max(int(n), 1)))


# -- PXI:51-92 ----------------------------------------------------------------------------
def assess_convergence(i, check_every, stop_crit, last_crit, stop_thr, Theta, Theta_prev, Beta, nY, Y, ix_u, ix_i, nYv,
                       Yval, ix_u_val, ix_i_val, errs, k, nthreads, verbose, full_llk, has_valset):
    """-> (has_converged, last_crit); fills errs[0] (llk) / errs[1] (rmse), overwrites Theta_prev in 'diff-norm'
    mode, prints when verbose -- the reference's statements on host arrays, the sums on the device."""
    if stop_crit == "diff-norm":
        ops = _make_ops()
        d = (torch.from_numpy(np.ascontiguousarray(Theta, dtype=np.float32)).to(ops.device).double()
             - torch.from_numpy(np.ascontiguousarray(Theta_prev, dtype=np.float32)).to(ops.device).double())
        last_crit = float(np.float32(torch.sqrt((d * d).sum()).item()))
        if verbose:
            print_norm_diff(i + 1, check_every, last_crit)
        if last_crit < stop_thr:
            return True, last_crit
        Theta_prev[:, :] = Theta
    else:
        if has_valset:
            _eval_llk_rmse(errs, Theta, Beta, Yval, ix_u_val, ix_i_val, nYv, full_llk, verbose, "yhat")
        else:
            _eval_llk_rmse(errs, Theta, Beta, Y, ix_u, ix_i, nY, full_llk, verbose, "colsums")
        if verbose:
            print_llk_iter(i + 1, errs[0], float(errs[1]), has_valset)
        if stop_crit != "maxiter":
            if (i + 1) == check_every:
                last_crit = errs[0]
            else:
                if (1.0 - errs[0] / last_crit) <= stop_thr:
                    return True, last_crit
                last_crit = errs[0]
    return False, last_crit


# -- PXI:94-113 ---------------------------------------------------------------------------
def eval_after_term(stop_crit, verbose, nthreads, full_llk, k, nY, nYv, has_valset, Theta, Beta, errs, Y, ix_u, ix_i,
                    Yval, ix_u_val, ix_i_val):
    """Final llk/rmse for the criteria that do not track llk ('maxiter', 'diff-norm') when verbose; else None."""
    if stop_crit in ("diff-norm", "maxiter") and verbose > 0:
        if has_valset:
            _eval_llk_rmse(errs, Theta, Beta, Yval, ix_u_val, ix_i_val, nYv, full_llk, verbose, "listed-rows")
        else:
            _eval_llk_rmse(errs, Theta, Beta, Y, ix_u, ix_i, nY, full_llk, verbose, "colsums")
        return errs[0]
    return None


def save_parameters(verbose, save_folder, file_names, obj_list):
    """PXI:44-49"""
    if verbose:
        print("Saving final parameters to .csv files...")
    for name, obj in zip(file_names, obj_list):
        np.savetxt(os.path.join(save_folder, name), obj, fmt="%.10f", delimiter=",")


def _as_index_tensor(a, n_max, what, dev):
    """Ids on the device (int64), range-checked there: no pass over the host array besides the copy itself."""
    t = layout.ids_to_device(a, dev)
    if t.numel():
        lo, hi = torch.aminmax(t)
        if int(lo) < 0 or int(hi) >= n_max:
            raise ValueError("%s contains an id outside [0, %d)" % (what, n_max))
    return t


def start_init_draw(ops, dist, random_seed, nU, nI, k):
    """Starts the reference's random initialisation (PXI:127-141) on the device: the MT19937 recurrence for this seed
    (seed <= 0: OS entropy -- every rank of a sharded fit then takes rank 0's state) on a side stream, so that it runs
    under the caller's uploads and CSR/CSC build.  -> (words, done event or None) for finish_init_draw."""
    dev = ops.device
    mt_state = cavi.mt19937_state_words(random_seed).to(dev)
    if dist:
        dist.broadcast(mt_state, 0)
    if dev.type != "cuda":
        return cavi.draw_init_words(ops, mt_state, nU, nI, k), None
    side = _streams.side_stream(dev, "initial-draw")
    side.wait_stream(torch.cuda.current_stream(dev))
    mt_state.record_stream(side)          # (the kernel leaves the stream's new position in it when it ends)
    with torch.cuda.stream(side):
        raw = cavi.draw_init_words(ops, mt_state, nU, nI, k)
        done = torch.cuda.Event()
        done.record(side)
    return raw, done


def finish_init_draw(dev, draw):
    """The drawn words, usable on the current stream."""
    raw, done = draw
    if done is not None:
        cur = torch.cuda.current_stream(dev)
        cur.wait_event(done)
        raw.record_stream(cur)
    return raw


class _Engine:
    """Glue between the reference-shaped host arrays and cavi.FullBatchCavi (handles sharding)."""

    def __init__(self, hyper, Y, ix_u, ix_i, nU, nI, Yval=None, ix_u_val=None, ix_i_val=None, device_triplets=None,
                 random_seed=None, draw_init=False):
        self.ops = _make_ops()
        self.device = self.ops.device
        dist = cavi._dist()
        self.dist = dist
        self.rank = dist.get_rank() if dist else 0
        self.world = dist.get_world_size() if dist else 1
        dev = self.device
        if draw_init:
            self._init_draw = start_init_draw(self.ops, dist, random_seed, nU, nI, hyper.k)
        if device_triplets is not None:      # the caller's triplets are on the device already (hpfrec_amd.HPF.fit)
            tu, ti, ty = (t.to(dev) for t in device_triplets)
            if tu.numel() and (int(tu.max()) >= nU or int(ti.max()) >= nI):
                raise ValueError("UserId / ItemId contains an id out of range")
        else:
            tu = _as_index_tensor(ix_u, nU, "UserId", dev)
            ti = _as_index_tensor(ix_i, nI, "ItemId", dev)
            ty = torch.from_numpy(np.ascontiguousarray(Y, dtype=np.float32)).to(dev)
        lu, li, ly, (self.u0, self.u1) = cavi.shard_users(tu, ti, ty, nU, self.rank, self.world)
        self.nU_global, self.nI = int(nU), int(nI)
        self.model = cavi.FullBatchCavi(self.ops, dev, lu, li, ly, self.u1 - self.u0, nI, hyper)
        self.nnz_global = int(Y.shape[0])
        self.val = None
        if Yval is not None and Yval.shape[0] > 0:
            vu = _as_index_tensor(ix_u_val, nU, "val UserId", dev)
            vi = _as_index_tensor(ix_i_val, nI, "val ItemId", dev)
            vy = torch.from_numpy(np.ascontiguousarray(Yval, dtype=np.float32)).to(dev)
            keep = (vu >= self.u0) & (vu < self.u1)
            self.val = ((vu[keep] - self.u0).to(torch.int32), vi[keep].to(torch.int32), vy[keep])
            self.nval_global = int(Yval.shape[0])

    def init_state(self):
        self.model.init_state(finish_init_draw(self.device, self._init_draw), self.u0, self.nU_global)
        self._init_draw = None

    def gather_users(self, name, out=None):
        """Full (all users) host copy of a user-side array (written into `out` when given)."""
        if not self.dist:
            return self.model.fetch(name, out)
        local = self.model.fetch(name)
        full = torch.zeros((self.nU_global, local.shape[1]), dtype=torch.float32, device=self.device)
        full[self.u0: self.u1] = torch.from_numpy(local).to(self.device)
        self.dist.all_reduce(full)
        if out is None:
            return full.cpu().numpy()
        torch.from_numpy(out).copy_(full)
        return out

    def theta_norm_diff(self, prev):
        d = self.model.Theta - prev
        sq = (d.double() * d.double()).sum().reshape(1)
        if self.dist:
            self.dist.all_reduce(sq)
        return float(np.sqrt(sq.item()))


FIT_TIMINGS = {}


def _phase_clock():
    """HPF_TIMING=1: wall time per phase of the last fit_hpf call in FIT_TIMINGS (device synchronised at phase ends)."""
    if os.environ.get("HPF_TIMING") != "1":
        return lambda phase: None
    FIT_TIMINGS.clear()
    last = [time.perf_counter()]

    def tick(phase):
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        now = time.perf_counter()
        FIT_TIMINGS[phase] = FIT_TIMINGS.get(phase, 0.0) + now - last[0]
        last[0] = now
    return tick


# -- PXI:147-418 --------------------------------------------------------------------------
def fit_hpf(a, a_prime, b_prime, c, c_prime, d_prime, Y, ix_u, ix_i, Theta, Beta, maxiter, stop_crit,
            check_every, stop_thr, users_per_batch, items_per_batch, step_size, sum_exp_trick, st_ix_u,
            save_folder, random_seed, verbose, nthreads, par_sh, has_valset, Yval, ix_u_val, ix_i_val,
            full_llk, keep_all_objs, alloc_full_phi, device_triplets=None, resident=None):
    """Same contract as the reference's fit_hpf (PXI:147-162, returns PXI:413-418):
    fills Theta/Beta in place, returns (i, (Gamma_shp, Gamma_rte, Lambda_shp, Lambda_rte, k_rte,
    t_rte) or None, last_llk) with i the 0-based index of the last iteration run.

    `nthreads`, `par_sh` (allow_inconsistent_math) and `alloc_full_phi` are accepted and ignored:
    the device path is always parallel, always reproducible and never materialises phi.
    `sum_exp_trick` is honoured implicitly: E rows are rescaled per row (power of two) in every mode.
    `device_triplets` (not in the reference's signature): (ix_u, ix_i, Y) as device tensors, when the caller has the
    triplets there already -- the host arrays `ix_u`, `ix_i` are then not read (they may be empty) and of `Y` only the
    length is used.
    `resident` (not in the reference's signature): a resident.ResidentState.  A single-process fit with
    keep_all_objs and no save_folder then ENDS ON THE DEVICE: the state tables are handed to it (`adopt`), `Theta` /
    `Beta` are NOT filled and None is returned in place of the six arrays -- hpfrec_amd.HPF reads its attributes
    through that object, which makes host copies when somebody asks for them.
    """
    nU, k = Theta.shape
    nI = Beta.shape[0]
    hy = cavi.Hyper(k, a, a_prime, b_prime, c, c_prime, d_prime)
    tick = _phase_clock()
    if verbose > 0:
        print("Initializing parameters...")
    full_updates = (users_per_batch == 0) and (items_per_batch == 0)
    if not full_updates:
        draw = start_init_draw(_make_ops(), None, random_seed, nU, nI, k)
        outs = [np.empty((n, w), dtype=np.float32) for n, w in ((nU, k), (nU, k), (nI, k), (nI, k), (nU, 1), (nI, 1))]
        return svi.fit_hpf_svi(hy, Y, ix_u, ix_i, Theta, Beta, *outs, maxiter, stop_crit, check_every, stop_thr,
                               users_per_batch, items_per_batch, step_size, save_folder, random_seed, verbose,
                               has_valset, Yval, ix_u_val, ix_i_val, full_llk, keep_all_objs, _make_ops,
                               device_triplets=device_triplets, init_draw=draw, resident=resident, tick=tick)

    # the reference's initialisation (4 numpy RNG passes over (nU+nI)*k floats, 0.36 s on the host at C3) is drawn on
    # the device from the same MT19937 stream, bit for bit: the sequential recurrence on a side stream under the
    # CSR/CSC build, the tables from its words afterwards (cavi.init_state); nothing of the state is uploaded
    eng = _Engine(hy, Y, ix_u, ix_i, nU, nI, Yval if has_valset else None, ix_u_val, ix_i_val,
                  device_triplets=device_triplets, random_seed=random_seed, draw_init=True)
    tick("triplets to the device, CSR/CSC layout (+ the MT19937 recurrence)")
    eng.init_state()
    tick("initial tables from the drawn words")
    model = eng.model
    errs = np.zeros(2, dtype=np.longdouble)
    last_crit = -np.inf
    Theta_prev = model.Theta.clone() if stop_crit == "diff-norm" else None

    def evaluate(final=False):
        """llk + rmse the way assess_convergence / eval_after_term compute them (PXI:66-79, 99-112)."""
        if has_valset and eng.val is not None:
            t = model.pair_llk_terms(eng.val[0], eng.val[1], eng.val[2], full_llk)
            if final:
                # PXI:105 subtracts Theta[ix_u_val].sum(0).Beta[ix_i_val].sum(0) here (sic)
                sub = _val_colsum_dot(eng)
            else:
                sub = t[2]
            errs[0] = np.longdouble(t[0]) - np.longdouble(sub)
            errs[1] = np.sqrt(np.longdouble(t[1]) / eng.nval_global)
        else:
            t = model.llk_terms(full_llk)
            errs[0] = np.longdouble(t[0]) - np.longdouble(model.colsum_dot())
            errs[1] = np.sqrt(np.longdouble(t[1]) / eng.nnz_global)

    if verbose > 0:
        print("Initializing optimization procedure...")
    st_time = time.time()
    i = -1
    while i + 1 < maxiter:
        # the six [n,k] state tables (shapes, rates, Theta/Beta) are outputs / llk inputs only: written on
        # check iterations (the loop may stop there) and on the last one; the E tables, k_rte/t_rte and the
        # column sums that carry the iteration are current after every iteration.  The iterations in between go
        # to the engine in one call (it may replay them from a captured hipGraph, cavi.iterate_many).
        nxt = maxiter - 1
        if check_every > 0:
            nxt = min(nxt, (i + 1) + (check_every - 1 - ((i + 1) % check_every)))
        model.iterate_many(nxt - (i + 1), store=False)
        i = nxt
        is_check = check_every > 0 and ((i + 1) % check_every) == 0
        model.iterate(store=True)
        if is_check:
            if stop_crit == "diff-norm":
                last_crit = eng.theta_norm_diff(Theta_prev)
                if verbose:
                    _print_norm_diff(i + 1, check_every, last_crit)
                if last_crit < stop_thr:
                    break
                Theta_prev.copy_(model.Theta)
            else:
                evaluate()
                if verbose:
                    _print_llk_iter(i + 1, errs[0], float(errs[1]), has_valset)
                if stop_crit != "maxiter":
                    if (i + 1) == check_every:
                        last_crit = errs[0]
                    else:
                        if (1.0 - errs[0] / last_crit) <= stop_thr:
                            break
                        last_crit = errs[0]

    last_llk = None
    if stop_crit in ("diff-norm", "maxiter") and verbose > 0:
        evaluate(final=True)
        last_llk = errs[0]
    minutes = (time.time() - st_time) / 60.0
    if verbose:
        _print_final_msg(i + 1, errs[0], float(errs[1]), minutes)
    tick("iterations and checks")

    if resident is not None and eng.dist is None and keep_all_objs and save_folder == "":
        model.materialize_rates()         # (also flushes a deferred item side)
        names = ("Gamma_shp", "Gamma_rte", "Lambda_shp", "Lambda_rte", "Theta", "Beta", "eT", "eB", "k_rte", "t_rte")
        resident.adopt(svi.DeviceModel(eng.ops, k, nU, nI, tables={n: getattr(model, n) for n in names}))
        tick("state handed over on the device")
        return i, None, last_llk

    eng.gather_users("Theta", out=Theta)        # (straight into the caller's arrays, PXI:140-141,251,256)
    model.fetch("Beta", out=Beta)
    temp = None
    if keep_all_objs or save_folder != "":
        temp = (eng.gather_users("Gamma_shp"), eng.gather_users("Gamma_rte"), model.fetch("Lambda_shp"),
                model.fetch("Lambda_rte"), eng.gather_users("k_rte"), model.fetch("t_rte"))
    if save_folder != "" and eng.rank == 0:
        save_parameters(verbose, save_folder,
                        ["Theta", "Beta", "Gamma_shp", "Gamma_rte", "Lambda_shp", "Lambda_rte", "kappa_rte", "tau_rte"],
                        [Theta, Beta] + list(temp))
    if not keep_all_objs:
        temp = None
    if eng.dist:
        model.release_exchange()      # (peer-mapped memory is freed behind a barrier, never by the garbage collector)
    tick("outputs to the host")
    return i, temp, last_llk


def _val_colsum_dot(eng):
    m = eng.model
    vu, vi, _ = eng.val
    a = m.Theta[vu.long()].sum(dim=0)
    if eng.dist:
        eng.dist.all_reduce(a)
    b = m.Beta[vi.long()].sum(dim=0)
    if eng.dist:
        eng.dist.all_reduce(b)
    return float(np.dot(a[: m.k].cpu().numpy(), b[: m.k].cpu().numpy()))


# -- PXI:423-473 --------------------------------------------------------------------------
def partial_fit(Y_batch, ix_u_batch, ix_i_batch, Theta, Beta, Gamma_shp, Gamma_rte, Lambda_shp, Lambda_rte, k_rte,
                t_rte, add_k_rte, add_t_rte, a, c, k_shp, t_shp, k, users_this_batch, items_this_batch, par_sh,
                step_size_batch, multiplier_batch, nthreads, user_batch):
    """One SVI step on caller-supplied triplets; mutates all eight arrays in place (PXI:443-473)."""
    svi.partial_fit_step(_make_ops(), Y_batch, ix_u_batch, ix_i_batch, Theta, Beta, Gamma_shp, Gamma_rte,
                         Lambda_shp, Lambda_rte, k_rte, t_rte, add_k_rte, add_t_rte, a, c, k_shp, t_shp, int(k),
                         users_this_batch, items_this_batch, step_size_batch, multiplier_batch, bool(user_batch))


def _pair_operands(M1, M2, ix_u, ix_i, dev):
    """Device operands for the listed-pair kernels.  Few pairs relative to the tables: ship only the
    rows the pairs touch (host-side row gather is indexing, not arithmetic); many pairs: ship the tables."""
    k = int(M1.shape[1])
    ld = cavi._lib.ld_for_k(k)
    n = int(ix_u.shape[0])
    iu = np.ascontiguousarray(ix_u).astype(np.int64)
    ii = np.ascontiguousarray(ix_i).astype(np.int64)
    if n and (int(iu.max()) >= M1.shape[0] or int(ii.max()) >= M2.shape[0]):
        raise ValueError("user/item id out of range")
    if 2 * n < M1.shape[0] + M2.shape[0]:
        T = _padded(M1[iu], ld, dev)
        B = _padded(M2[ii], ld, dev)
        ar = torch.arange(n, dtype=torch.int32, device=dev)
        return T, B, ar, ar, k, ld
    return (_padded(M1, ld, dev), _padded(M2, ld, dev), torch.from_numpy(iu).to(dev).to(torch.int32),
            torch.from_numpy(ii).to(dev).to(torch.int32), k, ld)


# -- PXI:525-534 --------------------------------------------------------------------------
def calc_llk(Y, ix_u, ix_i, Theta, Beta, k, nthreads, full_llk):
    """sum_n Y_n log(yhat_n) [- lgamma(Y_n+1)] - sum_n yhat_n over the listed pairs (HPF.eval_llk)."""
    ops = _make_ops()
    T, B, iu, ii, k, ld = _pair_operands(Theta, Beta, ix_u, ix_i, ops.device)
    y = torch.from_numpy(np.ascontiguousarray(Y, dtype=np.float32)).to(ops.device)
    t = ops.pair_llk(T, B, iu, ii, y, k, ld, bool(full_llk)).cpu().numpy()
    return np.longdouble(t[0]) - np.longdouble(t[2])


# -- PXI:538-543 --------------------------------------------------------------------------
def predict_arr(M1, M2, ix_u, ix_i, nthreads):
    ops = _make_ops()
    T, B, iu, ii, k, ld = _pair_operands(M1, M2, ix_u, ix_i, ops.device)
    out = torch.zeros(iu.shape[0], dtype=torch.float32, device=ops.device)
    ops.pair_dot(T, B, iu, ii, out, k, ld)
    return out.cpu().numpy()


def _padded(host_arr, ld, dev):
    n, k = host_arr.shape
    t = torch.zeros((n, ld), dtype=torch.float32, device=dev)
    t[:, :k] = torch.from_numpy(np.ascontiguousarray(host_arr, dtype=np.float32)).to(dev)
    return t


# -- not in the reference's extension: the scoring product of HPF.topN (hpfrec/__init__.py:1337-1356) ----
def top_items(theta_row, Beta, n, exclude=None):
    """Ids of the n rows of Beta with the largest theta_row . Beta[i], best first, optionally skipping
    `exclude` (ids).  `Beta`: the padded device table [nitems][ld] (hpfrec_amd.HPF keeps it resident,
    hpfrec_amd.resident) or a host array [nitems][k] (uploaded for this call: nothing is cached behind the
    caller's back).  One GEMV over Beta, a mask and a top-k -- the reference does a host GEMV + argpartition +
    setdiff1d + argsort per query."""
    ops = _make_ops()
    dev = ops.device
    if torch.is_tensor(theta_row):                    # a padded row [ld] of the resident user table (pads are zero)
        assert torch.is_tensor(Beta)
        ld = int(Beta.shape[1])
        k, vec = ld, theta_row.contiguous()
        assert vec.shape[0] == ld
    else:
        k = int(np.asarray(theta_row).reshape(-1).shape[0])
        ld = cavi._lib.ld_for_k(k)
        vec = torch.zeros(ld, dtype=torch.float32, device=dev)
        vec[:k] = torch.from_numpy(np.ascontiguousarray(theta_row, dtype=np.float32).reshape(-1)).to(dev)
    tab = Beta if torch.is_tensor(Beta) else _padded(Beta, ld, dev)
    assert tab.shape[1] == ld
    scores = torch.empty(tab.shape[0], dtype=torch.float32, device=dev)
    ops.score_rows(vec, tab, scores, k, ld)
    n_avail = int(tab.shape[0])
    if exclude is not None and len(exclude) > 0:
        if torch.is_tensor(exclude):                  # (a slice of the seen-items list: no duplicates, on the device)
            ex = exclude.to(torch.int64)
        else:
            ex = torch.from_numpy(np.unique(np.asarray(exclude).astype(np.int64))).to(dev)
        scores[ex] = -float("inf")
        n_avail -= int(ex.shape[0])
    n = int(max(0, min(n, n_avail)))
    if n == 0:
        return np.empty(0, dtype=np.int64)
    return torch.topk(scores, n, largest=True, sorted=True).indices.cpu().numpy()


def pair_dots_device(T, B, ix_u, ix_i, k):
    """predict_arr on tables that are already on the device (padded [rows][ld])."""
    ops = _make_ops()
    ld = cavi._lib.ld_for_k(int(k))
    iu = torch.from_numpy(np.ascontiguousarray(ix_u).astype(np.int64)).to(ops.device).to(torch.int32)
    ii = torch.from_numpy(np.ascontiguousarray(ix_i).astype(np.int64)).to(ops.device).to(torch.int32)
    if iu.numel() and (int(iu.max()) >= T.shape[0] or int(ii.max()) >= B.shape[0]):
        raise ValueError("user/item id out of range")
    out = torch.zeros(iu.shape[0], dtype=torch.float32, device=ops.device)
    ops.pair_dot(T, B, iu, ii, out, int(k), ld)
    return out.cpu().numpy()


def calc_llk_device(Y, ix_u, ix_i, T, B, k, full_llk):
    """calc_llk on tables that are already on the device (padded [rows][ld])."""
    ops = _make_ops()
    ld = cavi._lib.ld_for_k(int(k))
    iu = torch.from_numpy(np.ascontiguousarray(ix_u).astype(np.int64)).to(ops.device).to(torch.int32)
    ii = torch.from_numpy(np.ascontiguousarray(ix_i).astype(np.int64)).to(ops.device).to(torch.int32)
    if iu.numel() and (int(iu.max()) >= T.shape[0] or int(ii.max()) >= B.shape[0]):
        raise ValueError("user/item id out of range")
    y = torch.from_numpy(np.ascontiguousarray(Y, dtype=np.float32)).to(ops.device)
    t = ops.pair_llk(T, B, iu, ii, y, int(k), ld, bool(full_llk)).cpu().numpy()
    return np.longdouble(t[0]) - np.longdouble(t[2])


# -- PXI:476-520 --------------------------------------------------------------------------
def calc_user_factors(a, a_prime, b_prime, c, c_prime, d_prime, Y, ix_i, Theta, Beta, Lambda_shp, Lambda_rte, nY, k,
                      maxiter, nthreads, random_seed, stop_thr, return_all, resident=None):
    """Fold-in of one new user with item parameters fixed (HPF.predict_factors / add_user).  `resident` (not in
    the reference's signature): a DeviceModel already holding the item tables, see svi.calc_user_factors."""
    ops = _make_ops()
    return svi.calc_user_factors(ops, a, a_prime, b_prime, c, c_prime, d_prime, Y, ix_i, Theta, Beta,
                                 Lambda_shp, Lambda_rte, int(nY), int(k), int(maxiter), int(random_seed),
                                 float(stop_thr), bool(return_all), resident=resident)
