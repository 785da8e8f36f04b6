"""The user-sharded (multi-GPU) half of cavi.FullBatchCavi: SURVEY.md section 8(e).

One process per GPU.  Users are sharded in contiguous ranges balanced by nonzeros, the item E table is replicated, and per
iteration the item statistics are summed over the ranks ("scatter" exchange): every rank ends up with the global
statistics of 1/N of the items (its slices of the item ranges), finishes those, and every rank gets all finished rows.
Both sweeps of an iteration read only last iteration's E tables, so the ITEM sweep goes first and the user side runs
under the exchange.  The reference has no counterpart (single-node OpenMP, /root/reference/hpfrec/cython_loops.pxi:4);
the statements being distributed are PXI:236-259.

ONE switch, HPF_SCHEDULE, picks how the exchange is carried:

  direct                the default on GPUs: no collective library.  The exchange buffers live in peer-mapped memory
                        (hpfrec_amd/p2p.py); the kernels of the iteration pull the peers' rows themselves, ordered by flag
                        words (include/hpf_hip.h, HPF_SCHEDULE_DIRECT).  C-issued only.
  gather-early          RCCL reduce-scatter per item range, split item finalizer, ONE all-gather of the [numerators |
                        base rate] rows under the user sweep (HPF_SCHEDULE_GATHER_EARLY); the fallback when peer mapping is
                        not available, and -- issued call by call, in order -- what gloo / CPU runs execute.
  finalize-then-gather  reduce-scatter, one-part finalizer after the user side, all-gather of the new E rows per range
                        (HPF_SCHEDULE_FINALIZE_THEN_GATHER); Python form: overlapped on an exchange stream.

On a GPU the whole iteration is issued by ONE C call (hpf_hip_shard_iterate; HPF_NATIVE_SHARD=0 or model.native = False:
call by call from Python).  The first C-issued iterations of a process on real links are CHECKED against the call-by-call
form on the same state (every rank votes); a failure strikes the schedule for the process and moves every rank on to the
next C-issued one (direct -> gather-early -> finalize-then-gather), the call-by-call form being the last resort.
"""
import contextlib
import os
import warnings

import torch

from . import _streams, layout, p2p

SCHEDULES = ("direct", "gather-early", "finalize-then-gather")
_EARLY = ("direct", "gather-early")      # split item finalizer: [numerators | base] payload rows
_DIRECT_COMMS = {}
_VERIFIED = {}                  # (schedule, world, device) -> bool: the first-iterations check of this process
_PASSED = {}                    # ... -> checked iterations passed so far
CHECKED_ITERATIONS = 3          # the first C-issued iterations of a process that are checked (see _needs_first_check)
_FAILED = set()                 # the same keys: schedules whose check failed -- not set up again in this process
NATIVE_PLANS_CREATED = [0]      # how many models of this process run their sharded iteration from C (tests, bench)
LAST_SCHEDULE = [None]          # "<schedule>" / "<schedule>, call by call" of the last model that set its exchange up
_STATUS_EVERY = int(os.environ.get("HPF_DIRECT_STATUS_EVERY", "256"))
_side_stream = _streams.side_stream


def requested_schedule():
    s = os.environ.get("HPF_SCHEDULE", "auto")
    if s not in SCHEDULES + ("auto",):
        raise ValueError("HPF_SCHEDULE must be one of %s or auto, not %r" % (", ".join(SCHEDULES), s))
    return s


class ShardedMixin:
    """State and schedules of the sharded iteration; mixed into cavi.FullBatchCavi (which owns the tables)."""

    # ---- set-up ---------------------------------------------------------------------------------------------------
    def _init_sharded(self, ops):
        self.world = self.dist.get_world_size() if self.dist else 1
        self.rank = self.dist.get_rank() if self.dist else 0
        self.shard_mode = "scatter" if self.dist else None
        self.schedule = None
        self._plan = None
        self._region = None
        self.comm = None
        self.native = True            # (False: issue call by call even when a plan exists -- bench's per-kernel event pass)
        self._last_native = False
        self.native_error = None
        self._chunk_views = None
        self._tables_split = False
        if not self.dist:
            self.item_bounds = self.item_chunks = None
            self.nI_alloc = self.nI
            return
        want = requested_schedule()
        cuda = self.device.type == "cuda"
        c_issue = cuda and os.environ.get("HPF_NATIVE_SHARD", "1") == "1"
        self._want_auto = want == "auto"
        if want == "auto":
            want = "direct" if c_issue else "gather-early"
        if want == "direct" and not c_issue:
            raise ValueError("HPF_SCHEDULE=%s is issued from C only: it needs a GPU and HPF_NATIVE_SHARD != 0" % want)
        self._want = want
        self.schedule = want
        # item ranges per iteration: the exchange of one range runs under the sweep of the next; each extra range costs
        # launches and stream dependencies (0.06 ms per iteration at 8 ranks): two
        nranges = int(os.environ.get("HPF_ITEM_RANGES", "2"))
        self.item_bounds = self._item_bounds(nranges)
        self.nI_alloc = self.item_bounds[-1][1]       # item tables carry a few pad rows: every range splits into N slices

    @property
    def gather_early(self):
        return self.schedule in _EARLY

    def _item_bounds(self, nchunks):
        """Contiguous item ranges [(lo, hi)], identical on every rank (cut on all-reduced degrees), each a multiple of
        4 x the world size long; the last one runs past nI into pad rows."""
        it = self.items
        deg = (it.indptr[1:] - it.indptr[:-1]).clone()
        self.dist.all_reduce(deg)
        # HPF_RANGE_ROW_WEIGHT = w: a row counts as its nonzeros + w x the mean row's (0: equal nonzeros = equal sweep
        # time, 18 % / 82 % of the rows at C3; large: equal rows = equal exchange bytes).  2 (31 % / 69 % of the
        # nonzeros): with equal nonzeros the exchange of the range swept first (82 % of the bytes) ended after the
        # iteration did (profiles/r03_shard_probe_gather_carried.txt, "range split")
        w = float(os.environ.get("HPF_RANGE_ROW_WEIGHT", "2"))
        if w > 0 and self.nI > 0:
            deg = deg + int(round(w * float(deg.sum().item()) / self.nI))
        gptr = torch.zeros(self.nI + 1, dtype=torch.int64, device=deg.device)
        torch.cumsum(deg, 0, out=gptr[1:])
        cuts = [lo for lo, _ in layout.nnz_balanced_ranges(gptr, max(1, nchunks))] + [self.nI]
        W, fixed = 4 * self.world, [0]      # (4 x world: a slice then starts on a 16-byte boundary of the packed [.][k]
        for c in cuts[1:-1]:                #  buffer for every k -- the pull-reduce reads it with 16-byte loads)
            c = fixed[-1] + ((c - fixed[-1] + W - 1) // W) * W
            if fixed[-1] < c < self.nI:
                fixed.append(c)
        fixed.append(fixed[-1] + ((self.nI - fixed[-1] + W - 1) // W) * W)
        return [(lo, hi) for lo, hi in zip(fixed[:-1], fixed[1:]) if hi > lo]

    def _item_chunks(self, early):
        """[(row_lo, row_hi, SideView over the range's segments, its split rows)] in issue order."""
        from .cavi import _SideView
        it = self.items
        rsp = it.row_seg_ptr.cpu()
        ptr = it.indptr.cpu()
        out = []
        for lo, hi in self.item_bounds:
            top = min(hi, self.nI)
            multi = it.multi_rows[(it.multi_rows >= lo) & (it.multi_rows < top)]
            # only rows cut into SEVERAL segments need their partial rows summed into the exchange buffer; a row without
            # local nonzeros (most tail items on a rank of many) keeps the zeros the buffer was allocated with -- nothing
            # else ever writes a row of acc_i
            multi = multi[(it.row_seg_ptr[multi + 1] - it.row_seg_ptr[multi]) >= 2].contiguous()
            out.append((lo, hi, _SideView(it, int(rsp[lo]), int(rsp[top]), nnz=int(ptr[top] - ptr[lo])), multi))
        if early:
            # MOST rows first: the whole exchange runs under what is left of the iteration, so the range with most of the
            # bytes must start after the FIRST sweep (profiles/r03_timeline_links_*.txt)
            out.sort(key=lambda c: c[0] - c[1])
        else:
            # finalize-then-gather: fewest rows first -- the all-gather of the big range then overlaps the sweep of the
            # small one in the next iteration
            out.sort(key=lambda c: c[1] - c[0])
        return out

    # ---- buffers + plan ---------------------------------------------------------------------------------------------
    def _scatter_views(self):
        """Per item range: the range, this rank's slice of it and the exchange buffers; then the C plan.  Schedules are
        tried in order of preference until one can run on EVERY rank."""
        if self._chunk_views is not None:
            return self._chunk_views
        cuda = self.device.type == "cuda"
        native_wanted = cuda and os.environ.get("HPF_NATIVE_SHARD", "1") == "1" and self.fused
        if self._want == "direct" and not native_wanted and self._want_auto:
            # the default resolved to the C-issued-only exchange, but this model issues call by call (set_fused(False),
            # HPF_NATIVE_SHARD=0 set after construction): the in-order split-finalizer form, like a CPU / gloo run
            self._want = "gather-early"
        order = [self._want]
        if native_wanted:      # what to fall back to when the preferred schedule cannot get a plan on every rank
            order += {"direct": ["gather-early", "finalize-then-gather"],
                      "gather-early": ["finalize-then-gather"]}.get(self._want, [])
        # (a schedule whose first-iteration check failed in this process is not tried again; the verdict was every rank's)
        struck = [s_ for s_ in order if (s_, self.world, str(self.device)) in _FAILED]
        order = [s_ for s_ in order if s_ not in struck]
        if not order:
            # every C-issued schedule this model could use has produced wrong results in this process: no plan at all,
            # the call-by-call form on torch.distributed collectives (the checker itself) carries the fit
            warnings.warn("hpfrec_amd: every C-issued schedule (%s) failed its first-iteration check in this process; "
                          "iterating call by call" % ", ".join(struck))
            order, native_wanted = ["gather-early" if "gather-early" in struck or "direct" in struck
                                    else "finalize-then-gather"], False
        errors = []
        for idx, sched in enumerate(order):
            self.schedule = sched
            try:
                self._chunk_views = self._alloc_exchange(sched)
            except p2p.P2PError as exc:      # (raised on every rank or on none: PeerRegion votes before it connects)
                errors.append("%s: %s" % (sched, str(exc)[:200]))
                self._plan = None
                if idx + 1 == len(order):
                    break
                continue
            self._plan, err = self._make_plan(self._chunk_views) if native_wanted else (None, None)
            if err:
                errors.append("%s: %s" % (sched, err))
            if self._plan is not None or idx + 1 == len(order):
                break
            if sched == "direct" or err:
                continue        # C-issued only / a plan was expected and failed: the next preference (the last one,
            break               # finalize-then-gather, has the overlapped call-by-call form)
        if self._plan is None and self.schedule == "direct":
            raise RuntimeError("hpfrec_amd: the %s schedule is C-issued only and no plan could be created (%s)"
                               % (self.schedule, "; ".join(errors) or "HPF_NATIVE_SHARD=0 / unfused"))
        self.native_error = "; ".join(errors) if errors else None
        LAST_SCHEDULE[0] = self.schedule + ("" if self._plan is not None else ", call by call")
        if errors:
            warnings.warn("hpfrec_amd: sharded schedule fell back to %s%s (%s)"
                          % (self.schedule, "" if self._plan is not None else ", call by call from Python", self.native_error))
        return self._chunk_views

    def _alloc_exchange(self, sched):
        ops, ld, k, W, r = self.ops, self.ld, self.k, self.world, self.rank
        f32 = dict(dtype=torch.float32, device=self.device)
        cuda = self.device.type == "cuda"
        early = sched in _EARLY
        self.item_chunks = self._item_chunks(early)
        nIa = self.nI_alloc
        total = sum((hi - lo) // W for lo, hi, _, _ in self.item_chunks)
        e_ld = ops.gather_payload_ld(k) if early else ld     # row stride of the send buffer
        if self._region is not None:
            self._region.close()
            self._region = None
        if sched == "direct":
            # the packed accumulators and the finished rows live in ONE peer-mapped allocation
            acc_bytes = ((nIa * k * 4 + 255) // 256) * 256
            send_bytes = ((total * e_ld * 4 + 255) // 256) * 256
            local = bool(getattr(self.dist, "native_dry_run", False)) or W == 1
            self._region = p2p.PeerRegion(self.device, acc_bytes + send_bytes, ld, dist=self.dist, rank=r, world=W,
                                          local=local,
                                          timeout_ms=float(os.environ.get("HPF_DIRECT_TIMEOUT_MS", "20000")))
            self._region_offsets = (0, acc_bytes)
            self.acc_i = self._region.tensor(0, (nIa, k))
            self.e_own_all = self._region.tensor(acc_bytes, (total, e_ld))
        else:
            self.acc_i = torch.zeros((nIa, k), **f32)
            self.e_own_all = torch.zeros((total, e_ld), **f32)
        self.acc_own_all = torch.zeros((total, k), **f32)
        if early:                                     # [k numerators | base rate] rows of every owner
            self.ag_recv_all = torch.ones((W * total, e_ld), **f32)
            self.shp_own_all = torch.zeros((total, ld), **f32)      # the shapes, between the finalizer's two halves
        views, t0, ranges = [], 0, []
        for lo, hi, view, multi in self.item_chunks:
            m = (hi - lo) // W
            o0 = lo + r * m
            n_real = max(0, min(m, self.nI - o0))
            if n_real > 0:
                ranges.append((n_real, t0, o0))
            views.append(dict(
                lo=lo, hi=hi, m=m, o0=o0, o1=o0 + m, n_real=n_real, view=view, multi=multi, nmulti=int(multi.shape[0]),
                part=self.part_i[view.seg_lo:], acc=self.acc_i[lo:hi], acc_own=self.acc_own_all[t0:t0 + m],
                e_own=self.e_own_all[t0:t0 + m], eB_range=self.eB[lo:hi],
                # dedicated events (re-recorded every iteration, waited for before the next record)
                sw_done=torch.cuda.Event() if cuda else None, ag_done=torch.cuda.Event() if cuda else None))
            t0 += m
        self._fin_ranges = ranges
        self._range_rows = [(c["lo"], c["hi"]) for c in views]       # in issue order (= slice order inside a rank's block)
        grid = ops.finalize_grid(max(1, sum(n for n, _, _ in ranges)))
        if early:      # (the apply kernel streams ALL item rows: gx blocks for each rank's block, a multiple of the world size)
            # (2048 workgroups at C3: 1024 and 2048 are equal, 512 and 4096 slower -- tools/apply_probe.py)
            grid = W * max(len(self.item_chunks), -(-2 * ops.finalize_grid(self.nI) // W))
        self.csB_part_sc = torch.zeros((grid, ld), **f32)
        self._csT_ready = torch.cuda.Event() if cuda else None
        self._sc_fresh = True
        return views

    def _make_plan(self, views):
        """(plan, error): the C-issued form of this model's schedule (hpf_hip_shard_iterate) over its tensors.  Needs a way
        to run the exchange from C: the peer-mapped region (direct), RCCL (backend "nccl": a communicator of our own), or
        what a stand-in for torch.distributed brings (`native_collective`: a callback, tests with gloo ranks;
        `native_dry_run`: this rank alone, probes).  Every rank ends up with a plan, or none does: rank-local failures go
        through the vote, only conditions identical on every rank return early."""
        dist = self.dist
        if len(views) > 8:
            return None, "more than 8 item ranges"
        sched = self.schedule
        standin_cb, standin_dry = hasattr(dist, "native_collective"), bool(getattr(dist, "native_dry_run", False))
        if sched != "direct" and not (standin_cb or standin_dry):
            try:
                if dist.get_backend() != "nccl":
                    return None, None                   # (gloo: the call-by-call form is the path; not an error)
            except Exception:   # noqa: BLE001
                return None, None
        plan, err = None, None
        try:
            from . import rccl, shard_native as sn
            coll = comm = None
            dry = 0
            keep = []
            if sched == "direct":
                dry = 1 if self._region.local and self.world > 1 else 0
            elif standin_cb:
                coll = sn.COLLECTIVE_FN(dist.native_collective(self))
                keep.append(coll)
            elif standin_dry:
                dry = 1
                comm = dist.direct_comm(self.device, raw=True) if hasattr(dist, "direct_comm") else None
            else:
                key = (str(self.device), self.world, self.rank)
                if key not in _DIRECT_COMMS:
                    _DIRECT_COMMS[key] = rccl.DirectComm(self.device, dist, self.rank, self.world)
                comm = self.comm = _DIRECT_COMMS[key]
                if not comm.self_check():
                    raise RuntimeError("the communicator's self-check failed")
            d = sn.ShardDesc()
            hy, u, it = self.hy, self.users, self.items
            d.world, d.rank, d.k, d.ld, d.nU, d.nI = self.world, self.rank, self.k, self.ld, self.nU, self.nI
            d.u_segs, d.u_nseg, d.u_idx, d.u_y = u.segs.data_ptr(), u.nseg, u.idx.data_ptr(), u.y.data_ptr()
            d.u_row_seg_ptr, d.u_nmulti = u.row_seg_ptr.data_ptr(), u.nmulti
            d.u_multi_rows = u.multi_rows.data_ptr() if u.nmulti else None
            d.i_segs, d.i_idx, d.i_y, d.i_row_seg_ptr = (it.segs.data_ptr(), it.idx.data_ptr(), it.y.data_ptr(),
                                                         it.row_seg_ptr.data_ptr())
            d.nranges = len(views)
            for j, c in enumerate(views):
                r = d.ranges[j]
                r.lo, r.hi, r.seg_lo, r.nseg = c["lo"], c["hi"], c["view"].seg_lo, c["view"].nseg
                r.nmulti, r.short_rows = c["nmulti"], int(c["view"].short_rows)
                r.multi_rows = c["multi"].data_ptr() if c["nmulti"] else None
            for n in ("eB", "part_u", "part_i", "Gamma_shp", "Theta", "k_rte", "k_rte_prev", "Lambda_shp", "Beta", "t_rte",
                      "t_rte_prev", "csT", "csB", "csB_used", "csT_part", "acc_i"):
                setattr(d, n, getattr(self, n).data_ptr())
            d.csT_part_rows, d.user_sweep_grid = int(self.csT_part.shape[0]), self.gsu
            d.user_multi_grid = max(1, min(self.gu, (u.nmulti + 3) // 4))
            d.csB_part, d.csB_part_rows = self.csB_part_sc.data_ptr(), int(self.csB_part_sc.shape[0])
            d.acc_own, d.e_own = self.acc_own_all.data_ptr(), self.e_own_all.data_ptr()
            d.e_own_ld, d.item_sweep_grid = int(self.e_own_all.shape[1]), int(self.item_sweep_blocks)
            d.schedule = {"finalize-then-gather": 0, "gather-early": 1, "direct": 3}[sched]
            if sched in _EARLY:
                d.ag_recv, d.shp_own = self.ag_recv_all.data_ptr(), self.shp_own_all.data_ptr()
            if sched == "direct":
                d.p2p_region = self._region.handle.value
                d.p2p_acc_offset, d.p2p_send_offset = self._region_offsets
                d.direct_prefetch = int(os.environ.get("HPF_DIRECT_PREFETCH", "1") == "1")
                # the pulls run BESIDE the user sweep and are bound by the links, not by their grid: one workgroup per CU
                # for a slice's pull-reduce; the pull of the finished rows polls per owner, so few workgroups per owner
                cus = max(1, getattr(self.ops, "cu_count", 256))
                d.direct_pull_grid = cus                                  # (64-512 workgroups measured alike)
                d.direct_gather_gx = max(1, (cus // 2) // self.world)
            d.a, d.k_shp, d.add_k_rte = float(hy.a), float(hy.k_shp), float(hy.add_k_rte)
            d.c, d.t_shp, d.add_t_rte = float(hy.c), float(hy.t_shp), float(hy.add_t_rte)
            d.comm = comm.handle if comm is not None else None
            if coll is not None:
                d.coll = coll
            d.xstream = self._xstream().cuda_stream
            d.dry_run = dry
            if dry:      # (probes: hold the streams for the time real links would take, at an assumed bus bandwidth)
                d.dry_run_busbw_GBps = float(getattr(dist, "native_dry_run_busbw", 0.0))
                d.dry_run_latency_us = float(getattr(dist, "native_dry_run_latency_us", 0.0))
                d.dry_run_footprint_blocks = int(getattr(dist, "native_dry_run_footprint_blocks", 0))
            plan = sn.ShardPlan(d, keep=keep + [comm, views, self._region])
        except Exception as exc:   # noqa: BLE001
            plan, err = None, "%s: %s" % (type(exc).__name__, str(exc)[:200])
        # all or none (a rank that issued its exchange in another form than its peers would hang them)
        if self.world > 1 and hasattr(dist, "all_reduce") and not standin_dry:
            ok = torch.tensor([1.0 if plan is not None else 0.0], device=self.device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if float(ok.item()) < 1.0 and plan is not None:
                plan.close()
                plan, err = None, "another rank could not create its plan"
        if plan is not None:
            NATIVE_PLANS_CREATED[0] += 1
        return plan, err

    # ---- the iteration ------------------------------------------------------------------------------------------------
    def _iterate_scatter(self, store):
        """One sharded iteration: from C when a plan exists (checked once per process against the call-by-call form), else
        call by call."""
        self._scatter_views()
        if self._plan is not None and self.native and self.fused:
            if self._needs_first_check():
                return self._first_iteration_checked(store)
            return self._iterate_native(store)
        return self._iterate_python(store)

    def _iterate_native(self, store):
        if not self._last_native:
            self._sync_scatter_streams()      # (switching forms: the other one's exchanges first)
        self._last_native = True
        self._plan.iterate(self.eT, self.eT_next, store, torch.cuda.current_stream(self.device).cuda_stream)
        if self.schedule == "direct" and _STATUS_EVERY > 0:
            # a flag wait that ran into HPF_DIRECT_TIMEOUT_MS sets the region's error word and every later wait is skipped:
            # the iterations after it compute on garbage.  Look at the word now and then (one device synchronisation per
            # HPF_DIRECT_STATUS_EVERY iterations) so that a long fit without llk checks stops instead of going on
            self._since_status = getattr(self, "_since_status", 0) + 1
            if self._since_status >= _STATUS_EVERY:
                self._since_status = 0
                self._plan.status()
        self.rte_factored = True         # (the C call keeps colsum(Beta) on storing iterations)
        self._sc_fresh = False
        self._tables_split = True
        self.eT, self.eT_next = self.eT_next, self.eT
        self.niter_done += 1

    def _iterate_python(self, store):
        if self._last_native:
            self._sync_scatter_streams()
        self._last_native = False
        if self.gather_early:
            return self._iterate_gather_early(store)
        return self._iterate_finalize_then_gather(store)

    def _needs_first_check(self):
        """The C-issued iteration has met real links only in the driver's runs: its first CHECKED_ITERATIONS iterations in a
        process are each compared with the call-by-call form (torch.distributed collectives, in order) on the same state --
        more than one, because the failure a one-GPU test cannot show is a STALE read: a pull served from a cache line the
        previous iteration left behind, which the first iteration cannot have.  On for RCCL jobs with more than one rank;
        HPF_VERIFY_FIRST=1/0 forces it on (tests with gloo ranks) or off."""
        key = (self.schedule, self.world, str(self.device))
        if _VERIFIED.get(key) is True:
            return False
        if key in _FAILED:
            # struck earlier in this process and still holding a plan (nothing else was left to fall back to): never run it
            # unchecked -- _scatter_views builds no plan for a struck schedule, so this is a second line of defence
            raise RuntimeError("hpfrec_amd: the %s schedule failed its first-iteration check in this process" % self.schedule)
        flag = os.environ.get("HPF_VERIFY_FIRST")
        if flag is not None:
            on = flag == "1"
        else:
            try:
                on = self.world > 1 and self.dist.get_backend() == "nccl" and not getattr(self.dist, "native_dry_run", False)
            except Exception:   # noqa: BLE001
                on = False
        if not on:
            _VERIFIED[key] = True
        return on

    def _first_iteration_checked(self, store):
        key = (self.schedule, self.world, str(self.device))
        names = ("eB", "k_rte", "k_rte_prev", "t_rte", "t_rte_prev", "csB", "csT")
        snap = {n: getattr(self, n).clone() for n in names}
        done0 = self.niter_done
        self._iterate_python(store)
        self._sync_scatter_streams()
        ref = {n: getattr(self, n).clone() for n in ("eT", "eB", "csT", "csB")}
        self.eT, self.eT_next = self.eT_next, self.eT
        self.niter_done = done0
        for n, v in snap.items():
            getattr(self, n).copy_(v)
        err, worst = None, 0.0
        try:
            self._iterate_native(store)
            self._sync_scatter_streams()
            self._plan.status()
            for n, v in ref.items():
                got = getattr(self, n)
                worst = max(worst, float(((got - v).abs() / v.abs().clamp_min(1e-30)).max().item()))
            if not (worst < 1e-4):
                err = "first C-issued iteration differs from the call-by-call form (max rel %.3g)" % worst
        except Exception as exc:   # noqa: BLE001
            err = "%s: %s" % (type(exc).__name__, str(exc)[:200])
        ok = torch.tensor([0.0 if err else 1.0], device=self.device)
        self.dist.all_reduce(ok, op=self.dist.ReduceOp.MIN)
        good = float(ok.item()) > 0
        if self.schedule in os.environ.get("HPF_TEST_FAIL_FIRST_CHECK", "").split(","):     # (tests: the fall-back path)
            good, err = False, "failure injected by HPF_TEST_FAIL_FIRST_CHECK"
        prev = getattr(self, "first_check", None) or {}
        worst_all = max(worst, prev.get("max_rel_vs_call_by_call", 0.0) if prev.get("schedule") == self.schedule else 0.0)
        if good:
            _PASSED[key] = _PASSED.get(key, 0) + 1
            if _PASSED[key] >= CHECKED_ITERATIONS:
                _VERIFIED[key] = True
        else:
            _VERIFIED[key] = False
        self.first_check = {"schedule": self.schedule, "max_rel_vs_call_by_call": worst_all, "passed": good,
                            "iterations_checked": _PASSED.get(key, 0) + (0 if good else 1)}
        if good:
            return
        # Every rank takes the same way out (the vote was uniform): back to the state the check started with; the schedule
        # is struck for the rest of the process; the NEXT C-issued schedule is set up and gets its own check -- direct ->
        # gather-early on RCCL -> finalize-then-gather -- and only the last resort is the call-by-call form
        failed = self.schedule
        _FAILED.add(key)
        self.native_error = "%s: %s" % (failed, err or "the first-iteration check failed on another rank")
        self.first_checks_failed = getattr(self, "first_checks_failed", []) + [self.native_error]
        if self.niter_done != done0:
            self.eT, self.eT_next = self.eT_next, self.eT
            self.niter_done = done0
        for n, v in snap.items():
            getattr(self, n).copy_(v)
        torch.cuda.synchronize(self.device)
        nxt = {"direct": "gather-early", "gather-early": "finalize-then-gather"}.get(failed)
        if nxt is not None:
            warnings.warn("hpfrec_amd: %s; switching every rank to %s" % (self.native_error, nxt))
            self._plan.close()
            self._plan, self._chunk_views, self._want = None, None, nxt
            self._last_native, self._sc_fresh, self._tables_split = False, True, False
            return self._iterate_scatter(store)
        warnings.warn("hpfrec_amd: %s; continuing call by call" % self.native_error)
        self.native = False
        self._iterate_python(store)

    def _iterate_finalize_then_gather(self, store):
        """Call by call, overlapped.  Per item range (fewest rows first): sweep the local CSC slice into the packed exchange
        buffer, then a REDUCE-SCATTER on the exchange stream leaves each rank with the global statistics of its 1/N slice;
        the user side runs under the exchange.  Everything after it is ONE in-order chain on the exchange stream: the
        k-float all-reduce of colsum(Theta), the finalizer of this rank's slices (one dense launch), the ALL-GATHERS of the
        new E rows straight into the replicated E table -- each waited for only by the next iteration's sweep of that range
        -- and, behind the first of them, the k-float all-reduce of this rank's partial colsum(Beta).  Lambda_shp / Beta /
        t_rte are current on the owning rank only; flush_items() gathers them.  A cross-stream dependency costs ~15-20 us
        on the waiting stream (tools/handover_probe.py): the critical cycle crosses streams exactly twice."""
        ops, hy, k, ld, dist = self.ops, self.hy, self.k, self.ld, self.dist
        views = self._chunk_views
        xs = self._xstream()
        cuda = xs is not None
        cs = torch.cuda.current_stream(self.device) if cuda else None
        fresh = self._sc_fresh
        if cuda and fresh:
            xs.wait_event(self._mark(cs))      # first iteration after load_state: order after whatever the caller queued
        on = (lambda st: torch.cuda.stream(st)) if cuda else (lambda st: contextlib.nullcontext())
        for c in views:
            if cuda and not fresh:
                cs.wait_event(c["ag_done"])   # this range's E rows from the previous iteration's finalizers
            if c["view"].nseg > 0:
                ops.sweep(c["view"], self.eB, self.eT, c["part"], k, ld, acc_rows=self.acc_i, acc_ld=k,
                          grid_blocks=self.item_sweep_blocks)
            if c["nmulti"] > 0:
                ops.segsum(self.part_i, self.items.row_seg_ptr, c["nmulti"], self.acc_i, ld, row_list=c["multi"],
                           acc_ld=k, acc_by_row=True)
            if cuda:
                c["sw_done"].record(cs)
                xs.wait_event(c["sw_done"])
            with on(xs):
                dist.reduce_scatter_tensor(c["acc_own"], c["acc"])
        self._keep_csB(store)
        self._side_update(self.users, self.nU, self.eT, self.eB, self.eT_next, self.part_u, self.Gamma_shp,
                          self.k_rte_prev, self.Theta, self.k_rte, self.csB, self.csT_part, self.gsu, self.gu,
                          hy.a, hy.k_shp, hy.add_k_rte, store)
        ops.colsum_reduce(self.csT_part, self.csT, ld)
        if cuda:
            self._csT_ready.record(cs)
            xs.wait_event(self._csT_ready)
        with on(xs):
            dist.all_reduce(self.csT)
            if self._fin_ranges:
                ops.row_finalize_ranges(self.acc_own_all, self._fin_ranges, self.eB, self.e_own_all,
                                        self.Lambda_shp if store else None, None, self.Beta if store else None,
                                        self.t_rte, self.csT, self.csB_part_sc, hy.c, hy.t_shp, hy.add_t_rte, k, ld, k,
                                        rs_prev=self.t_rte_prev, e_new_ld=int(self.e_own_all.shape[1]))
            for j, c in enumerate(views):
                if j == len(views) - 1:
                    # colsum(Beta) -- read by the next USER side only -- goes ahead of the last all-gather
                    ops.colsum_reduce(self.csB_part_sc, self.csB, ld)      # this rank's partial colsum(Beta) ...
                    dist.all_reduce(self.csB)                              # ... summed over ranks
                dist.all_gather_into_tensor(c["eB_range"], c["e_own"])
                if cuda:
                    c["ag_done"].record(xs)
        self._sc_fresh = False
        self._tables_split = True
        self.eT, self.eT_next = self.eT_next, self.eT
        self.niter_done += 1

    def _iterate_gather_early(self, store):
        """The split-finalizer schedule call by call, IN ORDER on the current stream (gloo / stand-in runs, the checker of
        the first C-issued iteration, bench's per-kernel event pass): item sweeps + reduce-scatter per range; the shape half
        of the finalizer for this rank's slices; one all-gather of the [numerators | base rate] rows; the user side;
        colsum(Theta) summed over ranks; the rates applied to all items locally; colsum(Beta) summed over ranks."""
        ops, hy, k, ld, dist = self.ops, self.hy, self.k, self.ld, self.dist
        views = self._chunk_views
        self._sync_scatter_streams()
        for c in views:
            if c["view"].nseg > 0:
                ops.sweep(c["view"], self.eB, self.eT, c["part"], k, ld, acc_rows=self.acc_i, acc_ld=k,
                          grid_blocks=self.item_sweep_blocks)
            if c["nmulti"] > 0:
                ops.segsum(self.part_i, self.items.row_seg_ptr, c["nmulti"], self.acc_i, ld, row_list=c["multi"],
                           acc_ld=k, acc_by_row=True)
            dist.reduce_scatter_tensor(c["acc_own"], c["acc"])
        if self._fin_ranges:
            ops.item_shape_rows(self.acc_own_all, self._fin_ranges, self.eB, self.shp_own_all, self.e_own_all, self.t_rte,
                                hy.c, hy.t_shp, k, ld, rs_prev=self.t_rte_prev)
        dist.all_gather_into_tensor(self.ag_recv_all.view(-1), self.e_own_all.view(-1))
        self._keep_csB(store)
        self._side_update(self.users, self.nU, self.eT, self.eB, self.eT_next, self.part_u, self.Gamma_shp,
                          self.k_rte_prev, self.Theta, self.k_rte, self.csB, self.csT_part, self.gsu, self.gu,
                          hy.a, hy.k_shp, hy.add_k_rte, store)
        ops.colsum_reduce(self.csT_part, self.csT, ld)
        dist.all_reduce(self.csT)
        ops.item_apply_rows(self.ag_recv_all, self.shp_own_all, self.eB, self.Lambda_shp if store else None,
                            self.Beta if store else None, self.t_rte, self.csT, self.csB_part_sc, hy.add_t_rte, k, ld,
                            self.rank, self.world, self.nI, self._range_rows)
        ops.colsum_reduce(self.csB_part_sc, self.csB, ld)
        dist.all_reduce(self.csB)
        self._sc_fresh = True          # (in order on one stream: nothing stays in flight)
        self._tables_split = True
        self.eT, self.eT_next = self.eT_next, self.eT
        self.niter_done += 1

    # ---- exchange-stream plumbing (CPU tensors / no GPU: everything degenerates to plain in-order calls) ------------------
    def _xstream(self):
        if self.device.type != "cuda":
            return None
        if getattr(self, "_xs", None) is None:
            # HIGH priority: the exchange chain is latency-critical, and a priority stream is served by another hardware
            # queue than the normal-priority compute stream (ROCm multiplexes the streams of one priority over 4 hardware
            # queues in creation order: on the compute stream's queue a waiting exchange held back the sweeps behind it)
            self._xs = _side_stream(self.device, "exchange", -1)
        return self._xs

    def _event(self):
        """Events are re-used round-robin (a re-recorded event is only ever waited for after its latest record)."""
        pool = getattr(self, "_ev_pool", None)
        if pool is None:
            pool = self._ev_pool = [torch.cuda.Event() for _ in range(32)]
            self._ev_next = -1
        self._ev_next = (self._ev_next + 1) % 32
        return pool[self._ev_next]

    def _mark(self, xs):
        """Event at the current end of stream `xs` (None without one)."""
        if xs is None:
            return None
        ev = self._event()
        ev.record(xs)
        return ev

    def _wait(self, ev):
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)

    def _sync_scatter_streams(self):
        """The current stream waits for the exchanges still in flight on the side streams."""
        views = self._chunk_views or []
        if views and not getattr(self, "_sc_fresh", True):
            if self._last_native:
                self._plan.join(torch.cuda.current_stream(self.device).cuda_stream)
            else:
                for c in views:
                    self._wait(c["ag_done"])
            self._sc_fresh = True        # the next iteration re-synchronises its side streams with this one

    def _sync_scatter(self):
        """Wait for the outstanding exchanges and gather the per-owner item tables."""
        self._sync_scatter_streams()
        if self._tables_split:
            if self._plan is not None and self._last_native:
                self._plan.status()          # (direct exchange: a wait that ran out raises here, not garbage later)
            for c in self._scatter_views():
                for tab in (self.Lambda_shp, self.Beta, self.t_rte, self.t_rte_prev):
                    self.dist.all_gather_into_tensor(tab[c["lo"]: c["hi"]], tab[c["o0"]: c["o1"]].clone())
            self._tables_split = False

    def release_exchange(self):
        """End of a sharded fit (every rank calls it): wait for the exchanges, meet the other ranks on the host, THEN free
        the peer-mapped region and the plan -- no peer can still have a pull of this rank's memory queued.  The model can
        go on iterating afterwards (the exchange is set up again on first use)."""
        if not self.dist or self._chunk_views is None:
            return
        self._sync_scatter()
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        if self.world > 1 and hasattr(self.dist, "barrier"):
            self.dist.barrier()
        if self._plan is not None:
            self._plan.close()
        self._plan = None
        if self._region is not None:
            self.acc_i = self.e_own_all = None
            for c in self._chunk_views:
                c.pop("acc", None), c.pop("e_own", None)
            self._region.close()
            self._region = None
        self._chunk_views = None
        self._last_native, self._sc_fresh = False, True

    def flush_items(self, store=True):
        """Sharded path: make the item tables current on this rank (wait + gather)."""
        if self.dist:
            self._sync_scatter()
