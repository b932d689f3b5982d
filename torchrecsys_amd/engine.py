# -*- coding: utf-8 -*-
"""Training-step engine behind TorchRecSys.fit(): drives the fused HIP kernels for one optimiser step and honours the
user-supplied torch optimiser object (reference model.py:188-200 only ever calls zero_grad() and step() on it).

Optimiser dispatch:
  torch.optim.SGD (momentum=0, weight_decay=0)  -> fused row scatter  W[r] += -lr * g        (trs_score_sgd_update)
  torch.optim.SparseAdam                        -> accumulate + elected-owner lazy Adam rows  (trs_rows_apply_sparse_adam)
  torch.optim.Adagrad (weight_decay=0)          -> accumulate + elected-owner Adagrad rows    (trs_rows_apply_adagrad)
  anything else                                 -> the staged rows are handed to the optimiser as the sparse COO
                                                   gradients autograd would have produced, then optimizer.step().
Moments live in optimizer.state[p] under torch's own key names, so optimizer.state_dict() stays truthful.
"""
import os

import torch
from ctypes import addressof as C_addressof

from . import ops


def _group_of(optimizer, p):
    for g in optimizer.param_groups:
        for q in g["params"]:
            if q is p:
                return g
    return None


def classify_optimizer(optimizer, params):
    """'sgd' | 'sparse_adam' | 'adagrad' | 'generic' for the given embedding parameters."""
    groups = [_group_of(optimizer, p) for p in params]
    if any(g is None for g in groups):
        return "generic"
    t = type(optimizer)
    if t is torch.optim.SGD:
        ok = all(g["momentum"] == 0 and g["dampening"] == 0 and g["weight_decay"] == 0 and not g["nesterov"]
                 and not g.get("maximize", False) for g in groups)
        return "sgd" if ok else "generic"
    if t is torch.optim.SparseAdam:
        return "sparse_adam" if all(not g.get("maximize", False) for g in groups) else "generic"
    if t is torch.optim.Adam:
        # torch.optim.Adam raises on sparse gradients, so the reference cannot run it on these tables at all (SURVEY
        # §0.3, README quick-start).  Here it gets SparseAdam's lazy semantics (touched rows only) with Adam's own
        # hyper-parameters; moments live in optimizer.state[p] under Adam's key names.
        ok = all(g["weight_decay"] == 0 and not g.get("amsgrad", False) and not g.get("maximize", False) for g in groups)
        return "sparse_adam" if ok else "generic"
    if t is torch.optim.Adagrad:
        ok = all(g["weight_decay"] == 0 and not g.get("maximize", False) for g in groups)
        return "adagrad" if ok else "generic"
    return "generic"


class RowState:
    """Per-table scratch of the coalescing optimisers: gradient accumulator + owner-election stamps."""

    def __init__(self, p):
        self.acc = torch.zeros_like(p.data)
        self.stamp = torch.zeros(p.shape[0], dtype=torch.int32, device=p.device)
        self.step_id = 0

    def next_id(self):
        self.step_id += 1
        if self.step_id >= 2 ** 31 - 1:
            self.stamp.zero_()
            self.step_id = 1
        return self.step_id


def apply_rows(kind, optimizer, p, rs, idx, vals, ld=None):
    """One coalescing-optimiser update of table `p` from uncoalesced (idx, vals) entries."""
    g = _group_of(optimizer, p)
    st = optimizer.state[p]
    ops.rows_scatter_add(rs.acc, idx, vals, 1.0, ld=ld)
    if kind == "sparse_adam":
        if len(st) == 0:
            st["step"] = torch.tensor(0.0) if type(optimizer) is torch.optim.Adam else 0
            st["exp_avg"] = torch.zeros_like(p.data)
            st["exp_avg_sq"] = torch.zeros_like(p.data)
        st["step"] += 1
        b1, b2 = g["betas"]
        ops.rows_apply_sparse_adam(p.data, rs.acc, st["exp_avg"], st["exp_avg_sq"], rs.stamp, idx, rs.next_id(),
                                   g["lr"], b1, b2, g["eps"], int(st["step"]))
    elif kind == "adagrad":
        if "sum" not in st:  # torch.optim.Adagrad creates its state in __init__; be safe for foreign objects
            st["step"] = torch.tensor(0.0)
            st["sum"] = torch.full_like(p.data, g.get("initial_accumulator_value", 0.0))
        st["step"] += 1
        step = float(st["step"])
        clr = g["lr"] / (1 + (step - 1) * g["lr_decay"])
        ops.rows_apply_adagrad(p.data, rs.acc, st["sum"], rs.stamp, idx, rs.next_id(), clr, g["eps"])
    else:
        raise ValueError(kind)


class SparseScorerTrainer:
    """One training step of a Linear / FM scorer on a device-resident batch."""

    def __init__(self, net, optimizer, batch_capacity):
        self.net, self.opt = net, optimizer
        self.params = net.table_params()
        self.kind = classify_optimizer(optimizer, self.params)
        dev = self.params[0].device
        self.dev = dev
        self.D = self.params[0].shape[1]
        self.M = net.n_meta_tables()
        self.R = 3 + 2 * self.M
        self.cap = batch_capacity
        self.grad_rows = torch.empty((self.R, batch_capacity, self.D), dtype=torch.float32, device=dev)
        self.grad_lin = torch.empty((self.R, batch_capacity), dtype=torch.float32, device=dev)
        self.err = torch.zeros(1, dtype=torch.int32, device=dev)
        self.kernel_events = None  # bench.py: {"kernel name": [(start_event, end_event), ...]} on the launch stream
        self.loss_id = 0  # _lib.LOSS_ID: hinge (the reference) | bpr; set by fit(loss=...)
        # specialised exact 3-kernel SGD step (csrc/fast_step.hip): no metadata, plain SGD with one learning rate
        self.fast_lr = None
        self.fast_kind = None  # "sgd": C step loop on every path; "sparse_adam" / "adagrad": on the presorted path only
        hyper = {"sgd": ("lr",), "sparse_adam": ("lr", "betas", "eps"),
                 "adagrad": ("lr", "lr_decay", "eps", "initial_accumulator_value")}.get(self.kind)
        # metadata scorers: plain SGD on any shape; the adaptive rules where the staging kernel with the rule exists
        # (csrc/fast_step.hip meta_stage_kernel: up to 3 sorted columns, whole-row D)
        meta_adaptive_ok = (self.M <= 3 and self.D in (32, 64, 128, 256) and os.environ.get("TRS_META_SORTED", "1") != "0")
        if hyper is not None and (self.M == 0 or self.kind == "sgd" or meta_adaptive_ok):
            lrs = {tuple(_group_of(optimizer, p).get(k) for k in hyper) for p in self.params}
            if len(lrs) == 1:  # one rule for all tables
                self.fast_kind = self.kind
                self.fast_lr = _group_of(optimizer, self.params[0])["lr"]
                self.gz = torch.empty((2, batch_capacity), dtype=torch.float32, device=dev)
                self.du = torch.empty((batch_capacity, self.D), dtype=torch.float32, device=dev)
                self.id_bufs = [torch.empty(batch_capacity, dtype=torch.int32, device=dev) for _ in range(3)]
                self.scratch = ops.train_scratch(self.params[0].shape[0], self.params[1].shape[0], batch_capacity, self.D, dev)
                self.stamp = 1
        self.row_state = {}
        if self.kind in ("sparse_adam", "adagrad"):
            self.row_state = {id(p): RowState(p) for p in self.params}
        if self.fast_kind in ("sparse_adam", "adagrad"):
            # item runs cut at a chunk boundary: at most one per 64-reference chunk (+ slack)
            # (one list + counter pair for the item table and one per metadata column)
            self.cut_rows = torch.empty((1 + self.M, 2 * batch_capacity // 64 + 64), dtype=torch.int32, device=dev)
            self.cut_count = torch.zeros((1 + self.M, 2), dtype=torch.int32, device=dev)
            self.meta_lin_state = None  # Linear scorer: scratch in place of the 1-wide metadata tables' state
        # presorted epoch slices (csrc/presort.hip): two buffer sets, the one being built on the side stream and the one
        # the step kernels read
        self.ustage = None            # (capacity, D) pre-update user rows staged by K1 for the item update
        self._ps_fits = {}            # batch size -> do the two buffer sets fit
        self._ps_sets = [None, None]  # ops.EpochPresort per set
        self._ps_tags = [None, None]  # what each set holds (epoch, first batch, batches, batch size)
        self._ps_done = [None, None]  # (slice view, completion event) of a prefetched set
        self._ps_cur = 1              # set the step kernels currently read
        self._ps_stream = None        # side HIP stream of the prefetch
        self.xstage = self.meta_ids = None  # metadata scorers: staged field sums / user rows, metadata ids of a batch
        self.item_meta = None               # (n_items, M) int32 item -> metadata table (set by the runner)

    def _views(self, B):
        """Staging views for a batch of B <= capacity rows (contiguous (R,B,D) / (R,B) prefixes)."""
        if B == self.cap:
            return self.grad_rows, self.grad_lin
        gr = self.grad_rows.view(-1)[: self.R * B * self.D].view(self.R, B, self.D)
        gl = self.grad_lin.view(-1)[: self.R * B].view(self.R, B)
        return gr, gl

    def _sync(self):
        """(device arrival counter, host count of scheduled arrivals) of the one-launch flag-mode step
        (trs_train_args.sync_dev / sync_count_host), or None: TRS_FLAG_ONE_LAUNCH=0."""
        if os.environ.get("TRS_FLAG_ONE_LAUNCH", "1") == "0" or getattr(self, "_one_launch_off", False):
            return None
        if getattr(self, "sync", None) is None:
            import ctypes
            self.sync = (torch.zeros(288, dtype=torch.int32, device=self.dev), ctypes.c_uint32(0))
        return self.sync

    def _stamps(self, n):
        """First of n consecutive step stamps of the duplicate-detection scratch (never 0, re-zeroed before wrapping)."""
        if self.stamp + n >= 0xFFFFFFF0:
            self.scratch.zero_()
            self.stamp = 1
            if self.fast_kind in ("sparse_adam", "adagrad"):
                self.cut_count.zero_()  # the cut-run counters alternate by stamp parity
        first = self.stamp
        self.stamp += n
        return first

    # bench timing samples one step in 29: a sampled step carries four event records and runs ~20 us longer (rocprofv3
    # trace: 9 us between K1 and K2, 4-6 us before / after the step, against back-to-back launches otherwise).  A prime
    # stride: a power of two would pin the samples to the same positions of the 512-batch presort slices (the first
    # steps of a slice overlap the next slice's sort) and of the 64-step C calls
    EVENT_EVERY = 29

    def _make_events(self, n_steps):
        """Raw hipEvent_t handles, 4 per sampled step (None for the others), owned by an ops.TimingEvents."""
        self._ev_count = getattr(self, "_ev_count", 0)
        sampled = [s for s in range(n_steps) if (self._ev_count + s) % self.EVENT_EVERY == 0]
        self._ev_count += n_steps
        te = ops.TimingEvents(4 * len(sampled))
        evs = [None] * (4 * n_steps)
        for j, s in enumerate(sampled):
            evs[4 * s:4 * s + 4] = [te.handles[4 * j + q] for q in range(4)]
        return te, evs, len(sampled)

    def _collect_events(self, te, n_sampled, names=("fwd_stage_kernel", "item_update_kernel", "user_update_kernel")):
        """names: what ran between the four events of a step (item_update_kernel = phases a + b when not presorted)."""
        ke = self.kernel_events
        for j in range(n_sampled):
            for q, name in enumerate(names):
                ke.setdefault(name, []).append((te, 4 * j + q, 4 * j + q + 1))

    def fast_stream_steps(self, st, shuffle_key, sample_seed, first_pos, batch, n_steps, loss_sums):
        """n_steps fused steps straight from the resident stream `st` (dict user/pos/neg int32); loss_sums: (n_steps,)
        view.  Only valid when self.fast_lr is not None and batch == capacity."""
        te, evs, ns = self._make_events(n_steps) if self.kernel_events is not None else (None, None, 0)
        if "ui" not in st:
            st["ui"] = ops.interleave_stream(st["user"], st["pos"])
        ops.train_steps_sgd(self.net.NET, self.net.tables(), st["ui"], st["neg"], shuffle_key,
                            sample_seed, first_pos, batch, n_steps, self.fast_lr, *self.id_bufs, self.gz, self.du,
                            loss_sums, self.err, self.scratch, self._stamps(n_steps), evs, loss=self.loss_id)
        if te is not None:
            self._collect_events(te, ns)

    # ---- presorted item references (csrc/presort.hip) ---------------------------------------------------------------
    SLICE_BATCHES = int(os.environ.get("TRS_SLICE_BATCHES", "512"))  # batches grouped per presort call (bounds the buffers: 2 x 512 x B references per set)

    def wants_presort(self, batch):
        """Group the item references by row once per epoch slice (removes the float atomics from the item update and
        lets the step run as two launches).  Measured faster in both regimes — dense (c2: 131 072 references over 100 K
        items per step, 67 -> 49 us) and sparse (c4 shard: 65 536 over 1 M items, 68 -> 51 us) — so it is on whenever
        its buffers fit; TRS_PRESORT_MIN_DENSITY (references per step / n_items) is a tuning knob."""
        n_items = self.params[1].shape[0]
        dense = float(os.environ.get("TRS_PRESORT_MIN_DENSITY", "0"))
        if self.fast_lr is None or 2 * batch < dense * n_items:
            return False
        memo = self._ps_fits
        if batch not in memo:  # decided once: two buffer sets (one being sorted while the other is read)
            need = 2 * (ops.EpochFlags.bytes_needed(self.SLICE_BATCHES, batch) if self.sparse_regime(batch) else
                        ops.EpochPresort.bytes_needed(self.SLICE_BATCHES, batch, n_items, self.M))
            free, _ = torch.cuda.mem_get_info(self.dev)
            memo[batch] = need < 0.3 * free
        return memo[batch]

    def sparse_regime(self, batch):
        """Most item references of a batch are ALONE on their row (uniform ids: probability exp(-2B / n_items) — 0.94 at
        c4's per-GPU batch, 65 536 references over 1M items; 0.27 at c2, 131 072 over 100K).  Then K1 updates the lone
        rows in place, the few flagged references follow in a small launch of float atomics, and the epoch needs duplicate
        FLAGS only (trs_epoch_flags: one launch, no sort) — c4: 48.5 us per step with sorted runs for every reference ->
        38.6 with K1 taking the lone references -> see DESIGN.md for the flag mode.  In the dense regime the sorted runs
        stay (K1's flag loads and conditional stores cost more than they save: 45.7 -> 47.1 us at c2).
        TRS_SPARSE_REGIME=0/1 forces the choice; plain SGD without metadata only."""
        if self.fast_kind != "sgd" or self.M > 0:
            return False
        env = os.environ.get("TRS_SPARSE_REGIME")
        if env is not None:
            return env != "0"
        return 2 * batch <= 0.7 * self.params[1].shape[0] and batch <= ops.EpochFlags.MAX_BATCH  # lone share >= 1/2

    def _presort_run(self, i, n_batches, batch, st, shuffle_key, sample_seed, first_pos, given_ids):
        sets = self._ps_sets
        ps = sets[i]
        if (ps is None or ps.batch != batch or ps.n_batches < n_batches) and self.sparse_regime(batch):
            ps = sets[i] = ops.EpochFlags(max(n_batches, min(self.SLICE_BATCHES, n_batches)), batch,
                                          self.params[0].shape[0], self.params[1].shape[0], self.dev,
                                          ordered=os.environ.get("TRS_FLAG_ORDERED", "1") != "0")  # knob: A/B
        elif ps is None or ps.batch != batch or ps.n_batches < n_batches:
            meta_kw = {}
            if self.M > 0 and os.environ.get("TRS_META_SORTED", "1") != "0":  # knob: 0 = atomic scatter of staged fields
                meta_kw = dict(item_meta=self.item_meta, n_meta=[p.shape[0] for p in self.params[4:4 + self.M]])
            # plain SGD without metadata needs the users' duplicate FLAGS only (flagged users add their gradient with float
            # atomics) as long as few users repeat inside a batch — share ~ 1 - exp(-batch / n_users): c2 6 % (flags 43.5 us
            # per step, sorted runs 45.9), c1 29 % (flags 17.0, sorted runs 13.8); the adaptive rules and the metadata
            # scorers coalesce duplicated users through sorted runs
            few_dups = batch <= 0.15 * self.params[0].shape[0]
            user_sort = not (self.fast_kind == "sgd" and self.M == 0 and few_dups)
            if os.environ.get("TRS_USER_SORT") in ("0", "1"):  # tuning knob
                user_sort = os.environ["TRS_USER_SORT"] == "1" or not (self.fast_kind == "sgd" and self.M == 0)
            ps = sets[i] = ops.EpochPresort(max(n_batches, min(self.SLICE_BATCHES, n_batches)), batch,
                                            self.params[0].shape[0], self.params[1].shape[0], self.dev,
                                            user_sort=user_sort,
                                            item_flags=os.environ.get("TRS_ITEM_INLINE", "0") == "1", **meta_kw)
        if ps.n_batches != n_batches:  # a shorter tail slice: same buffers, fewer batches
            full = ps
            ps = type(full).__new__(type(full))
            ps.__dict__.update(full.__dict__)
            ps.n_batches = n_batches
        if st is not None:
            ps.run(st["ui"], st["neg"], shuffle_key, sample_seed, first_pos, self.err, sampler=getattr(self, "sampler", None))
        else:
            ps.run(None, None, 0, 0, 0, self.err, given_ids=given_ids)
        return ps

    def presort_slice(self, n_batches, batch, st=None, shuffle_key=0, sample_seed=0, first_pos=0, given_ids=None,
                      tag=None, prefetch=False):
        """Group the item references of `n_batches` whole batches (device stream `st`, or host-prepared ids).

        Two buffer sets: with prefetch=True the work is queued on a side HIP stream into the set the step kernels are
        not reading (the sort is bandwidth work, the step kernels are latency-bound, so the two overlap) and is picked
        up later by a call with the same `tag`."""
        if st is not None and "ui" not in st:
            st["ui"] = ops.interleave_stream(st["user"], st["pos"])
        tags, done, cur = self._ps_tags, self._ps_done, self._ps_cur
        main = torch.cuda.current_stream(self.dev)
        if prefetch:
            i = 1 - cur
            if tag is not None and tags[i] == tag:
                return None
            if self._ps_stream is None:
                self._ps_stream = torch.cuda.Stream(self.dev)
            side = self._ps_stream
            side.wait_stream(main)  # the steps that read set i were queued before this point
            with torch.cuda.stream(side):
                ps = self._presort_run(i, n_batches, batch, st, shuffle_key, sample_seed, first_pos, given_ids)
                ev = torch.cuda.Event()
                ev.record(side)
            tags[i], done[i] = tag, (ps, ev)
            return None
        for i in (0, 1):
            if tag is not None and tags[i] == tag and done[i] is not None:
                ps, ev = done[i]
                main.wait_event(ev)
                self._ps_cur = i
                return ps
        i = 1 - cur
        if self._ps_stream is not None:
            main.wait_stream(self._ps_stream)  # a prefetch may still be writing set i
        ps = self._presort_run(i, n_batches, batch, st, shuffle_key, sample_seed, first_pos, given_ids)
        tags[i], done[i] = tag, None
        self._ps_cur = i
        return ps

    def _meta_stage(self, batch, item_meta, meta_sorted=(), meta_ids=(None, None)):
        """_lib.TrsMetaStage of a metadata scorer: K1 looks the ids up in `item_meta` (n_items, M) int32."""
        from . import _lib
        if item_meta is None or item_meta.dtype != torch.int32:
            raise ValueError("metadata scorers need the (n_items, M) int32 item -> metadata table on the device")
        if self.xstage is None:
            passes = 2 if self.net.NET == "fm" else 1
            self.xstage = torch.empty((passes, self.cap, self.D), dtype=torch.float32, device=self.dev)
            self.meta_ids = torch.empty((2, self.cap, self.M), dtype=torch.int32, device=self.dev)
        if batch != self.cap:
            raise ValueError("the C step loop runs whole batches of the trainer's capacity")
        ms = _lib.TrsMetaStage()
        ms.item_meta_tab, ms.xstage, ms.meta_ids = ops.ptr(item_meta), ops.ptr(self.xstage), ops.ptr(self.meta_ids)
        ms.grad_rows, ms.grad_lin = ops.ptr(self.grad_rows), ops.ptr(self.grad_lin)
        for m, (k, v) in enumerate(meta_sorted):  # sorted references per column: runs instead of the atomic scatter
            ms.sorted_keys[m], ms.sorted_vals[m] = k, v
        if meta_sorted:
            if getattr(self, "lin_scratch", None) is None:
                self.lin_scratch = torch.zeros(max(p.shape[0] for p in self.params[4:4 + self.M]), dtype=torch.float32,
                                               device=self.dev)
            ms.lin_scratch = ops.ptr(self.lin_scratch)
            ms.pos_meta_ids, ms.neg_meta_ids = ops.ptr(meta_ids[0]), ops.ptr(meta_ids[1])
        return ms

    def fast_sorted_steps(self, ps, b_in_slice, batch, n_steps, loss_sums, item_meta=None):
        te, evs, ns = self._make_events(n_steps) if self.kernel_events is not None else (None, None, 0)
        if self.ustage is None:
            self.ustage = torch.empty_like(self.du)  # pre-update user rows staged by K1 for the item update
        if isinstance(ps, ops.EpochFlags) and te is None:  # the same call with its argument struct kept between calls
            T = self.net.tables()
            sig = (C_addressof(T), float(self.fast_lr), int(self.loss_id), ops.ptr(self.gz), ops.ptr(self.du),
                   ops.ptr(self.ustage), ops.ptr(self.scratch))
            fc = getattr(self, "_flag_call", None)
            if fc is None or fc.batch != batch or fc.sig != sig:
                fc = self._flag_call = ops.FlagStepCall(self.net.NET, T, batch, self.fast_lr, self.gz, self.du, self.err,
                                                        self.scratch, self.ustage, self.loss_id, self._sync())
            ops.stamp("fast_sorted_steps:before_call")
            fc(ps, b_in_slice, n_steps, loss_sums, self._stamps(n_steps))
            return
        if isinstance(ps, ops.EpochFlags):  # sparse regime: flags only, the flagged references follow K1 with atomics
            ids, udup, idup = ps.step_args(b_in_slice)
            sy = self._sync()
            arrivals = sy[1].value if sy else 0
            ops.train_steps_sgd(self.net.NET, self.net.tables(), None, None, 0, 0, 0, batch, n_steps, self.fast_lr, *ids,
                                self.gz, self.du, loss_sums, self.err, self.scratch, self._stamps(n_steps), evs,
                                user_dup=udup, item_dup=idup, ustage=self.ustage, loss=self.loss_id, sync=self._sync(),
                                n_flagged=ps.n_flagged_from(b_in_slice))
            if te is not None:
                # one launch per step (K1's own workgroups applied the flagged references: the library scheduled arrivals
                # on the counter): the second interval holds no kernel either
                one = sy is not None and sy[1].value != arrivals
                self.one_launch = one
                self._collect_events(te, ns, ("fwd_stage_kernel", "event_overhead_2" if one else "flagged_update_kernel",
                                              "event_overhead"))
            return
        ids, sk, sv, udup, usorted, idup = ps.step_args(b_in_slice)
        idup = None  # dense regime / adaptive rules / metadata scorers: every item reference goes through the runs
        if os.environ.get("TRS_ITEM_INLINE", "0") == "1" and self.fast_kind == "sgd" and self.M == 0:
            idup = ps.step_args(b_in_slice)[5]  # knob: sorted runs + K1 taking the lone item references (INL 2)
        opt = self._adaptive_rule(n_steps) if self.fast_kind != "sgd" else None
        meta = (self._meta_stage(batch, item_meta, ps.meta_step_args(b_in_slice), ps.meta_id_args(b_in_slice))
                if self.M > 0 else None)
        ops.train_steps_sgd(self.net.NET, self.net.tables(), None, None, 0, 0, 0, batch, n_steps, self.fast_lr, *ids,
                            self.gz, self.du, loss_sums, self.err, self.scratch, self._stamps(n_steps), evs, sk, sv,
                            ps.key_bytes, udup, self.ustage, usorted, opt, meta, idup, loss=self.loss_id)
        if te is not None:
            # 32-bit keys: item and duplicated-user updates are ONE launch, and the last two events are recorded back to
            # back — that interval is the cost of an event record itself
            fused = ps.key_bytes == 4 and ps.ukey_bytes in (0, 4)
            self._collect_events(te, ns, ("fwd_stage_kernel", "sorted_updates_fused_kernel", "event_overhead") if fused
                                 else ("fwd_stage_kernel", "sorted_item_update_kernel", "sorted_user_dup_update_kernel"))

    def _adaptive_rule(self, n_steps):
        """_lib.TrsOpt of the next n_steps SparseAdam / Adagrad steps; moments live in optimizer.state[p] under torch's
        own key names (created on first use like torch does) and the step counters are advanced here."""
        from . import _lib
        g = _group_of(self.opt, self.params[0])
        sts = []
        for p in self.params:
            st = self.opt.state[p]
            if self.kind == "sparse_adam" and "exp_avg" not in st:
                st["step"] = torch.tensor(0.0) if type(self.opt) is torch.optim.Adam else 0
                st["exp_avg"] = torch.zeros_like(p.data)
                st["exp_avg_sq"] = torch.zeros_like(p.data)
            if self.kind == "adagrad" and "sum" not in st:
                st["step"] = torch.tensor(0.0)
                st["sum"] = torch.full_like(p.data, g.get("initial_accumulator_value", 0.0))
            sts.append(st)
        step0 = int(sts[0]["step"])
        assert all(int(st["step"]) == step0 for st in sts), "embedding tables were stepped a different number of times"
        o = _lib.TrsOpt()
        o.lr, o.step0 = float(g["lr"]), step0
        if self.kind == "sparse_adam":
            o.kind, (o.beta1, o.beta2), o.eps = 1, g["betas"], float(g["eps"])
            s1, s2 = [st["exp_avg"] for st in sts], [st["exp_avg_sq"] for st in sts]
        else:
            o.kind, o.eps, o.lr_decay = 2, float(g["eps"]), float(g["lr_decay"])
            s1, s2 = [st["sum"] for st in sts], [None] * len(sts)
        o.user_s1, o.item_s1, o.user_lin_s1, o.item_lin_s1 = (ops.ptr(t) for t in s1[:4])
        o.user_s2, o.item_s2, o.user_lin_s2, o.item_lin_s2 = (ops.ptr(t) for t in s2[:4])
        # the generic path's (all-zero between steps) gradient accumulators double as the meeting point of cut runs
        o.gacc, o.gacc_lin = ops.ptr(self.row_state[id(self.params[1])].acc), ops.ptr(self.row_state[id(self.params[3])].acc)
        o.cut_rows, o.cut_count, o.cut_capacity = ops.ptr(self.cut_rows[0]), ops.ptr(self.cut_count[0]), self.cut_rows.shape[1]
        M = self.M
        has_lin = len(self.params) == 4 + 2 * M  # FM: 1-wide metadata tables follow the metadata tables
        if M > 0 and not has_lin and self.meta_lin_state is None:
            self.meta_lin_state = [torch.zeros((3, self.params[4 + m].shape[0]), dtype=torch.float32, device=self.dev)
                                   for m in range(M)]
        for m in range(M):
            o.meta_s1[m], o.meta_s2[m] = ops.ptr(s1[4 + m]), ops.ptr(s2[4 + m])
            o.meta_gacc[m] = ops.ptr(self.row_state[id(self.params[4 + m])].acc)
            if has_lin:
                o.meta_lin_s1[m], o.meta_lin_s2[m] = ops.ptr(s1[4 + M + m]), ops.ptr(s2[4 + M + m])
                o.meta_gacc_lin[m] = ops.ptr(self.row_state[id(self.params[4 + M + m])].acc)
            else:
                sc = self.meta_lin_state[m]
                o.meta_lin_s1[m], o.meta_lin_s2[m], o.meta_gacc_lin[m] = ops.ptr(sc[0]), ops.ptr(sc[1]), ops.ptr(sc[2])
            o.meta_cut_rows[m], o.meta_cut_count[m] = ops.ptr(self.cut_rows[1 + m]), ops.ptr(self.cut_count[1 + m])
        for st in sts:
            st["step"] += n_steps
        return o

    def fast_array_steps(self, ep, first, batch, n_steps, loss_sums):
        """n_steps fused steps over consecutive batches of host-prepared epoch id arrays `ep` (dict user/pos/neg int32,
        epoch order) starting at row `first`."""
        te, evs, ns = self._make_events(n_steps) if self.kernel_events is not None else (None, None, 0)
        e = first + n_steps * batch
        ops.train_steps_sgd(self.net.NET, self.net.tables(), None, None, 0, 0, 0, batch, n_steps, self.fast_lr,
                            ep["user"][first:e], ep["pos"][first:e], ep["neg"][first:e], self.gz, self.du, loss_sums,
                            self.err, self.scratch, self._stamps(n_steps), evs, loss=self.loss_id)
        if te is not None:
            self._collect_events(te, ns)

    def step(self, ids, loss_slot, auc_slot=None):
        """ids: dict user/pos/neg[/pos_meta/neg_meta] of GPU id tensors.  loss_slot: 1-element fp32 view that receives
        the SUM of the batch's hinge terms (caller divides by B)."""
        B = ids["user"].shape[0]
        net = self.net
        if self.fast_kind == "sgd" and self.M == 0 and auc_slot is None and ids["user"].dtype == torch.int32:
            te, evs, ns = self._make_events(1) if self.kernel_events is not None else (None, None, 0)
            ops.train_steps_sgd(net.NET, net.tables(), None, None, 0, 0, 0, B, 1, self.fast_lr, ids["user"],
                                ids["pos"], ids["neg"], self.gz, self.du, loss_slot, self.err, self.scratch,
                                self._stamps(1), evs, loss=self.loss_id)
            if te is not None:
                self._collect_events(te, ns)
            return
        T = net.tables()
        Bt, keep = ops.make_batch(ids["user"], ids["pos"], ids["neg"], ids.get("pos_meta"), ids.get("neg_meta"),
                                  self.err)
        gr, gl = self._views(B)
        ev = self.kernel_events
        if ev is not None:
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record()
        ops.score_fwd_bwd(net.NET, T, Bt, B, self.D, self.M, self.dev, loss_slot, auc_slot, want_scores=False,
                          grad_rows=gr, grad_lin=gl, loss=self.loss_id)
        if ev is not None:
            e1.record()
        if self.kind == "sgd":
            groups = [_group_of(self.opt, p) for p in self.params]
            lrs = {g["lr"] for g in groups}
            if len(lrs) == 1:
                ops.score_sgd_update(net.NET, T, Bt, gr, gl, lrs.pop())
                if ev is not None:
                    e2.record()
                    ev.setdefault("score_kernel<fwd_bwd>", []).append((e0, e1))
                    ev.setdefault("score_sgd_update_kernel", []).append((e1, e2))
            else:
                self._per_table(ids, gr, gl, lambda p, idx, vals, ld: ops.rows_scatter_add(
                    p.data, idx, vals, -_group_of(self.opt, p)["lr"], ld=ld))
        elif self.kind in ("sparse_adam", "adagrad"):
            self._per_table(ids, gr, gl, lambda p, idx, vals, ld: apply_rows(
                self.kind, self.opt, p, self.row_state[id(p)], idx, vals, ld))
        else:
            self.opt.zero_grad()
            self._per_table(ids, gr, gl, self._set_sparse_grad)
            self.opt.step()

    @staticmethod
    def _set_sparse_grad(p, idx, vals, ld):
        g = torch.sparse_coo_tensor(idx.reshape(1, -1).long(), vals, size=p.shape)
        p.grad = g if p.grad is None else p.grad + g

    def _per_table(self, ids, gr, gl, fn):
        """Call fn(param, idx (n,), vals (n, width), ld) once per embedding table with that table's COO entries."""
        B, D, M = ids["user"].shape[0], self.D, self.M
        ps = self.params
        item_idx = torch.cat([ids["pos"], ids["neg"]])
        fn(ps[0], ids["user"], gr[0], D)
        fn(ps[1], item_idx, gr[1:3].reshape(2 * B, D), D)
        fn(ps[2], ids["user"], gl[0].reshape(B, 1), 1)
        fn(ps[3], item_idx, gl[1:3].reshape(2 * B, 1), 1)
        for m in range(M):
            midx = torch.cat([ids["pos_meta"][:, m], ids["neg_meta"][:, m]]).contiguous()
            sl = slice(3 + 2 * m, 5 + 2 * m)
            fn(ps[4 + m], midx, gr[sl].reshape(2 * B, D), D)
            if net_has_meta_lin(self.net):
                fn(ps[4 + M + m], midx, gl[sl].reshape(2 * B, 1), 1)

    def check_errors(self):
        from .collaborative._scorer import check_err_flag
        try:
            check_err_flag(self.err, "fit")
        except RuntimeError:
            self._one_launch_off = True  # (the grid was not resident at once: two launches per step from now on)
            self._flag_call = None
            raise


def net_has_meta_lin(net):
    return bool(getattr(net, "META_LIN_NAME", None))
