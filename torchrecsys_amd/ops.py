# -*- coding: utf-8 -*-
"""Thin torch-tensor front of the C-ABI (include/trs.h).  Every function enqueues HIP kernels of libtrs_hip.so on
torch's current stream; torch only owns the memory.  No CPU path: tensors must live on the GPU."""
import ctypes as C

import torch

from . import _lib
from ._lib import TRS_MAX_META, TRS_NET_FM, TRS_NET_LINEAR, TrsBatch, TrsTables, check, ptr

NET_ID = {"linear": TRS_NET_LINEAR, "fm": TRS_NET_FM}


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """hipStream_t of torch's current stream on the current device (the raw getters: this sits on every kernel launch,
    and torch.cuda.current_stream() costs ~9 us of Python per call)."""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def _dev(t, name, dtype=None):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a GPU tensor: torchrecsys_amd computes on the MI355X only "
                           f"(no CPU fallback); got device {t.device}")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return t


def make_tables(user, item, user_lin=None, item_lin=None, metas=(), meta_lins=()):
    """Build a trs_tables struct from fp32 GPU tensors.  Returns (struct, keepalive list)."""
    T = TrsTables()
    keep = [user, item, user_lin, item_lin, list(metas), list(meta_lins)]
    _dev(user, "user table", torch.float32)
    _dev(item, "item table", torch.float32)
    if user.dim() != 2 or item.dim() != 2 or user.shape[1] != item.shape[1]:
        raise ValueError("user/item tables must be (n, D) with the same D")
    M = len(metas)
    if M > TRS_MAX_META:
        raise ValueError(f"at most {TRS_MAX_META} metadata columns are supported, got {M}")
    T.user, T.item = ptr(user), ptr(item)
    T.user_lin = ptr(_dev(user_lin, "user 1-wide table", torch.float32))
    T.item_lin = ptr(_dev(item_lin, "item 1-wide table", torch.float32))
    T.n_users, T.n_items = user.shape[0], item.shape[0]
    T.D, T.M = user.shape[1], M
    for m in range(M):
        _dev(metas[m], f"metadata table {m}", torch.float32)
        if metas[m].shape[1] != user.shape[1]:
            raise ValueError("metadata tables must have the same D as user/item")
        T.meta[m] = ptr(metas[m])
        T.n_meta[m] = metas[m].shape[0]
        if meta_lins:
            T.meta_lin[m] = ptr(_dev(meta_lins[m], f"metadata 1-wide table {m}", torch.float32))
    return T, keep


def make_batch(user, pos, neg=None, pos_meta=None, neg_meta=None, err_flag=None):
    """Build a trs_batch struct from int32/int64 GPU id tensors (all the same dtype)."""
    Bt = TrsBatch()
    ids = [t for t in (user, pos, neg, pos_meta, neg_meta) if t is not None]
    dt = user.dtype
    if dt not in (torch.int32, torch.int64):
        raise TypeError(f"ids must be int32 or int64, got {dt}")
    for t in ids:
        _dev(t, "id tensor", dt)
    B = user.shape[0]
    if pos.shape[0] != B or (neg is not None and neg.shape[0] != B):
        raise ValueError("user/pos/neg id tensors must have the same length")
    Bt.user, Bt.pos, Bt.neg = ptr(user), ptr(pos), ptr(neg)
    Bt.pos_meta, Bt.neg_meta = ptr(pos_meta), ptr(neg_meta)
    Bt.B = B
    Bt.idx_bytes = 4 if dt == torch.int32 else 8
    Bt.err_flag_dev = ptr(err_flag)
    return Bt, ids + [err_flag]


def score_forward(net, T, Bt, B, device, want_neg=True):
    lib = _lib.load()
    pos = torch.empty(B, dtype=torch.float32, device=device)
    neg = torch.empty(B, dtype=torch.float32, device=device) if want_neg else None
    check(lib.trs_score_forward(NET_ID[net], C.byref(T), C.byref(Bt), ptr(pos), ptr(neg), _stream()),
          "trs_score_forward")
    return pos, neg


def score_fwd_bwd(net, T, Bt, B, D, M, device, loss_sum, auc_count=None, want_scores=True, grad_rows=None,
                  grad_lin=None, loss=0):
    """Returns (pos, neg, grad_rows (R,B,D), grad_lin (R,B)); loss_sum/auc_count are accumulated in place."""
    lib = _lib.load()
    R = 3 + 2 * M
    pos = torch.empty(B, dtype=torch.float32, device=device) if want_scores else None
    neg = torch.empty(B, dtype=torch.float32, device=device) if want_scores else None
    if grad_rows is None:
        grad_rows = torch.empty((R, B, D), dtype=torch.float32, device=device)
    if grad_lin is None:
        grad_lin = torch.empty((R, B), dtype=torch.float32, device=device)
    inv_B = 1.0 / B if B > 0 else 0.0
    check(lib.trs_score_fwd_bwd(NET_ID[net], C.byref(T), C.byref(Bt), inv_B, ptr(pos), ptr(neg), ptr(loss_sum),
                                ptr(auc_count), ptr(grad_rows), ptr(grad_lin), int(loss), _stream()),
          "trs_score_fwd_bwd")
    return pos, neg, grad_rows, grad_lin


def score_backward(net, T, Bt, B, D, M, device, gpos, gneg):
    lib = _lib.load()
    R = 3 + 2 * M
    grad_rows = torch.empty((R, B, D), dtype=torch.float32, device=device)
    grad_lin = torch.empty((R, B), dtype=torch.float32, device=device)
    check(lib.trs_score_backward(NET_ID[net], C.byref(T), C.byref(Bt), ptr(gpos), ptr(gneg), ptr(grad_rows),
                                 ptr(grad_lin), _stream()), "trs_score_backward")
    return grad_rows, grad_lin


def score_sgd_update(net, T, Bt, grad_rows, grad_lin, lr):
    check(_lib.load().trs_score_sgd_update(NET_ID[net], C.byref(T), C.byref(Bt), ptr(grad_rows), ptr(grad_lin),
                                           float(lr), _stream()), "trs_score_sgd_update")


class TimingEvents:
    """n HIP events created by the library; elapsed(i, j) after the stream was synchronised."""

    def __init__(self, n):
        self.n = n
        self.handles = (C.c_void_p * n)()
        check(_lib.load().trs_events_create(n, self.handles), "trs_events_create")

    def elapsed_ms(self, i, j):
        out = C.c_float()
        check(_lib.load().trs_events_elapsed_ms(self.handles[i], self.handles[j], C.byref(out)), "trs_events_elapsed_ms")
        return out.value

    def __del__(self):
        try:
            _lib.load().trs_events_destroy(self.n, self.handles)
        except Exception:
            pass


def train_scratch(n_users, n_items, batch, D, device):
    """Zeroed scratch of trs_train_steps_sgd (ownership marks, duplicate stamps, item-bucket lists)."""
    nbytes = _lib.load().trs_train_scratch_bytes(int(n_users), int(n_items), int(batch), int(D))
    return torch.zeros(nbytes // 8 + 1, dtype=torch.int64, device=device)


def interleave_stream(stream_user, stream_item):
    """(N,2) int32 {user, item} pairs: the layout trs_train_steps_sgd reads the resident stream in."""
    return torch.stack([stream_user, stream_item], dim=1).contiguous()


class Sampler:
    """trs_sampler (include/trs.h): negative-sampler options beyond the reference's, for the device RNG mode.
      k            negatives per positive (the epoch then has k * N positions)
      popularity   draw candidates proportionally to the items' frequency in `stream_item`
      seen         (offsets (n_users+1,) int64, items int32 sorted per user): reject items the user has interacted with
      max_tries    candidates tried before the last one is kept"""

    def __init__(self, k=1, popularity=False, seen=None, stream_item=None, max_tries=8):
        self.c = _lib.TrsSampler()
        self.c.k_neg, self.c.popularity, self.c.max_tries = int(k), int(bool(popularity)), int(max_tries)
        self.keep = [seen, stream_item]
        if seen is not None:
            off, items = seen
            _dev(off, "seen offsets", torch.int64)
            _dev(items, "seen items", torch.int32)
            self.c.seen_off, self.c.seen_items, self.c.seen_users = ptr(off), ptr(items), off.numel() - 1
        if popularity:
            _dev(stream_item, "stream items", torch.int32)
            self.c.pop_items, self.c.pop_n = ptr(stream_item), stream_item.numel()
        self.k = int(k)

    @staticmethod
    def seen_csr(user, item, n_users, n_items):
        """CSR of the distinct (user, item) pairs of a device-resident stream: (offsets int64 (n_users+1,), items int32)."""
        key = torch.unique(user.long() * int(n_items) + item.long())
        u = torch.div(key, int(n_items), rounding_mode="floor")
        off = torch.zeros(n_users + 1, dtype=torch.int64, device=user.device)
        off[1:] = torch.cumsum(torch.bincount(u, minlength=n_users), 0)
        return off, (key - u * int(n_items)).to(torch.int32).contiguous()


def _samp(sampler):
    return C.byref(sampler.c) if sampler is not None else None


class EpochPresort:
    """Buffers + result of trs_epoch_presort for n_batches whole batches: id arrays and the item references of every
    batch sorted by row.  `step_args(b)` gives what trs_train_steps_sgd needs to start at batch b of the slice."""

    def __init__(self, n_batches, batch, n_users, n_items, device, item_meta=None, n_meta=(), user_sort=True,
                 item_flags=True):
        """item_meta (n_items, M) int32 + n_meta (categories per column): also sort every metadata column's references.
        user_sort=False (plain SGD without metadata): user duplicates as FLAGS only (trs_epoch_user_flags: the LDS-bitmap
        kernel, no grouping of the users); the step then adds the flagged users' gradients with float atomics."""
        lib = _lib.load()
        self.user_sort = bool(user_sort) or batch > EpochFlags.MAX_BATCH
        self.item_flags = bool(item_flags)  # False: skip the item-duplicate flag pass (only K1's INL 2 mode reads them)
        kb, ktot, vtot, tmp = (C.c_int64() for _ in range(4))
        check(lib.trs_epoch_presort_sizes(n_batches, batch, n_items, C.byref(kb), C.byref(ktot), C.byref(vtot),
                                          C.byref(tmp)), "trs_epoch_presort_sizes")
        self.n_batches, self.batch, self.n_users, self.n_items = n_batches, batch, n_users, n_items
        self.key_bytes = kb.value
        n_pos = n_batches * batch
        self.ids = [torch.empty(n_pos, dtype=torch.int32, device=device) for _ in range(3)]
        self.keys = torch.empty(ktot.value, dtype=torch.uint8, device=device)
        self.vals = torch.empty(vtot.value, dtype=torch.uint8, device=device)
        uk, uv, ut = (C.c_int64() for _ in range(3))
        check(lib.trs_epoch_user_dups_sizes(n_batches, batch, n_users, C.byref(uk), C.byref(uv), C.byref(ut)),
              "trs_epoch_user_dups_sizes")
        self.ukeys = torch.empty(uk.value if self.user_sort else 8, dtype=torch.uint8, device=device)
        self.uvals = torch.empty(uv.value if self.user_sort else 8, dtype=torch.uint8, device=device)
        self.temp_bytes = max(tmp.value, ut.value)
        self.temp = torch.empty(self.temp_bytes, dtype=torch.uint8, device=device)
        self.user_dup = torch.empty(n_pos, dtype=torch.uint8, device=device)  # 1: user has another reference in its batch
        # {pos, neg} per position: 1 = the item row has another reference in the batch (K1 updates the others in place)
        self.item_dup = torch.empty((n_pos, 2), dtype=torch.uint8, device=device)
        self.sorted_keys = self.sorted_vals = None
        self.item_meta, self.n_meta = item_meta, [int(c) for c in n_meta]
        self.meta_keys = [torch.empty(ktot.value, dtype=torch.uint8, device=device) for _ in self.n_meta]
        self.meta_vals = [torch.empty(vtot.value, dtype=torch.uint8, device=device) for _ in self.n_meta]
        self.meta_sorted = []
        M = len(self.n_meta)
        self.pos_meta = torch.empty((n_pos, M), dtype=torch.int32, device=device) if M else None
        self.neg_meta = torch.empty((n_pos, M), dtype=torch.int32, device=device) if M else None

    @staticmethod
    def bytes_needed(n_batches, batch, n_items, n_meta_cols=0):
        lib = _lib.load()
        kb, ktot, vtot, tmp = (C.c_int64() for _ in range(4))
        check(lib.trs_epoch_presort_sizes(n_batches, batch, n_items, C.byref(kb), C.byref(ktot), C.byref(vtot),
                                          C.byref(tmp)), "trs_epoch_presort_sizes")
        return (12 * n_batches * batch + (1 + n_meta_cols) * (ktot.value + vtot.value) + tmp.value
                + 19 * n_batches * batch)

    def run(self, stream_ui, neg_static, shuffle_key, sample_seed, first_pos, err_flag, given_ids=None, sampler=None):
        """Generate (stream_ui given) or adopt (given_ids = (user, pos, neg) int32 tensors) the ids and sort the refs."""
        if given_ids is not None:
            n_pos = self.n_batches * self.batch  # may be a shorter tail slice in the same buffers
            for dst, src in zip(self.ids, given_ids):
                dst[:n_pos].copy_(src[:n_pos])
        sk, sv = C.c_void_p(), C.c_void_p()
        N = 0 if stream_ui is None else stream_ui.shape[0]
        check(_lib.load().trs_epoch_presort(ptr(stream_ui), ptr(neg_static), N, int(shuffle_key), int(sample_seed),
                                            int(first_pos), self.n_batches, self.batch, self.n_users, self.n_items,
                                            ptr(self.ids[0]), ptr(self.ids[1]), ptr(self.ids[2]), ptr(self.keys),
                                            ptr(self.vals), ptr(self.temp), self.temp_bytes, ptr(err_flag),
                                            C.byref(sk), C.byref(sv), ptr(self.item_dup) if self.item_flags else None,
                                            _samp(sampler), _stream()), "trs_epoch_presort")
        self.sorted_keys, self.sorted_vals = sk.value, sv.value
        if self.user_sort:
            uk, uv, ukb = C.c_void_p(), C.c_void_p(), C.c_int32()
            check(_lib.load().trs_epoch_user_dups(ptr(self.ids[0]), self.n_batches, self.batch, self.n_users,
                                                  ptr(self.ukeys), ptr(self.uvals), ptr(self.temp), self.temp_bytes,
                                                  ptr(self.user_dup), C.byref(uk), C.byref(uv), C.byref(ukb), _stream()),
                  "trs_epoch_user_dups")
            self.sorted_ukeys, self.sorted_uvals, self.ukey_bytes = uk.value, uv.value, ukb.value
        else:
            check(_lib.load().trs_epoch_user_flags(ptr(self.ids[0]), self.n_batches, self.batch, self.n_users,
                                                   ptr(self.user_dup), _stream()), "trs_epoch_user_flags")
            self.sorted_ukeys = self.sorted_uvals = None
            self.ukey_bytes = 0
        self.meta_sorted = []
        for m, n_cat in enumerate(self.n_meta):  # metadata columns: the same grouping by row, per column
            mk, mv = C.c_void_p(), C.c_void_p()
            check(_lib.load().trs_epoch_presort_meta(ptr(self.ids[1]), ptr(self.ids[2]), self.n_batches, self.batch,
                                                     ptr(self.item_meta), len(self.n_meta), m, n_cat,
                                                     ptr(self.meta_keys[m]), ptr(self.meta_vals[m]), ptr(self.temp),
                                                     self.temp_bytes, ptr(err_flag), C.byref(mk), C.byref(mv),
                                                     ptr(self.pos_meta), ptr(self.neg_meta), _stream()),
                  "trs_epoch_presort_meta")
            self.meta_sorted.append((mk.value, mv.value))

    def meta_step_args(self, b):
        """[(sorted keys address, sorted vals address)] of every metadata column for the steps starting at batch b."""
        o = b * self.batch
        return [(k + 2 * o * 4, v + 2 * o * 4) for k, v in self.meta_sorted]

    def meta_id_args(self, b):
        """(pos, neg) metadata id views (positions from batch b on, M) written by the presort, or (None, None)."""
        if not self.meta_sorted:
            return None, None
        o = b * self.batch
        return self.pos_meta[o:], self.neg_meta[o:]

    def step_args(self, b):
        """(user, pos, neg id views, sorted keys address, sorted vals address, user-duplicate flags view, sorted user
        runs, item-duplicate flags view) for the steps starting at batch b."""
        o = b * self.batch
        usorted = ((self.sorted_ukeys + o * self.ukey_bytes, self.sorted_uvals + o * 4, self.ukey_bytes, o)
                   if self.sorted_ukeys is not None else (None, None, 0, o))
        return ([t[o:] for t in self.ids], self.sorted_keys + 2 * o * self.key_bytes, self.sorted_vals + 2 * o * 4,
                self.user_dup[o:], usorted, self.item_dup[o:])


class EpochFlags:
    """The sparse regime's presort (trs_epoch_flags): ids of n_batches whole batches + conservative duplicate flags,
    one launch, no sort.  Same surface as EpochPresort where the step loop needs it."""

    MAX_BATCH = 262_144  # one workgroup per batch (csrc/presort.hip FLAG_THREADS * FLAG_U * FLAG_ROUNDS * FLAG_MAX_GROUPS)

    def __init__(self, n_batches, batch, n_users, n_items, device, ordered=True):
        self.n_batches, self.batch, self.n_users, self.n_items = n_batches, batch, n_users, n_items
        n_pos = n_batches * batch
        self.ids = [torch.empty(n_pos, dtype=torch.int32, device=device) for _ in range(3)]
        self.user_dup = torch.empty(n_pos, dtype=torch.uint8, device=device)
        self.item_dup = torch.empty((n_pos, 2), dtype=torch.uint8, device=device)
        # ordered: every batch comes out with its flagged triples first (trs_epoch_flags_ordered) and their count here —
        # what lets the one-launch step count a workgroup in early (trs_train_args.n_flagged_dev)
        self.n_flagged = torch.zeros(n_batches, dtype=torch.int32, device=device) if ordered else None
        self.key_bytes = self.ukey_bytes = 0

    @staticmethod
    def bytes_needed(n_batches, batch):
        return 15 * n_batches * batch

    def run(self, stream_ui, neg_static, shuffle_key, sample_seed, first_pos, err_flag, given_ids=None, sampler=None,
            n_batches=None):
        """n_batches: fewer batches than the buffers hold (an epoch's last, shorter slice)."""
        nb = self.n_batches if n_batches is None else int(n_batches)
        assert 0 < nb <= self.n_batches
        if given_ids is not None:
            n_pos = nb * self.batch
            for dst, src in zip(self.ids, given_ids):
                dst[:n_pos].copy_(src[:n_pos])
        N = 0 if stream_ui is None else stream_ui.shape[0]
        check(_lib.load().trs_epoch_flags_ordered(ptr(stream_ui), ptr(neg_static), N, int(shuffle_key), int(sample_seed),
                                                  int(first_pos), nb, self.batch, self.n_users,
                                                  self.n_items, ptr(self.ids[0]), ptr(self.ids[1]), ptr(self.ids[2]),
                                                  ptr(self.user_dup), ptr(self.item_dup), ptr(self.n_flagged),
                                                  ptr(err_flag), _samp(sampler), _stream()),
              "trs_epoch_flags_ordered")

    @property
    def base_ptrs(self):
        """Device addresses of (user, pos, neg ids; user / item duplicate flags) — FlagStepCall offsets them itself."""
        bp = getattr(self, "_base_ptrs", None)
        if bp is None:
            bp = self._base_ptrs = tuple(t.data_ptr() for t in self.ids) + (
                self.user_dup.data_ptr(), self.item_dup.data_ptr(),
                self.n_flagged.data_ptr() if self.n_flagged is not None else 0)
        return bp

    def step_args(self, b):
        """(id views, user-duplicate flags view, item-duplicate flags view) from batch b of the slice on."""
        o = b * self.batch
        return [t[o:] for t in self.ids], self.user_dup[o:], self.item_dup[o:]

    def n_flagged_from(self, b):
        """Flagged-triple counts from batch b on (None: the batches are in arbitrary order)."""
        return None if self.n_flagged is None else self.n_flagged[b:]


def train_steps_sgd(net, T, stream_ui, neg_static, shuffle_key, sample_seed, first_pos, batch, n_steps,
                    lr, user_buf, pos_buf, neg_buf, gz_buf, du_buf, loss_sums, err_flag, scratch=None, first_stamp=1,
                    events=None, sorted_keys=None, sorted_vals=None, key_bytes=0, user_dup=None, ustage=None,
                    user_sorted=None, opt=None, meta=None, item_dup=None, loss=0, sync=None, n_flagged=None):
    """n_steps fused steps driven from C (trs_train_steps_sgd).  stream_ui None: the steps' ids are already in
    user/pos/neg_buf.  events: optional flat list of 4*n_steps raw hipEvent_t handles.  opt: None (SGD with lr) or a
    _lib.TrsOpt (SparseAdam / Adagrad on the presorted path; keep the tensors it points to alive).  item_dup: the
    presort's item-duplicate flags (plain SGD without metadata): K1 also updates item rows referenced once.  Flag mode
    (sparse regime): user_dup + item_dup from an EpochFlags, ustage, and no sorted references; sync = (zeroed int32
    device tensor, ctypes c_uint32 host counter): the flagged references are applied by K1's own launch whenever its
    grid is resident at once."""
    a = _lib.TrsTrainArgs()
    if sync is not None:
        a.sync_dev, a.sync_count_host = ptr(sync[0]), C.pointer(sync[1])
    a.n_flagged_dev = ptr(n_flagged)  # (n_steps int32 of an ordered EpochFlags, or None)
    a.net, a.n_steps, a.tables, a.batch, a.lr = NET_ID[net], int(n_steps), C.pointer(T), int(batch), float(lr)
    a.first_stamp = int(first_stamp)
    a.loss = int(loss)
    a.stream_ui_dev, a.neg_static_dev = ptr(stream_ui), ptr(neg_static)
    a.N = 0 if stream_ui is None else stream_ui.shape[0]
    a.shuffle_key, a.sample_seed, a.first_pos = int(shuffle_key), int(sample_seed), int(first_pos)
    a.user_buf_dev, a.pos_buf_dev, a.neg_buf_dev = ptr(user_buf), ptr(pos_buf), ptr(neg_buf)
    a.gz_buf_dev, a.du_buf_dev, a.loss_sums_dev = ptr(gz_buf), ptr(du_buf), ptr(loss_sums)
    a.err_flag_dev, a.scratch_dev = ptr(err_flag), ptr(scratch)
    a.sorted_keys_dev, a.sorted_vals_dev, a.key_bytes = sorted_keys, sorted_vals, int(key_bytes)
    a.user_dup_flags_dev, a.item_dup_flags_dev, a.ustage_buf_dev = ptr(user_dup), ptr(item_dup), ptr(ustage)
    if user_sorted is not None:
        a.sorted_ukeys_dev, a.sorted_uvals_dev, a.ukey_bytes, a.slice_pos0 = user_sorted
    if opt is not None:
        a.opt = C.pointer(opt)
    if meta is not None:
        a.meta = C.pointer(meta)
    if events is not None:
        ev = (C.c_void_p * len(events))(*events)  # raw hipEvent_t handles (or None = step not timed)
        a.events = C.cast(ev, C.POINTER(C.c_void_p))
    check(_lib.load().trs_train_steps_sgd(C.byref(a), _stream()), "trs_train_steps_sgd")


TIMELINE = None  # diagnostic (bench.py TRS_BENCH_TIMELINE=1): a list that receives (label, time.perf_counter()) stamps


def stamp(label):
    if TIMELINE is not None:
        import time
        TIMELINE.append((label, time.perf_counter()))


class FlagStepCall:
    """trs_train_steps_sgd in flag mode (sparse regime) with the argument struct built ONCE: a call only writes the
    fields that change (step count, stamp, the slice offsets of ids / flags, the loss slot).  The generic wrapper above
    rebuilds ~40 ctypes fields per call — 50-80 us of host time that a short timed window sees as start-up latency."""

    def __init__(self, net, T, batch, lr, gz_buf, du_buf, err_flag, scratch, ustage, loss, sync=None):
        a = self.a = _lib.TrsTrainArgs()
        self.keep = (T, gz_buf, du_buf, err_flag, scratch, ustage, sync)
        if sync is not None:
            a.sync_dev, a.sync_count_host = ptr(sync[0]), C.pointer(sync[1])
        a.net, a.tables, a.batch, a.lr, a.loss = NET_ID[net], C.pointer(T), int(batch), float(lr), int(loss)
        a.gz_buf_dev, a.du_buf_dev = ptr(gz_buf), ptr(du_buf)
        a.err_flag_dev, a.scratch_dev, a.ustage_buf_dev = ptr(err_flag), ptr(scratch), ptr(ustage)
        self.batch = int(batch)
        self.ref = C.byref(a)
        self.fn = _lib.load().trs_train_steps_sgd
        self.sig = (C.addressof(T), float(lr), int(loss), ptr(gz_buf), ptr(du_buf), ptr(ustage), ptr(scratch))

    def __call__(self, ps, b_in_slice, n_steps, loss_sums, first_stamp):
        a, o = self.a, b_in_slice * self.batch
        base = ps.base_ptrs
        a.n_steps, a.first_stamp = int(n_steps), int(first_stamp)
        a.user_buf_dev, a.pos_buf_dev, a.neg_buf_dev = base[0] + 4 * o, base[1] + 4 * o, base[2] + 4 * o
        a.user_dup_flags_dev, a.item_dup_flags_dev = base[3] + o, base[4] + 2 * o
        a.n_flagged_dev = base[5] + 4 * b_in_slice if base[5] else None
        a.loss_sums_dev = loss_sums.data_ptr()
        check(self.fn(self.ref, _stream()), "trs_train_steps_sgd")


def rows_scatter_add(table, idx, vals, alpha, ld=None, err_flag=None):
    n_rows, D = table.shape
    n = idx.shape[0]
    ld = vals.stride(0) if ld is None else ld
    check(_lib.load().trs_rows_scatter_add(ptr(table), n_rows, D, ptr(idx), 4 if idx.dtype == torch.int32 else 8,
                                           ptr(vals), ld, n, float(alpha), ptr(err_flag), _stream()),
          "trs_rows_scatter_add")


def rows_apply_sparse_adam(table, acc, exp_avg, exp_avg_sq, stamp, idx, step_id, lr, beta1, beta2, eps, step_count):
    n_rows, D = table.shape
    check(_lib.load().trs_rows_apply_sparse_adam(ptr(table), ptr(acc), ptr(exp_avg), ptr(exp_avg_sq), ptr(stamp),
                                                 n_rows, D, ptr(idx), 4 if idx.dtype == torch.int32 else 8,
                                                 idx.shape[0], int(step_id), float(lr), float(beta1), float(beta2),
                                                 float(eps), int(step_count), _stream()),
          "trs_rows_apply_sparse_adam")


def rows_apply_adagrad(table, acc, state_sum, stamp, idx, step_id, clr, eps):
    n_rows, D = table.shape
    check(_lib.load().trs_rows_apply_adagrad(ptr(table), ptr(acc), ptr(state_sum), ptr(stamp), n_rows, D, ptr(idx),
                                             4 if idx.dtype == torch.int32 else 8, idx.shape[0], int(step_id),
                                             float(clr), float(eps), _stream()), "trs_rows_apply_adagrad")


def hinge_auc(pos, neg, loss_sum, auc_count, loss=0):
    """loss: 0 = hinge (the reference), 1 = BPR (_lib.LOSS_ID)."""
    check(_lib.load().trs_hinge_auc(ptr(pos), ptr(neg), pos.numel(), ptr(loss_sum), ptr(auc_count), int(loss),
                                    _stream()), "trs_hinge_auc")


def hinge_auc_batches(pos, neg, batch, loss_sums, auc_counts, loss=0):
    """Per-batch hinge sums / AUC counts of consecutive `batch`-row pieces of pos / neg into loss_sums[b], auc_counts[b]."""
    check(_lib.load().trs_hinge_auc_batches(ptr(pos), ptr(neg), pos.numel(), batch, ptr(loss_sums), ptr(auc_counts),
                                            int(loss), _stream()), "trs_hinge_auc_batches")


def hinge_backward(pos, neg, loss=0):
    """-> (gp, gn): the two halves of one (2B,) buffer (`gp._base` is the concatenated gradient the MLP backward reads)."""
    B = pos.numel()
    g = torch.empty(2 * B, dtype=pos.dtype, device=pos.device)
    gp, gn = g[:B], g[B:]
    check(_lib.load().trs_hinge_backward(ptr(pos), ptr(neg), B, 1.0 / B if B else 0.0, ptr(gp), ptr(gn), int(loss),
                                         _stream()), "trs_hinge_backward")
    return gp, gn


def hinge_auc_backward(pos, neg, loss_sum, auc_count, loss=0):
    """hinge_auc + hinge_backward in one launch -> (gp, gn), the halves of one (2B,) buffer."""
    B = pos.numel()
    g = torch.empty(2 * B, dtype=pos.dtype, device=pos.device)
    gp, gn = g[:B], g[B:]
    check(_lib.load().trs_hinge_auc_backward(ptr(pos), ptr(neg), B, 1.0 / B if B else 0.0, ptr(loss_sum),
                                             ptr(auc_count), ptr(gp), ptr(gn), int(loss), _stream()),
          "trs_hinge_auc_backward")
    return gp, gn


def sample_neg(pos, n_items, seed, offset):
    neg = torch.empty_like(pos)
    check(_lib.load().trs_sample_neg(ptr(pos), 4 if pos.dtype == torch.int32 else 8, pos.numel(), int(n_items),
                                     int(seed), int(offset), ptr(neg), _stream()), "trs_sample_neg")
    return neg


def batch_prepare(stream_user, stream_item, neg_static, shuffle_key, t0, B, n_items, seed, offset, item_meta=None,
                  out=None, sampler=None):
    """Returns dict of int32 GPU tensors user/pos/neg[/pos_meta/neg_meta] for epoch positions [t0, t0+B)."""
    dev = stream_user.device
    M = 0 if item_meta is None else item_meta.shape[1]
    if out is None:
        out = {k: torch.empty(B, dtype=torch.int32, device=dev) for k in ("user", "pos", "neg")}
        if M:
            out["pos_meta"] = torch.empty((B, M), dtype=torch.int32, device=dev)
            out["neg_meta"] = torch.empty((B, M), dtype=torch.int32, device=dev)
    check(_lib.load().trs_batch_prepare(ptr(stream_user), ptr(stream_item), ptr(neg_static), stream_user.numel(),
                                        int(shuffle_key), int(t0), int(B), int(n_items), int(seed), int(offset),
                                        ptr(item_meta), M, ptr(out["user"]), ptr(out["pos"]), ptr(out["neg"]),
                                        ptr(out.get("pos_meta")), ptr(out.get("neg_meta")), _samp(sampler), _stream()),
          "trs_batch_prepare")
    return out


def score_all_items(net, T, user_id, n_items, device, item_meta=None, item0=0, n=None):
    n = n_items - item0 if n is None else n
    out = torch.empty(n, dtype=torch.float32, device=device)
    check(_lib.load().trs_score_all_items(NET_ID[net], C.byref(T), int(user_id), int(item0), int(n), ptr(item_meta),
                                          ptr(out), _stream()), "trs_score_all_items")
    return out


def topk(scores, k):
    lib = _lib.load()
    n = scores.numel()
    ws_bytes = lib.trs_topk_workspace_bytes(n, k)
    ws = torch.empty(max(ws_bytes, 8), dtype=torch.uint8, device=scores.device)
    out = torch.empty(k, dtype=torch.int64, device=scores.device)
    check(lib.trs_topk(ptr(scores), n, k, ptr(out), ptr(ws), ws_bytes, _stream()), "trs_topk")
    return out


# ------------------------------------------------------------------------------------------------- MLP kernels
def mlp_gather_concat(T, Bt, passes, x=None, x16=None):
    """x: fp32 (rows, (2+M)D) and/or x16: the same image in bfloat16 (same row stride in elements)."""
    ld = (x if x is not None else x16).stride(0)
    if x is not None and x16 is not None and x16.stride(0) != ld:
        raise ValueError("mlp_gather_concat: x and x16 must share the row stride")
    check(_lib.load().trs_mlp_gather_concat(C.byref(T), C.byref(Bt), passes, ptr(x), ptr(x16), ld, _stream()),
          "trs_mlp_gather_concat")


def mlp_gather_gemm1(T, Bt, passes, W, bias, y, bn_part=None, x_image=None):
    """MLP layer 0 with the gather inside the GEMM (trs_mlp_gather_gemm1_fwd).  W: the fp32 weight (N, K) or its bfloat16
    image; y: preallocated (rows, N) fp32 or bfloat16 output; x_image: optional (rows, K) buffer for the x0 image the
    weight-gradient GEMM reads (fp32 with an fp32 W, bfloat16 with a bfloat16 W).  Returns False when the shape is not
    one the fused kernels take (nothing launched: run mlp_gather_concat + gemm)."""
    bf16 = W.dtype == torch.bfloat16
    y32 = y if y.dtype == torch.float32 else None
    y16 = y if y.dtype == torch.bfloat16 else None
    x32 = x_image if (x_image is not None and x_image.dtype == torch.float32) else None
    x16 = x_image if (x_image is not None and x_image.dtype == torch.bfloat16) else None
    rc = _lib.load().trs_mlp_gather_gemm1_fwd(C.byref(T), C.byref(Bt), passes, int(bf16), ptr(W), W.stride(0), ptr(bias),
                                              W.shape[0], ptr(y32), ptr(y16), y.stride(0), ptr(bn_part), ptr(x32),
                                              ptr(x16), x_image.stride(0) if x_image is not None else 0, _stream())
    if rc == 1:
        return False
    check(rc, "trs_mlp_gather_gemm1_fwd")
    return True


def mlp_embed_sgd_supported(T):
    return bool(_lib.load().trs_mlp_embed_sgd_update_supported(C.byref(T)))


def mlp_embed_sgd_update(T, Bt, dx0, lr, user_dup=None, item_dup=None):
    """Embedding-row SGD of an MLP step from d x0 ((2B, ld) fp32 or bfloat16): user / item rows (+ optional duplicate
    flags: unflagged references are plain read-modify-writes) and the owner-computes metadata update."""
    f32 = dx0 if dx0.dtype == torch.float32 else None
    b16 = dx0 if dx0.dtype == torch.bfloat16 else None
    if f32 is None and b16 is None:
        raise TypeError(f"mlp_embed_sgd_update: d x0 must be float32 or bfloat16, got {dx0.dtype}")
    check(_lib.load().trs_mlp_embed_sgd_update(C.byref(T), C.byref(Bt), ptr(f32), ptr(b16), dx0.stride(0), float(lr),
                                               ptr(user_dup), ptr(item_dup), _stream()), "trs_mlp_embed_sgd_update")


_gemm_ws = {}


def _workspace(dev, nbytes):
    """One grow-only scratch buffer per device for split-K slabs / reduction partials."""
    buf = _gemm_ws.get(dev)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=dev)
        _gemm_ws[dev] = buf
    return buf


GEMM_TILE_ROWS = 128  # rows per output tile = rows per fused-BN-statistics chunk


def gemm(transA, transB, A, B, out=None, bias=None, alpha=1.0, beta=0.0, bf16=False, bn_part=None):
    """out(M,N) = alpha * op(A) op(B) + beta * out (+ bias); fp32 MFMA, or bf16-rounded operands with fp32
    accumulation when bf16=True.  A, B, out: 2-D fp32 GPU tensors whose last
    dimension is contiguous (row stride = leading dimension)."""
    lib = _lib.load()
    M, K = (A.shape[1], A.shape[0]) if transA else (A.shape[0], A.shape[1])
    Kb, N = (B.shape[1], B.shape[0]) if transB else (B.shape[0], B.shape[1])
    if K != Kb:
        raise ValueError(f"gemm: inner dimensions differ ({K} vs {Kb})")
    for t in (A, B):
        if t.dtype != torch.float32 or t.stride(-1) != 1 or not t.is_cuda:
            raise ValueError("gemm operands must be fp32 GPU tensors with a contiguous last dimension")
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    wsb = 0 if bn_part is not None else lib.trs_gemm_f32_workspace_bytes(M, N, K)
    ws = _workspace(A.device, wsb) if wsb else None
    fn = lib.trs_gemm_bf16 if bf16 else lib.trs_gemm_f32
    check(fn(int(transA), int(transB), M, N, K, float(alpha), ptr(A), A.stride(0), ptr(B), B.stride(0),
             float(beta), ptr(out), out.stride(0), ptr(bias), ptr(bn_part), ptr(ws), wsb, _stream()), "trs_gemm")
    return out


def gemm_bf16in(tn, A, B, out=None, bias=None, bn_part=None, out_bf16=False, alpha=1.0, beta=0.0):
    """bf16-resident GEMM: tn False: out(M,N) = A(M,K) B(N,K)^T; tn True: out(M,N) = A(K,M)^T B(K,N).  A, B bfloat16 GPU
    tensors with a contiguous last dimension; fp32 accumulation; fp32 output, or bfloat16 when out_bf16 (or `out` is a
    bfloat16 tensor).  alpha, beta: out = alpha * (product) + beta * out (fp32 `out` given; beta != 0 reads it)."""
    lib = _lib.load()
    for t in (A, B):
        if t.dtype != torch.bfloat16 or t.stride(-1) != 1 or not t.is_cuda:
            raise ValueError("gemm_bf16in operands must be bfloat16 GPU tensors with a contiguous last dimension")
    if tn:
        K, M = A.shape
        Kb, N = B.shape
    else:
        M, K = A.shape
        N, Kb = B.shape
    if K != Kb:
        raise ValueError(f"gemm_bf16in: inner dimensions differ ({K} vs {Kb})")
    if out is None:
        out = torch.empty((M, N), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=A.device)
    o16 = out.dtype == torch.bfloat16
    wsb = 0 if (bn_part is not None or o16) else lib.trs_gemm_bf16in_workspace_bytes(M, N, K)
    ws = _workspace(A.device, wsb) if wsb else None
    if beta != 0.0 and (o16 or bn_part is not None):
        raise ValueError("gemm_bf16in: beta != 0 needs an fp32 `out` and no fused statistics")
    check(lib.trs_gemm_bf16in(int(tn), M, N, K, float(alpha), ptr(A), A.stride(0), ptr(B), B.stride(0), float(beta),
                              None if o16 else ptr(out), ptr(out) if o16 else None, out.stride(0), ptr(bias),
                              ptr(bn_part), ptr(ws), wsb, _stream()), "trs_gemm_bf16in")
    return out


def bn_relu_forward_forms_dot(y):
    """trs_bn_relu_forward forms the output-layer dot inside its launch for this pre-activation tensor (a row's columns in
    one wave: H a power of two in 4..256, 4-element-aligned rows) — the activations then need not be stored."""
    H = y.shape[1]
    return 4 <= H <= 256 and (H & (H - 1)) == 0 and y.stride(0) % 4 == 0 and y.data_ptr() % 16 == 0


def gemm_bf16in_ok(M, N, K):
    """Shapes the bf16-resident kernels take (all tiles interior)."""
    return M % GEMM_TILE_ROWS == 0 and N % 128 == 0 and K % 64 == 0


def f32_to_bf16(src, dst=None, dst_t=None):
    """bf16 copy (dst, same shape) and/or transposed bf16 copy (dst_t) of a 2-D fp32 GPU tensor."""
    rows, cols = src.shape
    check(_lib.load().trs_f32_to_bf16(ptr(src), rows, cols, src.stride(0), ptr(dst), ptr(dst_t), _stream()),
          "trs_f32_to_bf16")


class WeightImages:
    """trs_f32_to_bf16_multi with its argument arrays kept: the bf16 (and transposed bf16) images of up to 8 fp32 matrices
    refreshed by one launch (srcs[k] (rows, cols) -> dsts[k] (rows, cols), dsts_t[k] (cols, rows))."""

    MAX = 8

    def __init__(self, srcs, dsts, dsts_t):
        n = len(srcs)
        if not 1 <= n <= self.MAX:
            raise ValueError(f"WeightImages: 1..{self.MAX} matrices")
        self.keep = (list(srcs), list(dsts), list(dsts_t))
        self.key = tuple(t.data_ptr() for t in srcs)
        self.n = n
        self.src = (C.c_void_p * n)(*[t.data_ptr() for t in srcs])
        self.dst = (C.c_void_p * n)(*[ptr(t) for t in dsts])
        self.dst_t = (C.c_void_p * n)(*[ptr(t) for t in dsts_t])
        self.rows = (C.c_int64 * n)(*[t.shape[0] for t in srcs])
        self.cols = (C.c_int64 * n)(*[t.shape[1] for t in srcs])
        self.ld = (C.c_int64 * n)(*[t.stride(0) for t in srcs])

    def refresh(self):
        check(_lib.load().trs_f32_to_bf16_multi(self.n, self.src, self.rows, self.cols, self.ld, self.dst, self.dst_t,
                                                _stream()), "trs_f32_to_bf16_multi")


def bn_batch_stats(y, rows_per_pass, passes, momentum, mean_out, var_out, running_mean, running_var):
    lib = _lib.load()
    H = y.shape[1]
    ws = _workspace(y.device, 4 * lib.trs_bn_workspace_floats(rows_per_pass, H, passes))
    check(lib.trs_bn_batch_stats(ptr(y), rows_per_pass, H, y.stride(0), passes, float(momentum), ptr(mean_out),
                                 ptr(var_out), ptr(running_mean), ptr(running_var), ptr(ws), _stream()),
          "trs_bn_batch_stats")


def bn_stats_finalize(part, rows_per_pass, chunk_rows, H, passes, momentum, mean_out, var_out, running_mean,
                      running_var):
    check(_lib.load().trs_bn_stats_finalize(ptr(part), rows_per_pass, chunk_rows, H, passes, float(momentum),
                                            ptr(mean_out), ptr(var_out), ptr(running_mean), ptr(running_var), _stream()),
          "trs_bn_stats_finalize")


def bn_relu_forward(y, rows_per_pass, passes, use_bn, stat_passes, mean, var, gamma, beta, eps, out=None, out16=None,
                    momentum=0.0, running_mean=None, running_var=None, dot=None, tracked=None):
    """out: fp32 and/or out16: bfloat16 image of relu(bn(y)) (same row stride in elements).  running_mean/var: the
    momentum update of the running statistics rides in this launch (instead of bn_stats_finalize's), and with it
    tracked (BatchNorm1d.num_batches_tracked, int64 scalar tensor) += passes.  dot=(w, bias,
    scores): the H -> 1 output layer scores[r] = out[r] . w + bias from the same launch where the layer is narrow enough
    (rowdot on `out` otherwise)."""
    H = y.shape[1]
    ldo = (out if out is not None else out16).stride(0) if (out is not None or out16 is not None) else H
    check(_lib.load().trs_bn_relu_forward(ptr(y), int(y.dtype == torch.bfloat16), rows_per_pass, passes, H, y.stride(0),
                                          int(use_bn), stat_passes,
                                          ptr(mean), ptr(var), ptr(gamma), ptr(beta), float(eps), ptr(out), ptr(out16),
                                          ldo, float(momentum), ptr(running_mean), ptr(running_var), ptr(tracked),
                                          *((ptr(t) for t in dot) if dot is not None else (None, None, None)), _stream()),
          "trs_bn_relu_forward")


def bn_relu_backward(y, dx, rows_per_pass, passes, use_bn, mean, var, gamma, beta, eps, dy, dgamma, dbeta,
                     dy_colsum=None, dy16=None, phase=0, sums=None, stat_rows=0, outer=None, outer_xw=None):
    """phase 0: whole backward on the local batch.  Synchronised BatchNorm: phase 1 (reduce: `sums` (passes,2,H) fp32
    receives sum(d), sum(d*xhat)), all-reduce `sums`, phase 2 (apply with stat_rows = world * rows_per_pass).
    outer=(g, w) with dx=None: dx[r][c] = g[r] * w[c] formed inside the kernels (the output layer's input gradient);
    outer_xw (H): receives sum_r g[r] * relu(bn(y))[r] — the output layer's weight gradient without stored activations."""
    lib = _lib.load()
    H = y.shape[1]
    ws = _workspace(y.device, 4 * lib.trs_bn_backward_workspace_floats(rows_per_pass, H, passes))
    ldd = dx.stride(0) if dx is not None else (dy16 if dy16 is not None else dy).stride(0)
    og, ow = outer if outer is not None else (None, None)
    check(lib.trs_bn_relu_backward(ptr(y), int(y.dtype == torch.bfloat16), ptr(dx),
                                   int(dx is not None and dx.dtype == torch.bfloat16),
                                   rows_per_pass, passes, H, y.stride(0), ldd, int(use_bn),
                                   ptr(mean), ptr(var), ptr(gamma), ptr(beta), float(eps), ptr(dy), ptr(dy16),
                                   ptr(dgamma), ptr(dbeta), ptr(dy_colsum), ptr(ws), int(phase), ptr(sums),
                                   int(stat_rows), ptr(og), ptr(ow), ptr(outer_xw), _stream()), "trs_bn_relu_backward")


def colsum(x, out, row_weight=None, passes=1):
    lib = _lib.load()
    rows, H = x.shape
    rpp = rows // passes
    ws = _workspace(x.device, 4 * lib.trs_colsum_workspace_floats(rpp, passes, H))
    check(lib.trs_colsum(ptr(x), rpp, passes, H, x.stride(0), ptr(row_weight), ptr(out), ptr(ws), _stream()),
          "trs_colsum")


def rowdot(x, w, bias, out):
    rows, H = x.shape
    check(_lib.load().trs_rowdot(ptr(x), rows, H, x.stride(0), ptr(w), ptr(bias), ptr(out), _stream()), "trs_rowdot")


def outer(g, w, dx):
    rows, H = dx.shape
    check(_lib.load().trs_outer(ptr(g), ptr(w), rows, H, ptr(dx), dx.stride(0), _stream()), "trs_outer")
