# -*- coding: utf-8 -*-
"""Shared host glue of the Linear and FM scorers: builds the C-ABI structs from the module's parameters and bridges
the fused HIP forward/backward into torch.autograd (sparse COO gradients, as nn.Embedding(sparse=True) produces in
the reference — third-party EmbeddingBackward, called from reference model.py:197)."""
import torch

from .. import ops

_NO_GPU = ("torchrecsys_amd scorers compute on the MI355X only: the module's parameters are on {dev}. Construct the "
           "model on a machine with a gfx950 GPU (TorchRecSys moves the net there) — there is no CPU fallback.")


def as_id_matrix(meta, M):
    """Metadata ids as the (B, M) integer contract of the reference nets."""
    if meta is None:
        raise KeyError("this scorer was built with use_metadata=True: the batch needs a metadata id tensor")
    if meta.dim() == 1:
        meta = meta.reshape(-1, 1)
    if meta.dim() == 3 and meta.shape[2] == 1:
        meta = meta[:, :, 0]
    if meta.dim() != 2 or meta.shape[1] != M:
        raise ValueError(f"metadata ids must be (B, {M}) integers, one column per metadata table; got {tuple(meta.shape)}")
    return meta


class SparseScorer(torch.nn.Module):
    """Base of Linear / FM.  Subclasses define NET ('linear' | 'fm') and the attribute names of their tables."""
    NET = None
    LIN_NAMES = (None, None)        # attribute names of the 1-wide user / item tables
    META_LIN_NAME = None            # ModuleList attribute of the 1-wide metadata tables (FM) or None
    META_NAME = "metadata"

    # -------------------------------------------------------------------------------------------- structs
    def table_params(self):
        """[user, item, user_lin, item_lin, meta_0.., meta_lin_0..] weights, the order used by the autograd bridge."""
        ps = [self.user.weight, self.item.weight, getattr(self, self.LIN_NAMES[0]).weight,
              getattr(self, self.LIN_NAMES[1]).weight]
        if self.use_metadata:
            ps += [l.weight for l in getattr(self, self.META_NAME)]
            if self.META_LIN_NAME:
                ps += [l.weight for l in getattr(self, self.META_LIN_NAME)]
        return ps

    def n_meta_tables(self):
        return len(getattr(self, self.META_NAME)) if self.use_metadata else 0

    def tables(self):
        ps = self.table_params()
        dev = ps[0].device
        if dev.type != "cuda":
            raise RuntimeError(_NO_GPU.format(dev=dev))
        key = tuple(p.data_ptr() for p in ps)
        cache = getattr(self, "_tables_cache", None)
        if cache is None or cache[0] != key:
            M = self.n_meta_tables()
            metas = [p.data for p in ps[4:4 + M]]
            meta_lins = [p.data for p in ps[4 + M:4 + 2 * M]] if self.META_LIN_NAME else []
            T, keep = ops.make_tables(ps[0].data, ps[1].data, ps[2].data, ps[3].data, metas, meta_lins)
            cache = (key, T, keep)
            self._tables_cache = cache
        return cache[1]

    def device_ids(self, batch, user_key, item_key, metadata_key, neg_item_key=None, neg_metadata_key=None):
        """Move one batch's id tensors to the parameters' device (int32/int64 kept, contiguous)."""
        dev = self.user.weight.device
        if dev.type != "cuda":
            raise RuntimeError(_NO_GPU.format(dev=dev))
        M = self.n_meta_tables()

        def mv(t):
            if t.dtype not in (torch.int32, torch.int64):
                t = t.long()
            return t.to(dev, non_blocking=True).contiguous()

        ids = {"user": mv(batch[user_key]), "pos": mv(batch[item_key])}
        ids["neg"] = mv(batch[neg_item_key]) if neg_item_key else None
        if M:
            ids["pos_meta"] = mv(as_id_matrix(batch.get(metadata_key) if metadata_key else None, M))
            ids["neg_meta"] = mv(as_id_matrix(batch.get(neg_metadata_key), M)) if neg_item_key else None
        dts = {t.dtype for t in ids.values() if t is not None}
        if len(dts) > 1:
            ids = {k: (None if t is None else t.long()) for k, t in ids.items()}
        return ids

    # -------------------------------------------------------------------------------------------- forward
    def _shape_out(self, s):
        return s.reshape(-1, 1) if self.NET == "linear" else s

    def forward(self, batch, user_key, item_key, metadata_key=None):
        """One scoring pass (reference collaborative/linear.py:54-80, fm.py:60-101)."""
        ids = self.device_ids(batch, user_key, item_key, metadata_key)
        pos, _ = _PairScore.apply(self, ids, False, *self.table_params())
        return self._shape_out(pos)

    def forward_pair(self, batch, user_key="user_id", pos_key="pos_item_id", neg_key="neg_item_id",
                     pos_meta_key="pos_metadata_id", neg_meta_key="neg_metadata_id"):
        """Positive and negative pass fused in one kernel (the two net.forward calls of reference model.py:171-185);
        the user row is gathered once."""
        ids = self.device_ids(batch, user_key, pos_key, pos_meta_key, neg_key, neg_meta_key)
        pos, neg = _PairScore.apply(self, ids, True, *self.table_params())
        return self._shape_out(pos), self._shape_out(neg)


    # -------------------------------------------------------------------------------------------- inference
    def score_ids(self, ids):
        """Positive and negative scores of a device-resident batch (no autograd): evaluate()."""
        dev = self.user.weight.device
        err = torch.zeros(1, dtype=torch.int32, device=dev)
        B = ids["user"].shape[0]
        Bt, keep = ops.make_batch(ids["user"], ids["pos"], ids["neg"], ids.get("pos_meta"), ids.get("neg_meta"), err)
        pos, neg = ops.score_forward(self.NET, self.tables(), Bt, B, dev)
        check_err_flag(err, "evaluate")
        return pos, neg

    def score_all_items(self, user_id, item_meta_dev=None):
        """Scores of one user against every item, (n_items,) fp32 on the GPU: predict()."""
        if not 0 <= user_id < self.n_users:
            raise IndexError(f"index out of range in self (user_id {user_id} outside [0, {self.n_users}))")
        return ops.score_all_items(self.NET, self.tables(), user_id, self.n_items, self.user.weight.device,
                                   item_meta_dev)


def check_err_flag(err, what):
    code = int(err.item())
    if code & 4:  # bit 2: csrc/fast_step.hip, the one-launch flag-mode step
        err.zero_()
        raise RuntimeError(f"{what}: the one-launch training step could not get all its workgroups resident on the GPU "
                           "at once within 50 ms (another process is computing on the same device?); the step that "
                           "timed out is not exact.  Set TRS_FLAG_ONE_LAUNCH=0 to run the step as two launches when "
                           "several processes share one GPU.")
    if code & 8:  # bit 3: the one-launch step met a flagged reference behind the batch's first n_flagged triples
        err.zero_()
        raise RuntimeError(f"{what}: n_flagged_dev does not describe the id / flag arrays of this step (they must be the "
                           "ones trs_epoch_flags_ordered wrote); the step is not exact.")
    if code != 0:
        raise IndexError(f"index out of range in self ({what}: an id is outside its embedding table; ids must be "
                         f"dense 0..n-1 as in the reference, dataset/dataset.py:30-31,268-269)")


class _PairScore(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, ids, with_neg, *params):
        dev = params[0].device
        err = torch.zeros(1, dtype=torch.int32, device=dev)
        B = ids["user"].shape[0]
        Bt, keep = ops.make_batch(ids["user"], ids["pos"], ids["neg"] if with_neg else None, ids.get("pos_meta"),
                                  ids.get("neg_meta") if with_neg else None, err)
        pos, neg = ops.score_forward(net.NET, net.tables(), Bt, B, dev, want_neg=with_neg)
        check_err_flag(err, "forward")
        ctx.net, ctx.ids, ctx.with_neg = net, ids, with_neg
        if not with_neg:
            neg = pos.new_zeros(0)
        return pos, neg

    @staticmethod
    def backward(ctx, gpos, gneg):
        net, ids, with_neg = ctx.net, ctx.ids, ctx.with_neg
        params = net.table_params()
        dev = params[0].device
        B = ids["user"].shape[0]
        D, M = params[0].shape[1], net.n_meta_tables()
        gpos = gpos.reshape(-1).contiguous().float()
        if with_neg:
            gneg = gneg.reshape(-1).contiguous().float()
            neg_ids, neg_meta = ids["neg"], ids.get("neg_meta")
        else:  # single pass: run the pair kernel with a zero-gradient copy of the pass as the negative
            gneg = torch.zeros_like(gpos)
            neg_ids, neg_meta = ids["pos"], ids.get("pos_meta")
        Bt, keep = ops.make_batch(ids["user"], ids["pos"], neg_ids, ids.get("pos_meta"), neg_meta, None)
        gr, gl = ops.score_backward(net.NET, net.tables(), Bt, B, D, M, dev, gpos, gneg)

        def coo(idx, vals, p):
            return torch.sparse_coo_tensor(idx.reshape(1, -1).long(), vals, size=p.shape)

        if with_neg:
            item_idx = torch.cat([ids["pos"], ids["neg"]])
            item_rows, item_lin = gr[1:3].reshape(2 * B, D), gl[1:3].reshape(2 * B, 1)
        else:
            item_idx, item_rows, item_lin = ids["pos"], gr[1], gl[1].reshape(B, 1)
        grads = [coo(ids["user"], gr[0], params[0]), coo(item_idx, item_rows, params[1]),
                 coo(ids["user"], gl[0].reshape(B, 1), params[2]), coo(item_idx, item_lin, params[3])]
        for lin in (False, True):
            if lin and not net.META_LIN_NAME:
                break
            for m in range(M):
                p = params[4 + (M if lin else 0) + m]
                if with_neg:
                    idx = torch.cat([ids["pos_meta"][:, m], ids["neg_meta"][:, m]])
                    sl = slice(3 + 2 * m, 5 + 2 * m)
                    vals = gl[sl].reshape(2 * B, 1) if lin else gr[sl].reshape(2 * B, D)
                else:
                    idx = ids["pos_meta"][:, m]
                    vals = gl[3 + 2 * m].reshape(B, 1) if lin else gr[3 + 2 * m]
                grads.append(coo(idx, vals, p))
        return (None, None, None, *grads)
