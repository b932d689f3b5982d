# -*- coding: utf-8 -*-
"""MLP scorer: concat(user, item, metadata...) embeddings -> [Linear -> BatchNorm1d -> ReLU] x L -> Linear(.,1)
(reference collaborative/mlp.py:9-115), on the HIP kernels: embedding gather-concat, fp32-MFMA GEMMs, per-pass
BatchNorm statistics, fused BN+ReLU forward/backward (torchrecsys_amd/mlp_engine.py)."""
from typing import List

import torch

from .. import ops
from ..embeddings.init_embeddings import ScaledEmbedding
from ._scorer import _NO_GPU, as_id_matrix, check_err_flag


class MLP(torch.nn.Module):
    PREDICT_CHUNK = 262_144  # items scored per pass by score_all_items (eval mode: the result does not depend on it)

    def __init__(self, n_users, n_items, n_metadata, n_factors, use_metadata=True, use_batch_norm: bool = True,
                 hidden_layers: List[int] = None, use_cuda=False, use_bf16=False):
        super().__init__()
        self.n_users, self.n_items, self.n_metadata = n_users, n_items, n_metadata
        self.n_factors, self.use_metadata, self.use_batch_norm = n_factors, use_metadata, use_batch_norm
        self.hidden_layers = hidden_layers if hidden_layers is not None else [1024, 128]
        self.use_cuda = use_cuda
        self.use_bf16 = use_bf16
        self.input_shape = self.n_factors * 2
        if use_metadata:
            self.n_distinct_metadata = len(self.n_metadata.keys())
            self.input_shape += self.n_factors * self.n_distinct_metadata
        # creation order = RNG order of the reference (mlp.py:66-85)
        self.user = ScaledEmbedding(self.n_users, self.n_factors, sparse=True)
        self.item = ScaledEmbedding(self.n_items, self.n_factors, sparse=True)
        if use_metadata:
            self.metadata_embeddings = torch.nn.ModuleList(
                [ScaledEmbedding(size, self.n_factors, sparse=True) for _, size in self.n_metadata.items()])
        self.fcs = torch.nn.ModuleList()
        if self.use_batch_norm:
            self.bns = torch.nn.ModuleList()
        cur = self.input_shape
        for layer_size in self.hidden_layers:
            self.fcs.append(torch.nn.Linear(cur, layer_size))
            if self.use_batch_norm:
                self.bns.append(torch.nn.BatchNorm1d(layer_size))
            cur = layer_size
        self.output_layer = torch.nn.Linear(cur, 1)
        from ..mlp_engine import MLPCompute
        self.compute = MLPCompute(self)

    # ------------------------------------------------------------------------------------------ parameter views
    def n_meta_tables(self):
        return len(self.metadata_embeddings) if self.use_metadata else 0

    def embedding_params(self):
        ps = [self.user.weight, self.item.weight]
        if self.use_metadata:
            ps += [l.weight for l in self.metadata_embeddings]
        return ps

    def dense_params(self):
        ps = []
        for l, fc in enumerate(self.fcs):
            ps += [fc.weight, fc.bias]
            if self.use_batch_norm:
                ps += [self.bns[l].weight, self.bns[l].bias]
        return ps + [self.output_layer.weight, self.output_layer.bias]

    def all_params(self):
        return self.embedding_params() + self.dense_params()

    def tables(self):
        ps = self.embedding_params()
        if ps[0].device.type != "cuda":
            raise RuntimeError(_NO_GPU.format(dev=ps[0].device))
        key = tuple(p.data_ptr() for p in ps)
        cache = getattr(self, "_tables_cache", None)
        if cache is None or cache[0] != key:
            T, keep = ops.make_tables(ps[0].data, ps[1].data, None, None, [p.data for p in ps[2:]], [])
            cache = (key, T, keep)
            self._tables_cache = cache
        return cache[1]

    def _err_flag(self):
        dev = self.user.weight.device
        e = getattr(self, "_err", None)
        if e is None or e.device != dev:
            e = torch.zeros(1, dtype=torch.int32, device=dev)
            self._err = e
        return e

    def _check_err(self, what):
        check_err_flag(self._err_flag(), what)

    def device_ids(self, batch, user_key, item_key, metadata_key, neg_item_key=None, neg_metadata_key=None):
        dev = self.user.weight.device
        if dev.type != "cuda":
            raise RuntimeError(_NO_GPU.format(dev=dev))
        M = self.n_meta_tables()

        def mv(t):
            return t.long().to(dev, non_blocking=True).contiguous()

        ids = {"user": mv(batch[user_key]), "pos": mv(batch[item_key])}
        if neg_item_key:
            ids["neg"] = mv(batch[neg_item_key])
        if M:
            ids["pos_meta"] = mv(as_id_matrix(batch.get(metadata_key) if metadata_key else None, M))
            if neg_item_key:
                ids["neg_meta"] = mv(as_id_matrix(batch.get(neg_metadata_key), M))
        return ids

    # ------------------------------------------------------------------------------------------ forward
    def forward(self, batch, user_key, item_key, metadata_key=None):
        """One scoring pass -> (B, 1) (reference mlp.py:88-115).  Train mode normalises with this call's batch
        statistics and updates the running statistics once."""
        from ..mlp_engine import _MLPScore
        ids = self.device_ids(batch, user_key, item_key, metadata_key)
        return _MLPScore.apply(self, ids, 1, *self.all_params()).reshape(-1, 1)

    def forward_pair(self, batch, user_key="user_id", pos_key="pos_item_id", neg_key="neg_item_id",
                     pos_meta_key="pos_metadata_id", neg_meta_key="neg_metadata_id"):
        """Positive and negative pass of reference model.py:171-185 stacked into one 2B-row pipeline (BatchNorm
        statistics stay per pass; running statistics are updated positive pass first)."""
        from ..mlp_engine import _MLPScore
        ids = self.device_ids(batch, user_key, pos_key, pos_meta_key, neg_key, neg_meta_key)
        out = _MLPScore.apply(self, ids, 2, *self.all_params())
        B = ids["user"].shape[0]
        return out[:B].reshape(-1, 1), out[B:].reshape(-1, 1)

    # ------------------------------------------------------------------------------------------ inference
    def score_ids(self, ids):
        out, _ = self.compute.forward(ids, 2, self.training)
        self._check_err("evaluate")
        B = ids["user"].shape[0]
        return out[:B], out[B:]

    def score_all_items(self, user_id, item_meta_dev=None):
        if not 0 <= user_id < self.n_users:
            raise IndexError(f"index out of range in self (user_id {user_id} outside [0, {self.n_users}))")
        dev = self.user.weight.device
        # item chunks (the reference's prediction_batch_size loop, model.py:384): a whole-catalogue pass would keep every
        # layer's activations of all n_items rows alive at once (c5: 1M items x [1280, 1024, 512, 256] floats > 10 GB)
        out = torch.empty(self.n_items, dtype=torch.float32, device=dev)
        chunk = self.PREDICT_CHUNK
        if self.training and self.use_batch_norm:
            chunk = self.n_items  # train-mode BatchNorm normalises with the statistics of the rows it is given
        for s in range(0, self.n_items, chunk):
            e = min(s + chunk, self.n_items)
            ids = {"user": torch.full((e - s,), user_id, dtype=torch.int64, device=dev),
                   "pos": torch.arange(s, e, dtype=torch.int64, device=dev)}
            if self.n_meta_tables():
                ids["pos_meta"] = item_meta_dev[s:e].long().contiguous()
            part, ctx = self.compute.forward(ids, 1, self.training)
            out[s:e] = part
            del ctx
        self._check_err("predict")
        return out
