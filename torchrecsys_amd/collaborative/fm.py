# -*- coding: utf-8 -*-
"""Factorisation-machine scorer over the fields {user, item, metadata...} with a sigmoid output
(reference collaborative/fm.py:21-101), on the fused HIP scoring kernels (sum-of-squares pairwise term reduced with
wavefront shuffles, no (B,F,D) tensor materialised)."""
import torch

from ..embeddings.init_embeddings import ScaledEmbedding
from ._scorer import SparseScorer


class FM(SparseScorer):
    """score = sigmoid(sum_f w_f + 0.5 * sum_d[(sum_f v_fd)^2 - sum_f v_fd^2])  -> (B,)."""
    NET = "fm"
    LIN_NAMES = ("linear_user", "linear_item")
    META_LIN_NAME = "linear_metadata"

    def __init__(self, n_users, n_items, n_metadata, n_factors, use_metadata=True, use_cuda=False):
        super().__init__()
        self.n_users, self.n_items, self.n_metadata = n_users, n_items, n_metadata
        self.n_factors, self.use_metadata, self.use_cuda = n_factors, use_metadata, use_cuda
        self.n_input = n_users + n_items
        # creation order = RNG order of the reference (fm.py:42-56)
        self.user = ScaledEmbedding(n_users, n_factors, sparse=True)
        self.item = ScaledEmbedding(n_items, n_factors, sparse=True)
        self.linear_user = ScaledEmbedding(n_users, 1, sparse=True)
        self.linear_item = ScaledEmbedding(n_items, 1, sparse=True)
        if use_metadata:
            self.n_distinct_metadata = len(n_metadata.keys())
            self.metadata = torch.nn.ModuleList(
                [ScaledEmbedding(size, n_factors, sparse=True) for _, size in n_metadata.items()])
            self.linear_metadata = torch.nn.ModuleList(
                [ScaledEmbedding(size, 1, sparse=True) for _, size in n_metadata.items()])
