# -*- coding: utf-8 -*-
"""Linear scorer: biased dot-product matrix factorisation with metadata summed into the item vector
(reference collaborative/linear.py:24-80), on the fused HIP scoring kernels."""
import torch

from ..embeddings.init_embeddings import ScaledEmbedding, ZeroEmbedding
from ._scorer import SparseScorer


class Linear(SparseScorer):
    """score = sum_d user_d * (item_d + sum_m meta_m,d) + user_bias + item_bias  -> (B, 1)."""
    NET = "linear"
    LIN_NAMES = ("user_bias", "item_bias")
    META_LIN_NAME = None

    def __init__(self, n_users, n_items, n_metadata, n_factors, use_metadata=True, use_cuda=False):
        super().__init__()
        self.n_users, self.n_items, self.n_metadata = n_users, n_items, n_metadata
        self.n_factors, self.use_metadata, self.use_cuda = n_factors, use_metadata, use_cuda
        # creation order = RNG order of the reference (linear.py:43-51)
        if use_metadata:
            self.metadata = torch.nn.ModuleList(
                [ScaledEmbedding(size, n_factors, sparse=True) for _, size in n_metadata.items()])
        self.user = ScaledEmbedding(n_users, n_factors, sparse=True)
        self.item = ScaledEmbedding(n_items, n_factors, sparse=True)
        self.user_bias = ZeroEmbedding(n_users, 1, sparse=True)
        self.item_bias = ZeroEmbedding(n_items, 1, sparse=True)
