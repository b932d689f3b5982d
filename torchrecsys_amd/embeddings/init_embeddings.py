# -*- coding: utf-8 -*-
"""Embedding tables of the scorers (reference embeddings/init_embeddings.py:5-50, 53-97).

The tables are plain nn.Embedding parameters (row-major (n, D) fp32) so that state_dicts, user-built optimisers and
`model.parameters()` behave exactly as with the reference; the HIP kernels read and update `weight` in place through
its device pointer.  Initialisation draws from torch's global CPU generator in the same order as the reference, so a
seeded construction yields bit-identical initial weights.
"""
import torch


class ScaledEmbedding(torch.nn.Embedding):
    """weight ~ N(0, (1/embedding_dim)^2)  (init_embeddings.py:44-50)."""

    def reset_parameters(self):
        self.weight.data.normal_(0, 1.0 / self.embedding_dim)
        if self.padding_idx is not None:
            self.weight.data[self.padding_idx].fill_(0)


class ZeroEmbedding(torch.nn.Embedding):
    """weight = 0, used for the Linear scorer's biases (init_embeddings.py:91-97)."""

    def reset_parameters(self):
        self.weight.data.zero_()
        if self.padding_idx is not None:
            self.weight.data[self.padding_idx].fill_(0)
