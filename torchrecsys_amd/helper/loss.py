# -*- coding: utf-8 -*-
"""hinge_loss (reference helper/loss.py:5-9) on the HIP hinge kernels, differentiable through torch.autograd."""
import torch

from .. import ops


class _Pair(torch.autograd.Function):
    @staticmethod
    def forward(ctx, positive, negative, kind):
        p = positive.detach().reshape(-1).contiguous().float()
        n = negative.detach().reshape(-1).contiguous().float()
        acc = torch.zeros(1, dtype=torch.float32, device=p.device)
        ops.hinge_auc(p, n, acc, None, loss=kind)
        ctx.save_for_backward(p, n)
        ctx.kind = kind
        ctx.shapes = (positive.shape, negative.shape)
        return (acc / max(p.numel(), 1)).reshape(())

    @staticmethod
    def backward(ctx, g):
        p, n = ctx.saved_tensors
        gp, gn = ops.hinge_backward(p, n, loss=ctx.kind)
        return (gp * g).reshape(ctx.shapes[0]), (gn * g).reshape(ctx.shapes[1]), None


def hinge_loss(positive, negative):
    """mean(clamp(negative - positive + 1, min=0))."""
    return _Pair.apply(positive, negative, 0)


def bpr_loss(positive, negative):
    """mean(-log sigmoid(positive - negative)) — Bayesian personalised ranking; not in the reference (its helper/loss.py
    holds hinge_loss only), named by BASELINE.json's north_star beside hinge."""
    return _Pair.apply(positive, negative, 1)
