# -*- coding: utf-8 -*-
"""hinge_loss (reference helper/loss.py:5-9) on the HIP hinge kernels, differentiable through torch.autograd."""
import torch

from .. import ops


class _Hinge(torch.autograd.Function):
    @staticmethod
    def forward(ctx, positive, negative):
        p = positive.detach().reshape(-1).contiguous().float()
        n = negative.detach().reshape(-1).contiguous().float()
        acc = torch.zeros(1, dtype=torch.float32, device=p.device)
        ops.hinge_auc(p, n, acc, None)
        ctx.save_for_backward(p, n)
        ctx.shapes = (positive.shape, negative.shape)
        return (acc / max(p.numel(), 1)).reshape(())

    @staticmethod
    def backward(ctx, g):
        p, n = ctx.saved_tensors
        gp, gn = ops.hinge_backward(p, n)
        return (gp * g).reshape(ctx.shapes[0]), (gn * g).reshape(ctx.shapes[1])


def hinge_loss(positive, negative):
    """mean(clamp(negative - positive + 1, min=0))."""
    return _Hinge.apply(positive, negative)
