# -*- coding: utf-8 -*-
"""Optimiser updates as the reference gets them from torch.optim (model.py:198 `optimizer.step()`), restated on
dense-equivalent gradients + the set of touched rows (numpy fp32).  TEST INFRASTRUCTURE.

Third-party algorithm sources (torch 2.10.0): torch/optim/sgd.py (_single_tensor_sgd), torch/optim/_functional.py
(sparse_adam), torch/optim/adagrad.py (_single_tensor_adagrad, sparse branch), torch/optim/adam.py
(_single_tensor_adam).  SURVEY.md App. A.5.
"""
import math

import numpy as np

F32 = np.float32


def sgd_step(params, grads, lr):
    """torch.optim.SGD, momentum=0, weight_decay=0: p.add_(grad, alpha=-lr) (sparse or dense)."""
    for k, g in grads.items():
        params[k] -= F32(lr) * g


def sgd_momentum_step(params, grads, state, lr, momentum, dampening=0.0, nesterov=False):
    """torch.optim.SGD with momentum: buf = grad (first step) else buf*momentum + (1-dampening)*grad; p -= lr*buf.
    With sparse gradients the buffer is a sparse tensor whose support only grows, which is mathematically this dense
    form."""
    for k, g in grads.items():
        if k not in state:
            state[k] = g.copy()
        else:
            state[k] = (state[k] * F32(momentum) + F32(1.0 - dampening) * g).astype(F32)
        d = g + F32(momentum) * state[k] if nesterov else state[k]
        params[k] -= F32(lr) * d


def sparse_adam_rows(p, g, rows, exp_avg, exp_avg_sq, step, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.SparseAdam on the coalesced gradient restricted to `rows` (lazy: untouched rows and their moments
    do not move).  `step` is the 1-based step count of this parameter."""
    gv = g[rows]
    old_m, old_v = exp_avg[rows], exp_avg_sq[rows]
    dm = ((gv - old_m) * F32(1 - beta1)).astype(F32)
    dv = ((gv * gv - old_v) * F32(1 - beta2)).astype(F32)
    exp_avg[rows] = old_m + dm
    exp_avg_sq[rows] = old_v + dv
    numer = dm + old_m
    denom = np.sqrt(dv + old_v, dtype=F32) + F32(eps)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    step_size = F32(lr * math.sqrt(bc2) / bc1)
    p[rows] += -step_size * (numer / denom)


def adagrad_rows(p, g, rows, state_sum, step, lr, lr_decay=0.0, eps=1e-10):
    """torch.optim.Adagrad sparse branch on coalesced rows: sum += g^2; p -= clr * g / (sqrt(sum) + eps)."""
    clr = F32(lr / (1 + (step - 1) * lr_decay))
    gv = g[rows]
    state_sum[rows] += gv * gv
    std = np.sqrt(state_sum[rows], dtype=F32) + F32(eps)
    p[rows] += -clr * (gv / std)


def adam_dense(p, g, exp_avg, exp_avg_sq, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    """torch.optim.Adam (_single_tensor_adam, amsgrad=False, maximize=False) for the dense MLP parameters."""
    if weight_decay != 0:
        g = g + F32(weight_decay) * p
    exp_avg[...] = exp_avg + (g - exp_avg) * F32(1 - beta1)  # lerp_
    exp_avg_sq[...] = exp_avg_sq * F32(beta2) + F32(1 - beta2) * g * g
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    step_size = lr / bc1
    denom = (np.sqrt(exp_avg_sq, dtype=F32) / F32(math.sqrt(bc2))) + F32(eps)
    p -= F32(step_size) * (exp_avg / denom)
