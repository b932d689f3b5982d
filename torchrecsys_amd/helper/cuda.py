# -*- coding: utf-8 -*-
"""Device placement helpers (reference helper/cuda.py:3-16).

On this framework the compute device is always the MI355X: `gpu(x, True)` moves to it, `gpu(x, False)` leaves the
object where it is (the reference's CPU mode has no equivalent here — scorers raise if asked to compute on CPU)."""


def gpu(tensor, gpu=False):
    return tensor.cuda() if gpu else tensor


def cpu(tensor):
    return tensor.cpu() if tensor.is_cuda else tensor
