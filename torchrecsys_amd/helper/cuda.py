# -*- coding: utf-8 -*-
"""Device placement policy of the path (the reference's helper/cuda.py:3-16 `.cuda()` / `.cpu()` shims).

In this framework the compute device is always the MI355X PyTorch-ROCm exposes as `cuda`; the flag only decides
whether an object is moved there now.  `gpu(x, False)` returns `x` untouched (batches stay host tensors until a scorer
uploads them; scorers refuse to compute on host tensors' device)."""
import torch


def _accelerator():
    if not torch.cuda.is_available():
        raise RuntimeError("no MI355X visible to PyTorch-ROCm: torchrecsys_amd has no CPU compute path")
    return torch.device("cuda", torch.cuda.current_device())


def gpu(tensor, gpu=False):
    """Move `tensor` (or a module) to the accelerator when `gpu` is true."""
    return tensor.to(_accelerator()) if gpu else tensor


def cpu(tensor):
    """Host copy of an accelerator tensor (identity for host tensors)."""
    return tensor.to("cpu") if getattr(tensor, "is_cuda", False) else tensor
