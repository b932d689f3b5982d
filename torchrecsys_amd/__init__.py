# -*- coding: utf-8 -*-
"""torchrecsys_amd — the MI355X (gfx950) native hot path of FrancescoI/torchrecsys behind the reference's own API.

    from torchrecsys_amd.model import TorchRecSys      # drop-in for torchrecsys.model.TorchRecSys

Python host code on PyTorch-ROCm (memory, streams, torch.distributed) -> ctypes -> libtrs_hip.so (include/trs.h):
hand-written HIP kernels for the fused positive+negative embedding gather, the Linear / FM / MLP scorers, hinge, the
sparse embedding-row optimisers, the negative sampler and the predict top-k.  There is no CPU fallback.
"""
__version__ = "0.1.0"
