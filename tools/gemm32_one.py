"""Scratch: the fp32 GEMM shapes of c3's first layer, a few launches (for rocprofv3 --pmc)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchrecsys_amd import ops
dev = "cuda:0"
R, din, h = 131072, 384, 512
x = torch.randn(R, din, device=dev); W = torch.randn(h, din, device=dev); dy = torch.randn(R, h, device=dev)
for _ in range(4):
    ops.gemm(False, True, x, W)
    ops.gemm(False, False, dy, W)
    ops.gemm(True, False, dy, x)
torch.cuda.synchronize()
