"""Scratch: one bf16-resident GEMM shape, a few launches (for rocprofv3 --pmc)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchrecsys_amd import ops
dev = "cuda:0"
R, din, h = 65536, 1280, 1024
x = torch.randn(R, din, device=dev).bfloat16(); W = torch.randn(h, din, device=dev).bfloat16()
dy = torch.randn(R, h, device=dev).bfloat16()
for _ in range(4):
    ops.gemm_bf16in(False, x, W)
    ops.gemm_bf16in(True, dy, x)
torch.cuda.synchronize()
