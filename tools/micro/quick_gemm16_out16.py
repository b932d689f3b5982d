"""Scratch: trs_gemm_bf16in with bf16 OUTPUT (what the MLP step's forward / inner dgrad GEMMs write) at the c5 shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torchrecsys_amd import ops
dev = "cuda:0"
R = 65536
def bench(name, A, B, n=20, **kw):
    for _ in range(3): ops.gemm_bf16in(False, A, B, **kw)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): out = ops.gemm_bf16in(False, A, B, **kw)
    e1.record(); torch.cuda.synchronize()
    M, N = out.shape; K = A.shape[1]
    us = e0.elapsed_time(e1) / n * 1e3
    print(f"{name:40s} {M}x{N}x{K}: {us:8.1f} us {2*M*N*K/us/1e6:8.1f} TFLOP/s", flush=True)
for (din, h) in ((1280, 1024), (1024, 512), (512, 256)):
    x = torch.randn(R, din, device=dev).bfloat16(); W = torch.randn(h, din, device=dev).bfloat16()
    dy = torch.randn(R, h, device=dev).bfloat16(); Wt = W.t().contiguous()
    bias = torch.randn(h, device=dev)
    part = torch.empty((R // 128, 2, h), device=dev)
    bench(f"fwd {din}->{h} bf16 out + bias + BN partials", x, W, out_bf16=True, bias=bias, bn_part=part)
    bench(f"dgrad {h}->{din} bf16 out", dy, Wt, out_bf16=True)
