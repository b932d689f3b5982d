"""Scratch: bf16-resident GEMM rates (NT: fwd / dgrad with W^T image; TN: wgrad) at the c5 / c3 MLP shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torchrecsys_amd import ops
dev = "cuda:0"
R = int(os.environ.get("ROWS", "65536"))
def bench(name, tn, A, B, n=10):
    for _ in range(2): ops.gemm_bf16in(tn, A, B)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): out = ops.gemm_bf16in(tn, A, B)
    e1.record(); torch.cuda.synchronize()
    M, N = out.shape; K = A.shape[0] if tn else A.shape[1]
    ms = e0.elapsed_time(e1) / n
    print(f"{name:34s} M={M:7d} N={N:5d} K={K:7d}  {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:7.1f} TFLOP/s")
for (din, h) in ((1280, 1024), (1024, 512), (512, 256), (384, 512), (256, 128)):
    x = torch.randn(R, din, device=dev).bfloat16(); W = torch.randn(h, din, device=dev).bfloat16()
    dy = torch.randn(R, h, device=dev).bfloat16(); Wt = W.t().contiguous()
    bench(f"fwd  y=x W^T  ({din}->{h})", False, x, W)
    bench(f"dgrad dx=dy W ({h}->{din})", False, dy, Wt)
    bench(f"wgrad dW=dy^T x", True, dy, x)
x = torch.randn(8192, 8192, device=dev).bfloat16(); bench("square 8192 NT", False, x, x); bench("square 8192 TN", True, x, x)
