"""Scratch: fp32 NT GEMM rates at the c3 MLP shapes (131 072 rows): forward y = x W^T and input gradient dx = dy W through a
transposed W.  TRS_GEMM32_NO_GLDS=1 times the register-staged 128 x 128 kernel instead of the LDS-DMA one."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torchrecsys_amd import ops
dev = "cuda:0"
R = int(os.environ.get("ROWS", "131072"))
def bench(name, A, B, n=10):
    for _ in range(2): ops.gemm(False, True, A, B)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): out = ops.gemm(False, True, A, B)
    e1.record(); torch.cuda.synchronize()
    M, N = out.shape; K = A.shape[1]
    ms = e0.elapsed_time(e1) / n
    print(f"{name:28s} M={M:7d} N={N:5d} K={K:5d}  {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:7.1f} TFLOP/s")
for (din, h) in ((384, 512), (512, 256), (256, 128)):
    x = torch.randn(R, din, device=dev); W = torch.randn(h, din, device=dev)
    dy = torch.randn(R, h, device=dev); Wt = W.t().contiguous()
    bench(f"fwd   ({din}->{h})", x, W)
    bench(f"dgrad ({h}->{din})", dy, Wt)
