"""Scratch: fp32 MFMA GEMM rates per layout / MLP shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchrecsys_amd import ops
dev = "cuda:0"
R = 131072
BF = bool(int(os.environ.get("BF16", "0")))
def bench(name, tA, tB, A, B, n=10):
    _g = ops.gemm
    ops_gemm = lambda *a: _g(*a, bf16=BF)
    for _ in range(2): ops_gemm(tA, tB, A, B)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): out = ops_gemm(tA, tB, A, B)
    e1.record(); torch.cuda.synchronize()
    M, N = out.shape; K = A.shape[0] if tA else A.shape[1]
    ms = e0.elapsed_time(e1) / n
    print(f"{name:34s} M={M:7d} N={N:5d} K={K:7d}  {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:6.1f} TFLOP/s")
for (din, h) in ((384, 512), (512, 256), (256, 128), (1280, 1024), (1024, 512)):
    x = torch.randn(R, din, device=dev); W = torch.randn(h, din, device=dev); dy = torch.randn(R, h, device=dev)
    bench(f"fwd  y=x W^T  ({din}->{h})", False, True, x, W)
    bench(f"dgrad dx=dy W ({h}->{din})", False, False, dy, W)
    bench(f"wgrad dW=dy^T x", True, False, dy, x)
x = torch.randn(4096, 4096, device=dev); bench("square 4096 NT", False, True, x, x); bench("square 4096 NN", False, False, x, x)
