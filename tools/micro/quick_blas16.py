"""Scratch: what the vendor library (torch.matmul -> hipBLASLt) reaches on the c5 / c3 GEMM shapes, beside trs_gemm_bf16in."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
dev = "cuda:0"
def bench(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N, K) in ((65536, 1024, 1280), (65536, 512, 1024), (65536, 256, 512), (65536, 1280, 1024), (8192, 8192, 8192)):
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16); w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    us = bench(lambda: torch.matmul(a, w.t()))
    print(f"NT bf16 {M}x{N}x{K}: {us:8.1f} us {2*M*N*K/us/1e6:8.1f} TFLOP/s", flush=True)
    # wgrad: dW (N,K) = dY^T (N,M) @ X (M,K)
    dy = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
    us = bench(lambda: torch.matmul(dy.t(), a))
    print(f"TN bf16 {N}x{K}x{M}: {us:8.1f} us {2*M*N*K/us/1e6:8.1f} TFLOP/s", flush=True)
for (M, N, K) in ((131072, 512, 384), (131072, 256, 512), (131072, 384, 512)):
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev)
    us = bench(lambda: torch.matmul(a, w.t()))
    print(f"NT fp32 {M}x{N}x{K}: {us:8.1f} us {2*M*N*K/us/1e6:8.1f} TFLOP/s", flush=True)
    dy = torch.randn(M, N, device=dev)
    us = bench(lambda: torch.matmul(dy.t(), a))
    print(f"TN fp32 {N}x{K}x{M}: {us:8.1f} us {2*M*N*K/us/1e6:8.1f} TFLOP/s", flush=True)
