"""The north-star pass (trs_score_forward -> pair_scores_kernel) with FRESH triples per launch (nothing served from a
previous launch's cache footprint), at the c4 / c2 table shapes.  Tuning knobs are read by the library from the
environment (TRS_PASS_ITERS, TRS_PASS_GRID_CAP): run once per setting.
usage: python tools/pass_sweep.py [c4|c2] [B ...]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchrecsys_amd import _lib, ops

dev = "cuda:0"
cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
nu, ni, D = {"c4": (10_000_000, 1_000_000, 128), "c2": (1_000_000, 100_000, 64)}[cfg]
Bs = [int(b) for b in sys.argv[2:]] or [32_768, 262_144]
g = torch.Generator(device=dev)
g.manual_seed(0)
t = [torch.randn(nu, D, device=dev) * 0.1, torch.randn(ni, D, device=dev) * 0.1, torch.randn(nu, 1, device=dev),
     torch.randn(ni, 1, device=dev)]
T, keep = ops.make_tables(*t)
err = torch.zeros(1, dtype=torch.int32, device=dev)
lib = _lib.load()
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("TRS_"))
for B in Bs:
    groups, per = 8, 10
    n = (groups * per + 3) * B
    u = torch.randint(0, nu, (n,), device=dev, generator=g, dtype=torch.int32)
    p = torch.randint(0, ni, (n,), device=dev, generator=g, dtype=torch.int32)
    q = torch.randint(0, ni, (n,), device=dev, generator=g, dtype=torch.int32)
    ps, ns = torch.empty(B, device=dev), torch.empty(B, device=dev)
    bts = [ops.make_batch(u[k * B:(k + 1) * B], p[k * B:(k + 1) * B], q[k * B:(k + 1) * B], None, None, err)
           for k in range(groups * per + 3)]

    def launch(k):
        _lib.check(lib.trs_score_forward(1, C.byref(T), C.byref(bts[k][0]), ops.ptr(ps), ops.ptr(ns), ops._stream()))
    for k in range(3):
        launch(k)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(groups)]
    k = 3
    for e0, e1 in evs:
        e0.record()
        for _ in range(per):
            launch(k)
            k += 1
        e1.record()
    torch.cuda.synchronize()
    us = sorted(1e3 * e0.elapsed_time(e1) / per for e0, e1 in evs)
    byt = (16 + 3 * (4 * D + 4) + 8) * B
    mean = sum(us) / len(us)
    print(f"{cfg} D={D} B={B} [{tag}]: mean {mean:6.2f} us (min {us[0]:.2f} max {us[-1]:.2f})  "
          f"{byt / mean / 1e3:7.1f} GB/s = {byt / mean / 1e3 / 8000:.3f} of 8 TB/s", flush=True)
