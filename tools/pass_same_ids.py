"""The north-star pass by itself: fused pos+neg embedding gather + FM pairwise score (trs_score_forward) at the c2 and c4
table shapes, random ids; GB/s against the algorithmic 16 + R(4D+4) + 8 bytes per triple."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchrecsys_amd import ops
dev = "cuda:0"
SHAPES = ((1_000_000, 100_000, 64, 65_536), (10_000_000, 1_000_000, 128, 32_768), (10_000_000, 1_000_000, 128, 262_144))
if len(sys.argv) > 1:  # one shape only (PMC passes: one kernel shape per run)
    SHAPES = (SHAPES[int(sys.argv[1])],)
for (nu, ni, D, B) in SHAPES:
    g = torch.Generator(device=dev); g.manual_seed(0)
    t = [torch.randn(nu, D, device=dev) * 0.1, torch.randn(ni, D, device=dev) * 0.1, torch.randn(nu, 1, device=dev), torch.randn(ni, 1, device=dev)]
    T, keep = ops.make_tables(*t)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    best = None
    for idt in (torch.int32,):
        u = torch.randint(0, nu, (B,), device=dev, generator=g).to(idt); p = torch.randint(0, ni, (B,), device=dev, generator=g).to(idt); n = torch.randint(0, ni, (B,), device=dev, generator=g).to(idt)
        Bt, kb = ops.make_batch(u, p, n, None, None, err)
        for _ in range(3): ops.score_forward("fm", T, Bt, B, dev)
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.score_forward("fm", T, Bt, B, dev)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        byt = (16 + 3 * (4 * D + 4) + 8) * B
        print(f"{nu/1e6:.0f}M x {ni/1e6:.1f}M  D={D} B={B}: {us:7.1f} us  {byt/us/1e3:7.1f} GB/s  = {byt/us/1e3/8000:.2f} of the 8 TB/s peak")
    del t, T, keep
