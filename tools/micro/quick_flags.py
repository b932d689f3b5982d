"""trs_epoch_flags (the sparse regime's presort: ids + duplicate flags, one workgroup per batch) timed alone at the c4
shape: a full 512-batch slice and short slices.  usage: python tools/quick_flags.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torchrecsys_amd import ops

dev = "cuda:0"
NU, NI, B = 10_000_000, 1_000_000, 32_768
g = torch.Generator(device=dev)
g.manual_seed(0)
N = 80_000_000
ui = torch.stack([torch.randint(0, NU, (N,), device=dev, dtype=torch.int32, generator=g),
                  torch.randint(0, NI, (N,), device=dev, dtype=torch.int32, generator=g)], dim=1).contiguous()
err = torch.zeros(1, dtype=torch.int32, device=dev)
for nb in (512, 64, 16):
    ef = ops.EpochFlags(nb, B, NU, NI, dev)
    for _ in range(2):
        ef.run(ui, None, 0x1234567, 99, 0, err)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    for k, (e0, e1) in enumerate(evs):
        e0.record()
        ef.run(ui, None, 0x1234567 + k, 99, (k * nb * B) % (N - nb * B), err)
        e1.record()
    torch.cuda.synchronize()
    us = sorted(1e3 * e0.elapsed_time(e1) for e0, e1 in evs)
    print(f"c4 shape, {nb:4d} batches of {B}: {us[len(us) // 2]:8.1f} us per slice (min {us[0]:.1f})  = "
          f"{us[len(us) // 2] / nb:6.2f} us per step; users flagged {float(ef.user_dup.float().mean()):.4f}, "
          f"item refs flagged {float(ef.item_dup.float().mean()):.4f}", flush=True)
# ids given (no Feistel / Philox / stream reads): the bitmap passes alone
ef = ops.EpochFlags(16, B, NU, NI, dev)
given = [torch.randint(0, n, (16 * B,), device=dev, dtype=torch.int32, generator=g) for n in (NU, NI, NI)]
for _ in range(2):
    ef.run(None, None, 0, 0, 0, err, given_ids=given)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    check = ops._lib.load().trs_epoch_flags(None, None, 0, 0, 0, 0, 16, B, NU, NI, ops.ptr(ef.ids[0]), ops.ptr(ef.ids[1]),
                                            ops.ptr(ef.ids[2]), ops.ptr(ef.user_dup), ops.ptr(ef.item_dup), ops.ptr(err),
                                            None, ops._stream())
e1.record()
torch.cuda.synchronize()
print(f"given ids, 16 batches: {1e3 * e0.elapsed_time(e1) / 5:8.1f} us per slice (bitmap passes only)")
assert err.item() == 0
