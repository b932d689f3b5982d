"""Scratch: standalone time of one presort slice at the c2 shape (512 batches of 65536 triples from the resident stream)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from torchrecsys_amd import ops
dev = torch.device("cuda", 0)
cfg = bench.CONFIGS["c2"]
users, items = bench.synth_stream(cfg["n_users"], cfg["n_items"], cfg["n"], dev, seed=1000)
ui = ops.interleave_stream(users.to(torch.int32), items.to(torch.int32))
err = torch.zeros(1, dtype=torch.int32, device=dev)
B, nb = cfg["B"], 512
ps = ops.EpochPresort(nb, B, cfg["n_users"], cfg["n_items"], dev)
for _ in range(2):
    ps.run(ui, None, 0x1234, 77, 0, err)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(5):
    ps.run(ui, None, 0x1234 + i, 77 + i, 0, err)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"presort of {nb} batches x {B}: {ms:.3f} ms = {1e3*ms/nb:.2f} us per step; err {err.item()}")
