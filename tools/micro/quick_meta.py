"""Scratch: FM / Linear step time WITH one metadata column at the c2 shape (generic staged path)."""
import os, sys, time, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from torchrecsys_amd.model import TorchRecSys
dev = torch.device("cuda", 0)
cfg = bench.CONFIGS["c2"]
users, items = bench.synth_stream(cfg["n_users"], cfg["n_items"], 20_000_000, dev, seed=1000)
g = torch.Generator(device=dev); g.manual_seed(5)
meta = torch.randint(0, 10_000, (cfg["n_items"], 1), device=dev, dtype=torch.int32, generator=g)
meta[:10_000, 0] = torch.arange(10_000, device=dev, dtype=torch.int32)
for net in ("fm", "linear"):
    with contextlib.redirect_stdout(io.StringIO()):
        torch.manual_seed(7)
        model = TorchRecSys.from_tensors(users, items, n_users=cfg["n_users"], n_items=cfg["n_items"], item_metadata=meta,
                                         n_factors=64, net_type=net, split_ratio=0.8, dynamic_neg_sampling=True, rng="device", seed=7)
    opt = torch.optim.SGD(model.parameters(), lr=1e-2)
    r = model.make_runner(opt, cfg["B"]); model.net.train(); r.begin_epoch()
    r.run_steps(8); torch.cuda.synchronize()
    K = 64; t0 = time.perf_counter(); r.run_steps(K); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    print(f"{net:7s} M=1  {dt*1e6:8.1f} us/step  {2*cfg['B']/dt/1e9:.3f} G interactions/s")
