"""Scratch: the public fit() API at the c2 shape for a few epochs (device RNG): wall time per epoch, losses, AUC."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from torchrecsys_amd.model import TorchRecSys
dev = torch.device("cuda", 0)
cfg = bench.CONFIGS["c2"]
users, items = bench.synth_stream(cfg["n_users"], cfg["n_items"], cfg["n"], dev, seed=1000)
torch.manual_seed(7)
model = TorchRecSys.from_tensors(users, items, n_users=cfg["n_users"], n_items=cfg["n_items"], n_factors=64, net_type="fm",
                                 split_ratio=0.8, dynamic_neg_sampling=True, rng="device", seed=7)
opt = torch.optim.SGD(model.parameters(), lr=float(sys.argv[1]) if len(sys.argv) > 1 else 50.0)
t0 = time.perf_counter()
model.fit(optimizer=opt, epochs=2, batch_size=cfg["B"])
torch.cuda.synchronize(); ta = time.perf_counter()
model.fit(optimizer=opt, epochs=6, batch_size=cfg["B"])
torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"fit: first call {1e3*(ta-t0)/2:.1f} ms per epoch (2 epochs, includes allocations and the first presort); second call "
      f"{1e3*(t1-ta)/6:.1f} ms per epoch of {int(cfg['n']*0.8)//cfg['B']} steps = {2*cfg['n']*0.8*6/(t1-ta)/1e9:.2f} G interactions/s")
model.evaluate(batch_size=cfg["B"])
torch.cuda.synchronize(); print(f"evaluate: {1e3*(time.perf_counter()-t1):.1f} ms")
print("predict:", model.predict(user_id=0, top_k=5).tolist())
