"""Scratch: the public fit() / evaluate() / predict() API for an MLP with one metadata column at a c3-like shape (device RNG):
wall time per epoch, losses, AUC — the fused embedding update, LDS-DMA GEMMs and BatchNorm kernels inside fit()."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from torchrecsys_amd.model import TorchRecSys
dev = torch.device("cuda", 0)
NU, NI, N, B, D = 1_000_000, 100_000, 20_000_000, 65_536, 128
amp = len(sys.argv) > 1 and sys.argv[1] == "amp"
users, items = bench.synth_stream(NU, NI, N, dev, seed=1000)
g = torch.Generator(device=dev); g.manual_seed(3)
item_meta = torch.randint(0, 10_000, (NI, 1), device=dev, generator=g, dtype=torch.int32)
torch.manual_seed(7)
model = TorchRecSys.from_tensors(users, items, n_users=NU, n_items=NI, n_factors=D, net_type="mlp", split_ratio=0.8,
                                 dynamic_neg_sampling=True, rng="device", seed=7, item_metadata=item_meta,
                                 hidden_layers=[512, 256, 128], use_amp=amp)
opt = torch.optim.SGD(model.parameters(), lr=0.05)
t0 = time.perf_counter()
model.fit(optimizer=opt, epochs=1, batch_size=B)
torch.cuda.synchronize(); ta = time.perf_counter()
model.fit(optimizer=opt, epochs=2, batch_size=B)
torch.cuda.synchronize(); t1 = time.perf_counter()
steps = int(N * 0.8) // B
print(f"fit ({'bf16' if amp else 'fp32'}): first epoch {1e3*(ta-t0):.0f} ms; then {1e3*(t1-ta)/2:.0f} ms per epoch of {steps} steps = "
      f"{1e3*(t1-ta)/2/steps:.2f} ms per step, {2*N*0.8*2/(t1-ta)/1e6:.1f} M interactions/s")
model.evaluate(batch_size=B)
torch.cuda.synchronize(); print(f"evaluate: {1e3*(time.perf_counter()-t1):.0f} ms")
print("predict:", model.predict(user_id=0, top_k=5).tolist())
