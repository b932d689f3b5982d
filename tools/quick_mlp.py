"""Scratch: MLP c3-shaped step timing (B=65536, D=128, 1 metadata column of 10K categories, hidden [512,256,128])."""
import os, sys, time, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchrecsys_amd.model import TorchRecSys
dev = "cuda:0"
NU, NI, NC, D, B, N = 1_000_000, 100_000, 10_000, 128, 65536, 4_000_000
g = torch.Generator(device=dev); g.manual_seed(0)
users = torch.randint(0, NU, (N,), device=dev, dtype=torch.int32, generator=g)
items = torch.randint(0, NI, (N,), device=dev, dtype=torch.int32, generator=g)
users[:NU] = torch.arange(NU, device=dev, dtype=torch.int32); items[:NI] = torch.arange(NI, device=dev, dtype=torch.int32)
meta = torch.randint(0, NC, (NI, 1), device=dev, dtype=torch.int32, generator=g); meta[:NC, 0] = torch.arange(NC, device=dev, dtype=torch.int32)
with contextlib.redirect_stdout(io.StringIO()):
    model = TorchRecSys.from_tensors(users, items, n_users=NU, n_items=NI, item_metadata=meta, n_factors=D, net_type="mlp",
                                     hidden_layers=[512, 256, 128], dynamic_neg_sampling=True, rng="device")
opt = torch.optim.SGD(model.parameters(), lr=1e-2)
r = model.make_runner(opt, B); model.net.train(); r.begin_epoch()
r.run_steps(3); torch.cuda.synchronize()
K = 10; t0 = time.perf_counter(); r.run_steps(K); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
P = 384*512 + 512*256 + 256*128 + 128
print(f"MLP c3 step {dt*1e3:.2f} ms  interactions/s {2*B/dt/1e6:.2f} M  GEMM TFLOP/s (12P per triple) {12*P*B/dt/1e12:.1f}")
print("loss", r.end_epoch())
