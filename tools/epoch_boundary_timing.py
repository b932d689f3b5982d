"""Where the time goes around an epoch boundary at c2 (first steps under the next slice's presort, steady steps, tail)."""
import os, sys, time, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from torchrecsys_amd.model import TorchRecSys
dev = torch.device("cuda", 0)
cfg = bench.CONFIGS["c2"]
users, items = bench.synth_stream(cfg["n_users"], cfg["n_items"], cfg["n"], dev, seed=1000)
with contextlib.redirect_stdout(io.StringIO()):
    torch.manual_seed(7)
    model = TorchRecSys.from_tensors(users, items, n_users=cfg["n_users"], n_items=cfg["n_items"], n_factors=64, net_type="fm",
                                     split_ratio=0.8, dynamic_neg_sampling=True, rng="device", seed=7)
opt = torch.optim.SGD(model.parameters(), lr=1e-2)
B = cfg["B"]
r = model.make_runner(opt, B); model.net.train()
full = r.n_train // B
def T(): torch.cuda.synchronize(); return time.perf_counter()
r.begin_epoch(); r.run_steps(full); t0 = T()
for ep in range(3):
    a = time.perf_counter(); r.end_epoch(); b = T(); r.begin_epoch(); c = time.perf_counter()
    r.run_steps(64); d = T()
    r.run_steps(448); e = T()
    r.run_steps(512); f = T()
    r.run_steps(full - 1024); g = T()
    print(f"end_epoch {1e3*(b-a):.2f} ms  begin {1e3*(c-b):.2f}  first64 {1e3*(d-c):.2f} ({1e6*(d-c)/64:.1f} us/step)  next448 {1e6*(e-d)/448:.1f}  next512 {1e6*(f-e)/512:.1f}  tail{full-1024} {1e6*(g-f)/(full-1024):.1f} us/step")
