"""Scratch: c2 step time per optimiser (SGD fast path vs the generic staged path with SparseAdam / Adagrad)."""
import os, sys, time, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from torchrecsys_amd.model import TorchRecSys
dev = torch.device("cuda", 0)
cfg = bench.CONFIGS["c2"]
users, items = bench.synth_stream(cfg["n_users"], cfg["n_items"], 20_000_000, dev, seed=1000)
for name in sys.argv[1:] or ["sgd", "sparse_adam", "adagrad"]:
    with contextlib.redirect_stdout(io.StringIO()):
        torch.manual_seed(7)
        model = TorchRecSys.from_tensors(users, items, n_users=cfg["n_users"], n_items=cfg["n_items"], n_factors=64,
                                         net_type="fm", split_ratio=0.8, dynamic_neg_sampling=True, rng="device", seed=7)
    opt = {"sgd": lambda p: torch.optim.SGD(p, lr=1e-2), "sparse_adam": lambda p: torch.optim.SparseAdam(list(p), lr=1e-3),
           "adagrad": lambda p: torch.optim.Adagrad(p, lr=1e-2)}[name](model.parameters())
    r = model.make_runner(opt, cfg["B"]); model.net.train(); r.begin_epoch()
    r.run_steps(16); torch.cuda.synchronize()
    K = 128; t0 = time.perf_counter(); r.run_steps(K); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    print(f"{name:12s} {dt*1e6:8.1f} us/step  {2*cfg['B']/dt/1e9:.3f} G interactions/s")
    del model, opt, r
