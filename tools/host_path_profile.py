# -*- coding: utf-8 -*-
"""Where the host time of FitRunner.run_steps goes on the flag-mode (sparse regime) path: cProfile over many short calls
(a small model: the host path does not depend on the table sizes).  Usage: python tools/host_path_profile.py"""
import contextlib
import cProfile
import io
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchrecsys_amd.model import TorchRecSys  # noqa: E402

dev = "cuda:0"
g = torch.Generator(device=dev)
g.manual_seed(0)
n_users, n_items, n, B = 200_000, 400_000, 4_000_000, 4096
users = torch.randint(0, n_users, (n,), device=dev, dtype=torch.int32, generator=g)
items = torch.randint(0, n_items, (n,), device=dev, dtype=torch.int32, generator=g)
with contextlib.redirect_stdout(io.StringIO()):
    model = TorchRecSys.from_tensors(users, items, n_users=n_users, n_items=n_items, n_factors=64, net_type="fm",
                                     dynamic_neg_sampling=True, rng="device", seed=1)
from torchrecsys_amd.engine import SparseScorerTrainer  # noqa: E402
SparseScorerTrainer.SLICE_BATCHES = 16
opt = torch.optim.SGD(model.parameters(), lr=1e-2)
r = model.make_runner(opt, B)
model.net.train()
r.begin_epoch()
r.run_steps(5)
torch.cuda.synchronize()
# the bench's short window: sync, then 20 steps
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r.run_steps(20)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"window {rep}: enqueue {1e6 * (t1 - t0):.1f} us, synced {1e6 * (t2 - t0):.1f} us")
pr = cProfile.Profile()
pr.enable()
for _ in range(30):
    r.run_steps(20)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(35)
print(s.getvalue())
