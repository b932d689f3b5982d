"""What bit-exact index parity costs at c2 scale.  rng='reference' replays the reference's host RNG streams — the
torch.randperm shuffle (dataset/dataset.py:364-373) and the legacy-numpy sampler walk (:435-447) — so that every batch
holds exactly the reference's triples; the GPU then runs the same kernels as in the device-RNG mode.  This script times one
epoch of config c2 (FM, 1M users x 100K items, D=64, batch 65 536) through fit()'s runner in that mode and prints one JSON
line: host seconds per epoch for the permutation + sampler + upload, GPU seconds for the steps.
usage: python tools/parity_mode_cost.py [n_interactions (default 100_000_000)]"""
import contextlib
import io
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from torchrecsys_amd.helper.cuda import host_threads
from torchrecsys_amd.model import TorchRecSys

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
NU, NI, D, B = 1_000_000, 100_000, 64, 65_536
with host_threads():
    rs = np.random.default_rng(0)
    t0 = time.perf_counter()
    users = torch.from_numpy(np.concatenate([np.arange(NU), rs.integers(0, NU, n - NU)]))
    items = torch.from_numpy(np.concatenate([np.tile(np.arange(NI), NU // NI), rs.integers(0, NI, n - NU)]))
    t_gen = time.perf_counter() - t0
    np.random.seed(7)
    torch.manual_seed(7)
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        model = TorchRecSys.from_tensors(users, items, n_users=NU, n_items=NI, n_factors=D, net_type="fm",
                                         split_ratio=0.8, dynamic_neg_sampling=True, rng="reference")
    t_ingest = time.perf_counter() - t0
    opt = torch.optim.SGD(model.parameters(), lr=1e-2)
    runner = model.make_runner(opt, B)
    model.net.train()
    out = {"config": f"c2 shape, {n} interactions ({runner.n_train} train triples, {runner.num_batches} batches of {B}), "
                     f"rng='reference'", "ingest_split_s": t_ingest}
    for epoch in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        runner.begin_epoch()  # host: torch.randperm + gathers + the sampler walk over every batch; then the upload
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        runner.run_steps(runner.num_batches)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        loss = runner.end_epoch()
        out[f"epoch{epoch}"] = {"host_shuffle_sampler_upload_s": t1 - t0, "gpu_steps_s": t2 - t1, "loss": loss,
                                "interactions_per_s_gpu_only": 2.0 * runner.n_train / (t2 - t1),
                                "interactions_per_s_whole_epoch": 2.0 * runner.n_train / (t2 - t0)}
    out["note"] = ("fit() overlaps the NEXT epoch's host preparation with the current epoch's GPU steps "
                   "(FitRunner.prepare_next_epoch); here the two are timed back to back")
    print(json.dumps(out))
