# -*- coding: utf-8 -*-
"""Data-parallel fit() with two processes sharing the one GPU of the test box (gloo moves the few collectives through
the host; on a real node the backend is nccl = RCCL over xGMI, one GPU per rank)."""
import contextlib
import io
import os
import socket

import numpy as np
import pandas as pd
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, net_type, n, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from torchrecsys_amd.model import TorchRecSys
        rs = np.random.RandomState(0)
        n_u, n_i = 200, 60
        df = pd.DataFrame({"user": np.concatenate([np.arange(n_u), rs.randint(0, n_u, n - n_u)]),
                           "item": np.concatenate([np.arange(n_i), rs.randint(0, n_i, n - n_i)])})
        torch.manual_seed(100 + rank)  # different local init: the broadcast must make the replicas equal
        np.random.seed(5)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            kw = {"hidden_layers": [32, 16]} if net_type == "mlp" else {}
            model = TorchRecSys(df, "user", "item", n_factors=16, net_type=net_type, dynamic_neg_sampling=True,
                                rng="device", seed=3, **kw)
            w0 = model.net.user.weight.detach().clone()
            g = [torch.empty_like(w0) for _ in range(world)]
            dist.all_gather(g, w0)
            assert torch.equal(g[0], g[1]), "initial weights were not broadcast"
            opt = torch.optim.SGD(model.parameters(), lr=0.05)
            model.fit(opt, epochs=2, batch_size=256)
            model.evaluate(batch_size=256)
        # after the per-epoch table average the replicas hold identical tables; dense MLP parameters never diverged
        for name, p in model.net.named_parameters():
            g = [torch.empty_like(p.data) for _ in range(world)]
            dist.all_gather(g, p.data.contiguous())
            assert torch.allclose(g[0], g[1], rtol=0, atol=1e-6), name
        losses = [float(x.split(":")[-1]) for x in buf.getvalue().splitlines() if "Training Loss" in x]
        assert len(losses) == 2 and losses[1] < losses[0] + 1e-3
        n_train = n - int(np.ceil(0.2 * n))
        assert model.make_runner(opt, 256).n_train == n_train // world  # equally long shards: same step count on every rank
        ret[rank] = losses
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("net_type,n", [("fm", 6000), ("mlp", 6000), ("mlp", 5762)])
def test_two_rank_data_parallel_fit(net_type, n):
    """n = 5762: the training split (4608 + 1 rows) does not divide by the world size and sits right at a batch boundary —
    with unequal shards one rank would run a 10th batch and wait forever in the per-step gradient all-reduce."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), net_type, n, ret), nprocs=world, join=True)
    assert sorted(ret.keys()) == [0, 1]
    assert ret[0] == ret[1]  # the printed loss is the mean over ranks: identical on both
