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


def _worker(rank, world, port, net_type, n, ret, partition="user"):
    os.environ["TRS_FLAG_ONE_LAUNCH"] = "0"  # two processes on one GPU (see test_bench_two_ranks_rehearsal_on_one_gpu)
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
            model.dp_partition = partition
            w0 = model.net.user.weight.detach().clone()
            g = [torch.empty_like(w0) for _ in range(world)]
            dist.all_gather(g, w0)
            assert torch.equal(g[0], g[1]), "initial weights were not broadcast"
            opt = torch.optim.SGD(model.parameters(), lr=0.05)
            model.fit(opt, epochs=2, batch_size=256)
            model.evaluate(batch_size=256)
        # after the per-epoch average of the shared tables (and, by-user partition, the all-gather of the owners' user
        # rows at the end of fit) the replicas hold identical tables; dense MLP parameters never diverged
        for name, p in list(model.net.named_parameters()) + [(n_, b_) for n_, b_ in model.net.named_buffers()
                                                              if b_.is_floating_point()]:
            g = [torch.empty_like(p.data) for _ in range(world)]
            dist.all_gather(g, p.data.contiguous())
            assert torch.allclose(g[0], g[1], rtol=0, atol=1e-6), name
        losses = [float(x.split(":")[-1]) for x in buf.getvalue().splitlines() if "Training Loss" in x]
        assert len(losses) == 2 and losses[1] < losses[0] + 1e-3
        n_train = n - int(np.ceil(0.2 * n))
        r = model.make_runner(opt, 256)
        both = [torch.zeros(3, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(both, torch.tensor([r.n_train, r.num_batches, r.own_batches]))
        # no row is dropped by the cut itself (contiguous blocks: the n % world last rows); the MLP, whose steps hold a
        # collective, runs the common (minimum) number of steps per epoch, Linear / FM every rank its own
        assert int(both[0][0]) + int(both[1][0]) == (n_train if partition == "user" else 2 * (n_train // world))
        if net_type == "mlp":
            assert int(both[0][1]) == int(both[1][1]) == min(int(both[0][2]), int(both[1][2]))
        else:
            assert r.num_batches == r.own_batches
        if partition == "user":  # the rank trains only its own users
            users = model._rank_rows(model.data_processor.train_data)["user_id"]
            assert bool(((users % world) == rank).all())
        else:
            assert r.n_train == n_train // world
        ret[rank] = losses
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("net_type,n,partition", [("fm", 6000, "user"), ("mlp", 6000, "user"), ("mlp", 5762, "user"),
                                                  ("linear", 6000, "user"), ("fm", 6000, "contiguous"),
                                                  ("mlp", 5762, "contiguous")])
def test_two_rank_data_parallel_fit(net_type, n, partition):
    """n = 5762: the training split (4608 + 1 rows) does not divide by the world size and sits right at a batch boundary —
    with unequal shards one rank would run a 10th batch and wait forever in the per-step gradient all-reduce.
    partition 'user' (default): stream cut by user_id % world, user rows gathered from their owners after fit()."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), net_type, n, ret, partition), nprocs=world, join=True)
    assert sorted(ret.keys()) == [0, 1]
    assert ret[0] == ret[1]  # the printed loss is the mean over ranks: identical on both


def _sync_bn_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from torchrecsys_amd import ops
        from torchrecsys_amd.collaborative.mlp import MLP
        from torchrecsys_amd.dist import FlatGradBucket
        NU, NI, D, B = 300, 200, 32, 256  # B per rank; the single-process reference sees 2B
        torch.manual_seed(7)  # same weights on both ranks
        net = MLP(NU, NI, {"c": 11}, D, use_metadata=True, hidden_layers=[64, 32]).to("cuda:0")
        net.train()
        rs = np.random.RandomState(3)
        full = {"user": rs.randint(0, NU, 2 * B), "pos": rs.randint(0, NI, 2 * B), "neg": rs.randint(0, NI, 2 * B),
                "pos_meta": rs.randint(0, 11, (2 * B, 1)), "neg_meta": rs.randint(0, 11, (2 * B, 1))}
        dev = lambda d: {k: torch.from_numpy(v).to("cuda:0").contiguous() for k, v in d.items()}
        state0 = {k: v.clone() for k, v in net.state_dict().items()}
        # ---- one process, batch 2B, ordinary BatchNorm
        net.compute.sync_bn = False
        sc, ctx = net.compute.forward(dev(full), 2, True)
        g = torch.cat([torch.full((2 * B,), -1.0 / (2 * B)), torch.full((2 * B,), 1.0 / (2 * B))]).to("cuda:0")
        g = g * torch.linspace(0.5, 1.5, 4 * B, device="cuda:0")  # not a constant: every BN sum matters
        ref_grads, ref_dx0 = net.compute.backward(ctx, g)
        ref_state = {k: v.clone() for k, v in net.state_dict().items()}
        ref_scores = sc.clone()
        # ---- two ranks, batch B each, synchronised statistics
        net.load_state_dict(state0)
        net.compute.sync_bn = True
        half = {k: v[rank * B:(rank + 1) * B] for k, v in full.items()}
        sc, ctx = net.compute.forward(dev(half), 2, True)
        gh = torch.cat([g[:2 * B][rank * B:(rank + 1) * B], g[2 * B:][rank * B:(rank + 1) * B]]) * world  # 1/B_local scaling
        bucket = FlatGradBucket(net.dense_params())
        grads, dx0 = net.compute.backward(ctx, gh, grad_of=bucket.grad_of)
        bucket.allreduce_mean_()
        want = torch.cat([ref_scores[:2 * B][rank * B:(rank + 1) * B], ref_scores[2 * B:][rank * B:(rank + 1) * B]])
        err = float((sc - want).abs().max() / want.abs().max())
        assert err < 1e-5, ("scores", err)
        for p in net.dense_params():
            a, b = bucket.grad_of(p), ref_grads[p]
            e = float((a - b).abs().max() / max(float(b.abs().max()), 1e-12))
            assert e < 2e-5 or float(b.abs().max()) < 1e-7, (e, tuple(p.shape))
        # this rank's rows of the embedding gradient: dx0 is scaled by world relative to the global-mean loss
        ref_rows = torch.cat([ref_dx0[:2 * B][rank * B:(rank + 1) * B], ref_dx0[2 * B:][rank * B:(rank + 1) * B]])
        e = float((dx0 / world - ref_rows).abs().max() / ref_rows.abs().max())
        assert e < 2e-5, ("dx0", e)
        for k, v in net.state_dict().items():
            if "running" in k:
                e = float((v - ref_state[k]).abs().max() / ref_state[k].abs().max())
                assert e < 1e-5, (k, e)
            if "num_batches_tracked" in k:
                assert int(v) == int(ref_state[k]) == 2
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


def test_two_rank_sync_batchnorm_equals_one_process():
    """fit(sync_bn=True): two ranks with batch B each compute, on the dense path, what one process computes with batch
    2B — scores at 1e-5, the all-reduced dense gradients, the embedding-gradient rows and the BatchNorm running
    statistics (forward statistics and the two backward sums are all-reduced per layer and pass)."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_sync_bn_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}


def test_bench_two_ranks_rehearsal_on_one_gpu():
    """The driver's multi-GPU command line (torch.distributed.run, one rank per GPU, bench.py --gpus N) rehearsed with two
    ranks sharing this box's GPU and gloo carrying the collectives: rank 0 prints ONE JSON line for the default (c4)
    workload with n_gpus = 2, the whole-job value, weak scaling, and the replica-synchronisation costs fit() adds."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # (two processes computing on ONE GPU: the one-launch flag-mode step needs its whole grid resident at once, which two
    # such grids plus their presort kernels are not — the documented setting for shared devices)
    env = dict(os.environ, TRS_BENCH_SHARE_DEVICE="1", TRS_DIST_BACKEND="gloo", TRS_FLAG_ONE_LAUNCH="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "24",
           "--warmup", "8"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 24 and d["scaling"] == "weak" and "c4" in d["config"]["workload"]
    assert d["config"]["global_batch"] == 2 * d["config"]["per_gpu_batch"] == 65_536
    assert d["value"] == pytest.approx(2 * 65_536 * 24 / (d["ms_per_step"] * 24 * 1e-3), rel=1e-6)
    rs = d["replica_sync"]
    assert rs["averaged_bytes"] == 4 * (1_000_000 * 128 + 1_000_000) and rs["per_epoch_average_ms"] > 0
    assert "cpu_baseline" not in d  # rank 0 at N = 1 only
