# -*- coding: utf-8 -*-
"""The N>1 path on CPU: world_size-2 `gloo` processes exercising stream sharding, the flat dense-gradient all-reduce,
table averaging and the initial broadcast (torchrecsys_amd/dist.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from torchrecsys_amd import dist as tdist
        assert tdist.world_info() == (rank, world)
        # --- stream sharding: contiguous and by-user partitions are disjoint and cover the stream
        users = torch.arange(1001) % 37
        items = torch.arange(1001) % 11
        u, i = tdist.shard_stream(users, items)
        s, e = tdist.shard_bounds(1001, rank, world)
        assert torch.equal(u, users[s:e]) and torch.equal(i, items[s:e])
        cnt = torch.tensor([u.numel()])
        dist.all_reduce(cnt)
        assert cnt.item() == 1001
        ub, ib = tdist.shard_stream(users, items, by_user=True)
        assert ((ub % world) == rank).all()
        cnt = torch.tensor([ub.numel()])
        dist.all_reduce(cnt)
        assert cnt.item() == 1001
        # --- flat dense-gradient bucket: ONE all-reduce gives every rank the mean gradient
        torch.manual_seed(0)
        params = [torch.nn.Parameter(torch.zeros(8, 4)), torch.nn.Parameter(torch.zeros(5))]
        bucket = tdist.FlatGradBucket(params)
        bucket.grad_of(params[0]).copy_(torch.full((8, 4), float(rank + 1)))
        bucket.grad_of(params[1]).copy_(torch.arange(5.0) * (rank + 1))
        bucket.allreduce_mean_()
        assert torch.allclose(bucket.grad_of(params[0]), torch.full((8, 4), 1.5))
        assert torch.allclose(bucket.grad_of(params[1]), torch.arange(5.0) * 1.5)
        # async form (overlap with backward)
        bucket.flat.fill_(float(rank))
        work = bucket.allreduce_mean_(async_op=True)
        bucket.finish_(work)
        assert torch.allclose(bucket.flat, torch.full_like(bucket.flat, 0.5))
        # --- replicas: broadcast of initial weights, periodic averaging, scalar reduction
        t = torch.full((6, 3), float(rank + 10))
        tdist.broadcast_([t])
        assert (t == 10).all()
        t += rank * 2
        tdist.average_tables_([t])
        assert torch.allclose(t, torch.full((6, 3), 11.0))
        tot = tdist.allreduce_scalar_sum([1.0, float(rank)], torch.device("cpu"))
        assert tot == [2.0, 1.0]
        # --- per-layer segments of the bucket: reduced one by one (as the backward produces them), scaled once
        segs = bucket.segments([[params[0]], [params[1]]])
        assert segs == [(0, 32), (32, 37)]
        bucket.flat.fill_(float(rank + 1))
        works = [bucket.allreduce_segment_async(sg) for sg in reversed(segs)]
        bucket.finish_segments(works)
        assert torch.allclose(bucket.flat, torch.full_like(bucket.flat, 1.5))
        # --- user-partitioned replicas: every rank owns the rows r, r+world, ...; all-gathers make the table whole, in
        # place and in bounded pieces (chunk_bytes = 24 -> 2 rows of 3 floats per rank and block: 7 rows = 2 blocks, the
        # second one ragged; 12 -> 1 row per block; default: one block)
        for nrows, chunk in ((7, 24), (7, 12), (7, 64 << 20), (8, 24), (1, 24), (0, 24)):
            tab = torch.full((nrows, 3), -1.0)
            tab[rank::world] = torch.arange(float(nrows))[rank::world, None] * 10 + rank
            tdist.gather_owned_rows_(tab, chunk_bytes=chunk)
            want = torch.arange(float(nrows))[:, None] * 10 + (torch.arange(nrows) % world)[:, None].float()
            assert torch.equal(tab, want.expand(nrows, 3)), (nrows, chunk)
        vec = torch.full((9,), -1.0)  # 1-wide tables stored as vectors work too
        vec[rank::world] = float(rank)
        tdist.gather_owned_rows_(vec, chunk_bytes=8)
        assert torch.equal(vec, (torch.arange(9) % world).float())
        assert tdist.allreduce_min_int(10 + rank, torch.device("cpu")) == 10
        # --- the model's per-rank view of a split: cut by user (default) or in contiguous blocks; NO row is dropped
        # (Linear / FM steps hold no collective; the MLP's lock-step is FitRunner's MIN over the ranks' step counts)
        from torchrecsys_amd.model import TorchRecSys
        data = {"user_id": torch.tensor([0, 1, 2, 3, 4, 5, 6, 8, 10, 12]), "pos_item_id": torch.arange(10)}
        m = TorchRecSys.__new__(TorchRecSys)
        m._dev_cache = {}
        m.dp_partition = "user"
        mine = m._rank_rows(data)  # even users: 0 2 4 6 8 10 12 (7 rows), odd users: 1 3 5 (3 rows)
        assert ((mine["user_id"] % world) == rank).all() and mine["user_id"].numel() == (7, 3)[rank]
        assert torch.equal(mine["pos_item_id"], torch.arange(10)[(data["user_id"] % world) == rank])
        assert m._rank_rows(data) is mine  # cached per split
        m2 = TorchRecSys.__new__(TorchRecSys)
        m2._dev_cache = {}
        m2.dp_partition = "contiguous"
        data11 = {"user_id": torch.arange(11), "pos_item_id": torch.arange(11)}
        assert torch.equal(m2._rank_rows(data11)["user_id"], torch.arange(11)[rank * 5:rank * 5 + 5])
        assert tdist.equal_shard_bounds(11, rank, world) == (rank * 5, rank * 5 + 5)  # same length on every rank
        # a caller's own shards (pre_sharded): kept whole, of any length — but under dp_partition 'user' they must BE
        # the cut by user_id % world (fit() never averages user tables then and gathers rows r::world from rank r);
        # any other cut raises on EVERY rank (the verdict is all-reduced), 'contiguous' accepts it
        m3 = TorchRecSys.__new__(TorchRecSys)
        m3._dev_cache = {}
        m3.dp_partition = "user"
        m3.pre_sharded = True
        own = {"user_id": torch.arange(10 - 3 * rank) * world + rank, "pos_item_id": torch.arange(10 - 3 * rank)}
        assert m3._rank_rows(own)["user_id"].numel() == 10 - 3 * rank
        bad = {"user_id": torch.arange(6) if rank == 1 else torch.arange(6) * world, "pos_item_id": torch.arange(6)}
        with pytest.raises(ValueError, match="user_id % world == rank"):  # rank 0's shard is fine: it raises too
            m3._rank_rows(bad)
        m4 = TorchRecSys.__new__(TorchRecSys)
        m4._dev_cache = {}
        m4.dp_partition = "contiguous"
        m4.pre_sharded = True
        assert m4._rank_rows(bad)["user_id"].numel() == 6
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_data_parallel_path():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}


def test_single_process_helpers_are_no_ops():
    from torchrecsys_amd import dist as tdist
    assert tdist.world_info() == (0, 1)
    u, i = tdist.shard_stream(torch.arange(5), torch.arange(5))
    assert u.numel() == 5
    b = tdist.FlatGradBucket([torch.nn.Parameter(torch.ones(3))])
    assert b.allreduce_mean_() is None
    assert tdist.shard_bounds(10, 0, 3) == (0, 4) and tdist.shard_bounds(10, 2, 3) == (7, 10)
