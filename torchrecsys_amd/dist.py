# -*- coding: utf-8 -*-
"""Multi-GPU plumbing (one process per GPU, torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).

The reference is single-process (no collective anywhere).  The data-parallel design of this framework
(BASELINE.json north_star):
  * the interaction stream is sharded across ranks — each rank trains on its own rows, no per-step exchange of ids;
  * embedding tables are replicated and updated locally (no all-to-all, no row exchange: xGMI's per-link bandwidth is
    ~1/50 of HBM's, so exchanging rows every step would cap 8 GPUs below one GPU's own roofline);
  * only the dense MLP parameters are kept identical: their gradients live in ONE flat buffer that is all-reduced
    (SUM, then scaled by 1/world) once per step — 1.45 MB at config c3, 7.9 MB at c5, a single ring collective;
  * optionally the replicated tables are re-averaged once per epoch (average_tables_).
Everything here is device-agnostic so the N>1 path is exercised by world_size-2 `gloo` tests on CPU.
"""
import torch
import torch.distributed as dist


def world_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_bounds(n, rank, world):
    """Contiguous [start, end) of `n` rows owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def equal_shard_bounds(n, rank, world):
    """[start, end) of equally long shards (n // world rows each; the n % world last rows are left out): every rank
    then runs the same number of steps, which the per-step collectives of the MLP path require."""
    per = n // world
    return rank * per, rank * per + per


def shard_stream(user_ids, item_ids, rank=None, world=None, by_user=False):
    """This rank's rows of the interaction stream.  by_user=True partitions by user_id % world (every user row is then
    owned by exactly one rank — the largest table never drifts between replicas)."""
    if rank is None:
        rank, world = world_info()
    if world == 1:
        return user_ids, item_ids
    if by_user:
        keep = (user_ids % world) == rank
        return user_ids[keep], item_ids[keep]
    s, e = shard_bounds(user_ids.shape[0], rank, world)
    return user_ids[s:e], item_ids[s:e]


class FlatGradBucket:
    """Dense parameters' gradients as views of one contiguous buffer, so a step needs ONE all-reduce."""

    def __init__(self, params):
        self.params = [p for p in params]
        n = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(n, dtype=ref.dtype, device=ref.device)
        self.views = []
        o = 0
        for p in self.params:
            v = self.flat[o:o + p.numel()].view_as(p)
            self.views.append(v)
            o += p.numel()

    def segments(self, groups):
        """[(start, end)] of the flat buffer for consecutive groups of parameters (`groups`: lists of parameters in the
        bucket's own order, e.g. one list per layer): lets a layer's gradients be reduced as soon as its backward has
        written them, while the layers below are still being differentiated."""
        out, o, i = [], 0, 0
        for g in groups:
            n = 0
            for p in g:
                assert self.params[i] is p, "segments() takes the bucket's parameters in order"
                n += p.numel()
                i += 1
            out.append((o, o + n))
            o += n
        assert i == len(self.params)
        return out

    def allreduce_segment_async(self, seg, group=None):
        """Start SUM over ranks of flat[seg[0]:seg[1]] on the collective stream; finish_segments() scales by 1/world."""
        rank, world = world_info()
        if world == 1:
            return None
        return dist.all_reduce(self.flat[seg[0]:seg[1]], op=dist.ReduceOp.SUM, group=group, async_op=True)

    def finish_segments(self, works):
        works = [w for w in works if w is not None]
        for w in works:
            w.wait()
        if works:
            self.flat.mul_(1.0 / world_info()[1])

    def grad_of(self, p):
        for q, v in zip(self.params, self.views):
            if q is p:
                return v
        raise KeyError("parameter not in bucket")

    def allreduce_mean_(self, group=None, async_op=False):
        """SUM over ranks, then 1/world: every replica applies the gradient of the global batch mean."""
        rank, world = world_info()
        if world == 1:
            return None
        work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        if async_op:
            return work
        self.flat.mul_(1.0 / world)
        return None

    def finish_(self, work):
        if work is not None:
            work.wait()
            self.flat.mul_(1.0 / world_info()[1])


def allreduce_min_int(value, device):
    """MIN of one host integer over ranks (the common number of rows / steps every rank can run)."""
    rank, world = world_info()
    if world == 1:
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item())


def gather_owned_rows_(table, group=None, chunk_bytes=64 << 20):
    """User-partitioned data parallelism (stream sharded by user_id % world): rank r is the only writer of the rows
    r, r + world, r + 2*world, ... of a user table; every other row of its replica is stale.  All-gathers of the
    owned rows (1/world of the table sent per rank — half the bytes per link of an all-reduce of masked tables) make
    every replica whole again.  Called once at the end of fit(), not per epoch: nobody reads a foreign user row in
    between.

    In place, in bounded pieces: the table is walked in blocks of C*world consecutive rows; of each block every rank
    sends its C owned rows (a contiguous staging copy of at most `chunk_bytes`) into ONE all_gather_into_tensor whose
    (world, C, ...) result is written back through the block's (C, world, ...) view with a single strided copy.  Peak
    extra memory = (world + 1) * chunk_bytes whatever the table size (c5's 10.2 GB user table: 0.6 GB at 8 ranks, where a
    list of per-rank receive buffers was a second copy of the table)."""
    rank, world = world_info()
    n = table.shape[0]
    if world == 1 or n == 0:
        return
    tail = tuple(table.shape[1:])
    row_bytes = max(table[0].numel() * table.element_size(), 1) if n else 1
    C = max(1, int(chunk_bytes) // row_bytes)
    send = torch.empty((min(C, (n + world - 1) // world),) + tail, dtype=table.dtype, device=table.device)
    C = send.shape[0]
    recv_flat = torch.empty((world * C,) + tail, dtype=table.dtype, device=table.device)  # rank-major (gloo wants 2-D+)
    recv = recv_flat.view((world, C) + tail)
    for r0 in range(0, n, C * world):  # block = rows [r0, r0 + C*world); its row r0 + c*world + k belongs to rank k
        rows = min(C * world, n - r0)
        full = rows // world          # whole (c, all ranks) groups of the block
        c_mine = (rows - rank + world - 1) // world if rows > rank else 0
        if c_mine:
            send[:c_mine].copy_(table[r0 + rank:r0 + rows:world])
        dist.all_gather_into_tensor(recv_flat, send, group=group)
        if full:
            table[r0:r0 + full * world].view((full, world) + tail).copy_(recv[:, :full].transpose(0, 1))
        for k in range(rows - full * world):  # the ragged last group of the table: ranks 0 .. rem-1 own one row more
            table[r0 + full * world + k].copy_(recv[k, full])


def average_tables_(tables, group=None):
    """Replace every replicated table by its mean over ranks (periodic re-synchronisation of embedding replicas)."""
    rank, world = world_info()
    if world == 1:
        return
    for t in tables:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        t.mul_(1.0 / world)


def broadcast_(tensors, src=0, group=None):
    """Initial weights from rank 0 so that every replica starts identical."""
    rank, world = world_info()
    if world == 1:
        return
    for t in tensors:
        dist.broadcast(t, src=src, group=group)


def allreduce_scalar_sum(values, device):
    """Sum of a few host scalars over ranks (epoch loss / AUC counters)."""
    rank, world = world_info()
    t = torch.tensor(values, dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.tolist()
