# -*- coding: utf-8 -*-
"""CPU baseline of the training loop: a port of the reference's op SEQUENCE onto torch CPU kernels, timed by bench.py's
`cpu_baseline` leg (kind "port") on the GPU box's host cores.  TEST INFRASTRUCTURE — never imported by torchrecsys_amd.

It runs what the reference runs per step: per-row Python sampler loop (dataset/dataset.py:435-447),
tensor[perm[i:i+B]] slicing (:420-422), two scoring passes over nn.Embedding(sparse=True) tables with the reference's
elementwise op chain (collaborative/fm.py:60-101, linear.py:54-80), hinge (helper/loss.py:5-9), autograd backward and
torch.optim.SGD on sparse gradients (model.py:188-200).  tests/test_oracle_golden.py pins its numbers to the golden
vectors of the real reference through oracle/nets.py (same math); this file exists to be TIMED.
"""
import time

import numpy as np
import torch


class _Tables(torch.nn.Module):
    def __init__(self, net_type, n_users, n_items, D):
        super().__init__()
        E = torch.nn.Embedding
        self.net_type = net_type
        self.user = E(n_users, D, sparse=True)
        self.item = E(n_items, D, sparse=True)
        self.user1 = E(n_users, 1, sparse=True)   # FM linear_user / Linear user_bias
        self.item1 = E(n_items, 1, sparse=True)
        with torch.no_grad():
            self.user.weight.normal_(0, 1.0 / D)
            self.item.weight.normal_(0, 1.0 / D)
            if net_type == "fm":
                self.user1.weight.normal_(0, 1.0)
                self.item1.weight.normal_(0, 1.0)
            else:
                self.user1.weight.zero_()
                self.item1.weight.zero_()

    def score(self, u, i):
        B = u.shape[0]
        D = self.user.embedding_dim
        if self.net_type == "fm":
            ue = self.user(u).reshape(B, 1, D)
            ie = self.item(i).reshape(B, 1, D)
            emb = torch.cat([ue, ie], dim=1)
            power_of_sum = emb.sum(dim=1).pow(2)
            sum_of_power = emb.pow(2).sum(dim=1)
            pairwise = (power_of_sum - sum_of_power).sum(1) * 0.5
            lin = torch.cat([self.user1(u).reshape(B, 1, 1), self.item1(i).reshape(B, 1, 1)], dim=1).sum(1).reshape(B)
            return torch.sigmoid(lin + pairwise)
        return (self.user(u) * self.item(i)).sum(1).view(-1, 1) + self.user1(u) + self.item1(i)


class _MLPTables(torch.nn.Module):
    """Op sequence of collaborative/mlp.py:66-115: sparse embeddings, concat, (Linear, BatchNorm1d, relu)*, Linear."""

    def __init__(self, n_users, n_items, D, meta_cats, hidden):
        super().__init__()
        E = torch.nn.Embedding
        self.user = E(n_users, D, sparse=True)
        self.item = E(n_items, D, sparse=True)
        self.meta = torch.nn.ModuleList([E(c, D, sparse=True) for c in meta_cats])
        with torch.no_grad():
            for t in [self.user, self.item, *self.meta]:
                t.weight.normal_(0, 1.0 / D)
        dims = [(2 + len(meta_cats)) * D] + list(hidden)
        self.fcs = torch.nn.ModuleList([torch.nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:])])
        self.bns = torch.nn.ModuleList([torch.nn.BatchNorm1d(b) for b in dims[1:]])
        self.out = torch.nn.Linear(dims[-1], 1)

    def score(self, u, i, m):
        x = torch.cat([self.user(u), self.item(i)] + [t(m[:, k]) for k, t in enumerate(self.meta)], dim=1)
        for fc, bn in zip(self.fcs, self.bns):
            x = torch.relu(bn(fc(x)))
        return self.out(x)


def _sample_loop(pos_list, n_items):
    out = []
    for p in pos_list:
        neg = np.random.randint(0, n_items)
        while neg == p:
            neg = np.random.randint(0, n_items)
        out.append(neg)
    return out


def time_steps(net_type, n_users, n_items, D, batch_size, n_rows, steps, warmup=1, dynamic=True, lr=1e-2, threads=None,
               seed=7, max_seconds=30.0, meta_cats=(), hidden=None, optimizer="sgd"):
    """Run `warmup` + up to `steps` training steps on a synthetic stream of `n_rows` interactions and return
    {"interactions_per_s", "steps", "seconds", "threads"}.  Stops early once `max_seconds` of timed work is reached."""
    if threads:
        torch.set_num_threads(threads)
    np.random.seed(seed)
    torch.manual_seed(seed)
    users = torch.randint(0, n_users, (n_rows,))
    items = torch.randint(0, n_items, (n_rows,))
    static_neg = torch.randint(0, n_items, (n_rows,))
    is_mlp = net_type == "mlp"
    if is_mlp:
        net = _MLPTables(n_users, n_items, D, list(meta_cats), list(hidden))
        item_meta = torch.stack([torch.randint(0, c, (n_items,)) for c in meta_cats], dim=1) if meta_cats else \
            torch.zeros((n_items, 0), dtype=torch.long)
    else:
        net = _Tables(net_type, n_users, n_items, D)
    opt = {"sgd": lambda ps: torch.optim.SGD(ps, lr=lr), "sparse_adam": lambda ps: torch.optim.SparseAdam(list(ps), lr=lr),
           "adagrad": lambda ps: torch.optim.Adagrad(ps, lr=lr)}[optimizer](net.parameters())
    perm = torch.randperm(n_rows)
    done, t_total, i = 0, 0.0, 0
    for s in range(warmup + steps):
        if i + batch_size > n_rows:
            i = 0
        t0 = time.perf_counter()
        idx = perm[i:i + batch_size]
        u, p = users[idx], items[idx]
        if dynamic:
            n = torch.tensor(_sample_loop(p.tolist(), n_items), dtype=torch.long)
        else:
            n = static_neg[idx]
        if is_mlp:  # metadata of an item by table lookup (the reference walks a dict per row, dataset.py:391-396)
            pos, neg = net.score(u, p, item_meta[p]), net.score(u, n, item_meta[n])
        else:
            pos, neg = net.score(u, p), net.score(u, n)
        loss = torch.clamp(neg - pos + 1.0, 0.0).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        loss.item()
        dt = time.perf_counter() - t0
        i += batch_size
        if s >= warmup:
            t_total += dt
            done += 1
            if t_total >= max_seconds:
                break
    return {"interactions_per_s": 2.0 * batch_size * done / t_total, "steps": done, "seconds": t_total,
            "threads": torch.get_num_threads()}
