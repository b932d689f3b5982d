# -*- coding: utf-8 -*-
"""TorchRecSys — drop-in for the reference's torchrecsys.model.TorchRecSys (reference model.py:18-452) whose
fit() / evaluate() / predict() run on hand-written HIP kernels on the MI355X.

Same constructor keywords, same methods, same printed strings, same state_dict keys.  Extra keywords (superset API):
  hidden_layers, use_batch_norm : reach the MLP (the reference advertises them but cannot pass them)
  rng : 'reference' (default) replays the reference's host RNG streams (torch.randperm shuffle, numpy legacy sampler) so
        a seeded run sees bit-identical batches;  'device' keeps the whole interaction stream in HBM and shuffles /
        samples on the GPU with counter-based generators (same distributions, different streams) — the mode the
        benchmark runs in
  seed : seed of the device-side generators (rng='device')
"""
import functools
import math
import os
from typing import List

import numpy as np
import pandas as pd
import torch
import torch.profiler

from . import dist as tdist
from . import ops
from .collaborative._scorer import check_err_flag
from .collaborative.fm import FM
from .collaborative.linear import Linear
from .dataset.dataset import FastDataLoader, ProcessData, sample_negatives_reference_stream
from .engine import SparseScorerTrainer
from .evaluate.metrics import Metrics
from .helper.cuda import gpu, host_threads
from .helper.loss import hinge_loss  # noqa: F401  (part of the reference's module surface)


def _host_side(fn):
    """Run a public entry point under helper.cuda.host_threads() (torch's CPU thread pool capped at the CPU budget)."""
    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        with host_threads():
            return fn(*args, **kwargs)
    return wrapped

_NO_GPU_MSG = ("torchrecsys_amd needs an AMD Instinct MI355X (gfx950) visible to PyTorch-ROCm; "
               "it has no CPU fallback (use_cuda=False only means: hand results back as CPU tensors)")


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError(_NO_GPU_MSG)
    return torch.device("cuda", torch.cuda.current_device())


def _mix64(a, b):
    """SplitMix64-style hash of two integers -> 64-bit key (device shuffle / sampler keys per epoch)."""
    x = (a * 0x9E3779B97F4A7C15 + b + 0x632BE59BD9B4E019) & 0xFFFFFFFFFFFFFFFF
    x ^= x >> 30
    x = (x * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    x ^= x >> 27
    x = (x * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    x ^= x >> 31
    return x or 1


class TorchRecSys(torch.nn.Module):

    @_host_side
    def __init__(self,
                 dataset: pd.DataFrame,
                 user_id_col: str,
                 item_id_col: str,
                 n_factors: int = 80,
                 net_type: str = 'linear',
                 metadata_id_col: List[str] = None,
                 split_ratio: float = 0.8,
                 dynamic_neg_sampling: bool = False,
                 use_amp: bool = False,
                 use_cuda: bool = False,
                 debug: bool = False,
                 path: str = './',
                 hidden_layers: List[int] = None,
                 use_batch_norm: bool = True,
                 rng: str = 'reference',
                 seed: int = 0,
                 neg_sampling: dict = None):
        super().__init__()
        self.neg_sampling = neg_sampling
        data_processor = ProcessData(dataset=dataset, user_id_col=user_id_col, item_id_col=item_id_col,
                                     metadata_id_col=metadata_id_col, split_ratio=split_ratio,
                                     dynamic_neg_sampling=dynamic_neg_sampling)
        self._setup(data_processor, metadata_id_col, n_factors, net_type, dynamic_neg_sampling, use_amp, use_cuda,
                    debug, path, hidden_layers, use_batch_norm, rng, seed)

    @classmethod
    def from_tensors(cls, user_ids, item_ids, n_users=None, n_items=None, item_metadata=None, metadata_names=None,
                     n_factors=80, net_type='linear', split_ratio=0.8, dynamic_neg_sampling=False, use_amp=False,
                     use_cuda=False, debug=False, path='./', hidden_layers=None, use_batch_norm=True, rng=None,
                     seed=0, pre_sharded=False, dp_partition='user', remap_ids=False, split=None, neg_sampling=None):
        """Tensor-native ingest (no DataFrame): id tensors on the CPU or already in HBM.  GPU tensors default to
        rng='device' (stream resident in HBM, on-device shuffle and sampler).  pre_sharded=True: under data parallelism
        the given interactions already ARE this rank's shard (each rank ingested its own part), so they are not cut
        again by rank; shards may differ in length (every rank then trains on the common number of rows, see
        _rank_rows).  dp_partition: how the stream is (pre_sharded: was) cut — 'user' = by user_id % world (each user
        row has ONE writer, see fit()), 'contiguous' = equal contiguous blocks.  remap_ids / split: see
        dataset.TensorProcessData (dense re-mapping of arbitrary ids; 'reference' = the reference's RandomState(42)
        split also for GPU tensors, 'device' = a seeded permutation drawn on the GPU)."""
        from .dataset.dataset import TensorProcessData
        self = cls.__new__(cls)
        torch.nn.Module.__init__(self)
        self.neg_sampling = neg_sampling
        dp = TensorProcessData(user_ids, item_ids, n_users, n_items, item_metadata, metadata_names, split_ratio,
                               dynamic_neg_sampling, remap_ids=remap_ids, split=split)
        if rng is None:
            rng = 'device' if user_ids.is_cuda else 'reference'
        if user_ids.is_cuda and rng != 'device':
            raise ValueError("GPU-resident id tensors require rng='device'")
        self._setup(dp, dp.metadata_id_col, n_factors, net_type, dynamic_neg_sampling, use_amp, use_cuda, debug, path,
                    hidden_layers, use_batch_norm, rng, seed)
        self.pre_sharded = bool(pre_sharded)
        assert dp_partition in ('user', 'contiguous')
        self.dp_partition = dp_partition
        return self

    def _setup(self, data_processor, metadata_id_col, n_factors, net_type, dynamic_neg_sampling, use_amp, use_cuda,
               debug, path, hidden_layers, use_batch_norm, rng, seed):
        assert rng in ('reference', 'device'), 'rng must be "reference" or "device"'
        ns = getattr(self, "neg_sampling", None)
        if ns:
            unknown = set(ns) - {"reject_seen", "popularity", "k", "max_tries"}
            if unknown or rng != 'device' or not dynamic_neg_sampling:
                raise ValueError("neg_sampling takes reject_seen / popularity / k / max_tries and needs rng='device' with "
                                 "dynamic_neg_sampling=True (the reference-RNG mode replays the reference's sampler)")
        self.path = path
        self.dynamic_neg_sampling = dynamic_neg_sampling
        self.use_amp = use_amp
        self.use_cuda = use_cuda
        self.rng = rng
        self.seed = seed
        self.grad_scaler = None  # bf16 GEMM inputs with fp32 accumulation need no loss scaling
        self.data_processor = data_processor
        self.data_processor.prepare_data()
        self.config = self.data_processor.config
        self.n_users = self.config.get('num_users')
        self.n_items = self.config.get('num_items')
        self.metadata_size = self.config.get('num_metadata')
        self.metadata_name = metadata_id_col if getattr(self.data_processor, 'metadata_id_col', None) else None
        self.n_factors = n_factors
        self.net_type = net_type
        self.use_metadata = True if self.metadata_name else False
        self.debug = debug
        self.hidden_layers = hidden_layers
        self.use_batch_norm = use_batch_norm
        self._fit_epochs_done = 0
        self._dev_cache = {}
        self.dp_partition = 'user'  # data-parallel cut of the interaction stream (fit() docstring)
        self._init_net(net_type=net_type)

    # ------------------------------------------------------------------------------------------------ net
    def _init_net(self, net_type='linear'):
        assert net_type in ('linear', 'mlp', 'neucf', 'fm', 'lstm'), \
            'Net type must be one of "linear", "mlp", "neu", "ease" or "lstm"'
        kw = dict(n_users=self.n_users, n_items=self.n_items, n_metadata=self.metadata_size,
                  n_factors=self.n_factors, use_metadata=self.use_metadata, use_cuda=self.use_cuda)
        if net_type == 'linear':
            print('Linear Collaborative Filtering')
            self.net = Linear(**kw)
        elif net_type == 'mlp':
            print('Multi Layer Perceptron')
            from .collaborative.mlp import MLP
            self.net = MLP(use_batch_norm=self.use_batch_norm, hidden_layers=self.hidden_layers,
                           use_bf16=self.use_amp, **kw)
        elif net_type == 'fm':
            print('Factorization Machine')
            self.net = FM(**kw)
        else:  # the reference silently leaves self.net undefined here (model.py:162-166)
            raise NotImplementedError(f'{net_type} is not implemented (nor is it in the reference)')
        # parameters are created on the host from torch's CPU generator (bit-identical init), then live in HBM
        if torch.cuda.is_available():
            self.net = self.net.to(_device())
            if tdist.world_info()[1] > 1:  # data parallel: every replica starts from rank 0's weights
                tdist.broadcast_([p.data for p in self.net.parameters()] + [b for b in self.net.buffers()])

    # ------------------------------------------------------------------------------------------------ forward
    def forward(self, net, batch):
        """Positive and negative scores of one batch (reference model.py:171-185), one fused kernel."""
        if hasattr(net, 'forward_pair'):
            return net.forward_pair(batch)
        positive = net.forward(batch, user_key='user_id', item_key='pos_item_id', metadata_key='pos_metadata_id')
        negative = net.forward(batch, user_key='user_id', item_key='neg_item_id', metadata_key='neg_metadata_id')
        return positive, negative

    def backward(self, loss_value, optimizer):
        """Generic autograd step (reference model.py:188-200) for callers that drive forward()/hinge_loss() themselves;
        fit() uses the fused engine instead."""
        optimizer.zero_grad()
        loss_value.backward()
        optimizer.step()
        return loss_value.item()

    # ------------------------------------------------------------------------------------------------ data staging
    def _id_dtype(self):
        big = max(self.n_users, self.n_items, *(list(self.metadata_size.values()) or [0]))
        return torch.int32 if big < 2 ** 31 else torch.int64

    def _host_epoch(self, data, loader):
        """All batches of one epoch in visiting order, consuming the reference's RNG streams exactly as its
        FastDataLoader would batch by batch (shuffle in __iter__, then the sampler walk in row order)."""
        order = loader.epoch_order()
        take = (lambda t: t[order]) if loader.shuffle else (lambda t: t)
        ep = {'user': take(data['user_id']), 'pos': take(data['pos_item_id'])}
        has_meta = 'pos_metadata_id' in data
        if has_meta:
            ep['pos_meta'] = take(data['pos_metadata_id'])
        if not self.dynamic_neg_sampling:
            ep['neg'] = take(data['neg_item_id'])
            if has_meta:
                ep['neg_meta'] = take(data['neg_metadata_id'])
        else:
            pos = ep['pos'].numpy()
            B = loader.batch_size
            neg = np.concatenate([sample_negatives_reference_stream(pos[i:i + B], self.n_items)
                                  for i in range(0, len(pos), B)]) if len(pos) else np.zeros(0, np.int64)
            ep['neg'] = torch.from_numpy(neg)
            if has_meta:
                ep['neg_meta'] = torch.from_numpy(self.data_processor.item_meta_table[neg])
        dt, dev = self._id_dtype(), _device()
        return {k: v.to(dt).contiguous().to(dev, non_blocking=True) for k, v in ep.items()}

    def _rank_rows(self, data):
        """This rank's rows of a split under data parallelism (the whole split in a single process).
        dp_partition 'user' (default): the rows whose user_id % world == rank — every user row then has exactly ONE
        writer, so the largest table never drifts between replicas and only item / metadata rows need the periodic
        average.  'contiguous': equal contiguous blocks.  pre_sharded: the caller's own cut — under
        dp_partition 'user' it is CHECKED to be the cut by user_id % world (one reduction per split, agreed over ranks so
        that every rank raises together): fit() never averages the user tables under that partition and its final
        gather_owned_rows_ takes rows r::world from rank r, so any other cut would silently replace trained user rows
        by another replica's stale ones.
        Rows are never dropped here: shards may differ in length.  Linear / FM steps contain no collective, so every
        rank simply runs its own number of steps; the MLP's lock-step (one gradient all-reduce per step) is kept by
        FitRunner, which runs the MINIMUM number of steps over ranks per epoch and leaves the shard whole — the rows
        beyond are a different set every epoch (the epoch shuffle), not a fixed tail.  evaluate() covers every row."""
        rank, world = tdist.world_info()
        if world == 1:
            return data
        key = id(data)
        cache = self._dev_cache.setdefault('shards', {})
        if key not in cache:
            if getattr(self, "pre_sharded", False):
                shard = data
                if getattr(self, "dp_partition", "user") == "user":
                    u = data['user_id']
                    ok = bool(((u % world) == rank).all()) if u.numel() else True
                    import torch.distributed as _d
                    dev = _device() if _d.get_backend() == 'nccl' else torch.device('cpu')
                    if tdist.allreduce_min_int(int(ok), dev) == 0:
                        raise ValueError(
                            "pre_sharded=True with dp_partition='user': every rank's interactions must satisfy "
                            f"user_id % world == rank (rank {rank}: {'ok' if ok else 'violated'}); pass "
                            "dp_partition='contiguous' for shards cut any other way (every table is then averaged)")
            elif getattr(self, "dp_partition", "user") == "user":
                keep = (data['user_id'] % world) == rank
                shard = {k: v[keep] for k, v in data.items()}
            else:
                s, e = tdist.equal_shard_bounds(data['user_id'].shape[0], rank, world)
                shard = {k: v[s:e] for k, v in data.items()}
            cache[key] = (data, shard)  # keeps `data` alive: its id() is the key
        return cache[key][1]

    def _device_stream(self, which):
        """The train/test interaction stream resident in HBM as int32 (rng='device')."""
        if which not in self._dev_cache:
            data = self._rank_rows(self.data_processor.train_data if which == 'train'
                                   else self.data_processor.test_data)
            dev = _device()
            d = {'user': data['user_id'].to(torch.int32).to(dev), 'pos': data['pos_item_id'].to(torch.int32).to(dev)}
            d['neg'] = data['neg_item_id'].to(torch.int32).to(dev) if 'neg_item_id' in data else None
            tab = self.data_processor.item_meta_table
            d['item_meta'] = None if tab is None else torch.from_numpy(tab).to(torch.int32).to(dev)
            self._dev_cache[which] = d
        return self._dev_cache[which]

    def _sampler(self):
        """ops.Sampler of the neg_sampling options, or None = the reference's sampler: uniform over the
        items other than the row's positive (dataset/dataset.py:435-447).  reject_seen uses the TRAIN split's (user, item)
        pairs; popularity the train split's item frequencies."""
        ns = getattr(self, "neg_sampling", None)
        if not ns:
            return None
        if 'sampler' not in self._dev_cache:
            st = self._device_stream('train')
            seen = ops.Sampler.seen_csr(st['user'], st['pos'], self.n_users, self.n_items) if ns.get("reject_seen") else None
            self._dev_cache['sampler'] = ops.Sampler(k=ns.get("k", 1), popularity=ns.get("popularity", False), seen=seen,
                                                     stream_item=st['pos'], max_tries=ns.get("max_tries", 8))
        return self._dev_cache['sampler']

    def _eval_sampler(self):
        """evaluate(): the same candidate rules (reject_seen / popularity), every test row once."""
        sm = self._sampler()
        if sm is None or sm.k == 1:
            return sm
        if 'eval_sampler' not in self._dev_cache:
            ns = self.neg_sampling
            self._dev_cache['eval_sampler'] = ops.Sampler(k=1, popularity=ns.get("popularity", False), seen=sm.keep[0],
                                                          stream_item=sm.keep[1], max_tries=ns.get("max_tries", 8))
        return self._dev_cache['eval_sampler']

    def _item_meta_dev(self):
        tab = self.data_processor.item_meta_table
        if tab is None:
            return None
        if 'item_meta' not in self._dev_cache:
            self._dev_cache['item_meta'] = torch.from_numpy(tab).to(torch.int32).to(_device())
        return self._dev_cache['item_meta']

    def _make_trainer(self, optimizer, batch_size):
        if self.net_type == 'mlp':
            from .mlp_engine import MLPTrainer
            return MLPTrainer(self.net, optimizer, batch_size)
        return SparseScorerTrainer(self.net, optimizer, batch_size)

    # ------------------------------------------------------------------------------------------------ fit
    def make_runner(self, optimizer, batch_size):
        """The step-level driver fit() is built on (bench.py times exactly this object)."""
        return FitRunner(self, optimizer, batch_size)

    @_host_side
    def fit(self, optimizer, epochs=10, batch_size=512, profile_epochs: int = 0, sync_tables_every: int = 1,
            sync_bn: bool = False, loss: str = 'hinge'):
        """Fits the model (reference model.py:203-288).  Per step: [shuffle slice + negative sampling] -> fused
        gather + scoring + hinge + backward -> sparse-row optimiser update; the loss stays on the device and is
        read back once per epoch (the reference syncs every step, model.py:200).

        Under torch.distributed (one process per GPU, RCCL over xGMI) every rank trains its shard of the training
        split with `batch_size` per rank and NO per-step collective on the embedding rows:
          * dp_partition 'user' (default): the stream is cut by user_id % world, so a user row is only ever written by
            its owner — user tables are never averaged; the owners' rows are all-gathered ONCE at the end of fit()
            (c4: 5.1 GB table, 0.64 GB sent per rank).  Item / metadata tables (and BatchNorm running statistics) are
            averaged every `sync_tables_every` epochs (0 = never): c4 = 516 MB per all-reduce, ~6 ms on one xGMI ring
            against an epoch of ~100 ms per rank;
          * dp_partition 'contiguous': every table is averaged at that cadence (c4: 5.6 GB, ~64 ms: use 'user');
          * MLP: the dense gradients are all-reduced every step, layer by layer while the backward is still running;
            sync_bn=True takes the train-mode BatchNorm statistics over the GLOBAL batch (two all-reduces of 2*H floats
            per layer and pass), so N ranks with batch B reproduce one process with batch N*B on the dense path.
        The printed loss is the mean over ranks."""
        if self.net_type == 'mlp':
            self.net.compute.sync_bn = bool(sync_bn)
        # loss: 'hinge' = the reference's only loss (helper/loss.py:5-9, model.py:282); 'bpr' = -log sigmoid(pos - neg),
        # the alternative BASELINE.json's north_star names (evaluate() then reports that loss too)
        from ._lib import LOSS_ID
        if loss not in LOSS_ID:
            raise ValueError(f"loss must be one of {sorted(LOSS_ID)}")
        self.loss = loss
        runner = self.make_runner(optimizer, batch_size)
        runner.trainer.loss_id = LOSS_ID[loss]
        for epoch in range(epochs):
            self.net = self.net.train()
            prof = None
            if profile_epochs > 0 and epoch == 0:
                print(f"\n--- Starting Profiling for Epoch {epoch+1} ---")
                prof = torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU,
                                                          torch.profiler.ProfilerActivity.CUDA],
                                              record_shapes=True, profile_memory=True, with_stack=True)
                prof.__enter__()
            runner.more_epochs = epoch < epochs - 1  # lets the last slice's steps hide the next epoch's first presort
            runner.begin_epoch()
            runner.run_steps(runner.num_batches)
            if runner.more_epochs:
                runner.prepare_next_epoch()  # host work of epoch e+1 while the GPU runs epoch e
            avg_loss = runner.end_epoch()
            world = tdist.world_info()[1]
            if world > 1:
                avg_loss = tdist.allreduce_scalar_sum([avg_loss], _device())[0] / world
                if sync_tables_every and (epoch + 1) % sync_tables_every == 0:
                    self._sync_replicas()
            if prof is not None:
                prof.__exit__(None, None, None)
                print("--- Profiler Results (First Epoch) ---")
                print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=20))
            print(f'|--- Epoch {epoch+1}/{epochs} --- Training Loss: {avg_loss:.4f}')
        if tdist.world_info()[1] > 1 and self.dp_partition == 'user':
            for t in self._user_tables():  # every replica gets the owners' user rows
                tdist.gather_owned_rows_(t.data)

    def _user_tables(self):
        """Tables indexed by user id (Linear / FM: the embedding and the 1-wide term; MLP: the embedding)."""
        if hasattr(self.net, 'embedding_params'):
            return [self.net.embedding_params()[0]]
        ps = self.net.table_params()
        return [ps[0], ps[2]]

    def _sync_replicas(self):
        """Periodic re-synchronisation of the replicas: the mean over ranks of every embedding table that has more than
        one writer (all but the user tables under dp_partition 'user') and of the BatchNorm running statistics."""
        emb = self.net.embedding_params() if hasattr(self.net, 'embedding_params') else self.net.table_params()
        if self.dp_partition == 'user':
            owned = {id(p) for p in self._user_tables()}
            emb = [p for p in emb if id(p) not in owned]
        bufs = [b for b in self.net.buffers() if b.is_floating_point()]
        tdist.average_tables_([p.data for p in emb] + bufs)

    # ------------------------------------------------------------------------------------------------ evaluate
    @_host_side
    def evaluate(self, batch_size=512, eval_metrics=['loss', 'auc']):
        """reference model.py:292-338: eval-mode scores of the test split, hinge loss and pairwise AUC per batch,
        unweighted means over batches, printed; returns None."""
        self.net = self.net.eval()
        if self.data_processor.test_data.get('user_id', torch.empty(0)).numel() == 0:
            print("|--- No test data to evaluate.")
            return
        data = self._rank_rows(self.data_processor.test_data)
        n_test = data['user_id'].numel()
        dev = _device()
        loader = FastDataLoader(data=data, batch_size=batch_size, shuffle=False,
                                dynamic_neg_sampling=self.dynamic_neg_sampling, n_items=self.n_items,
                                item_to_metadata_map=self.data_processor.item_meta_table,
                                metadata_id_cols=self.metadata_name)
        nb = loader.num_batches
        loss_sums = torch.zeros(nb, dtype=torch.float32, device=dev)
        auc_counts = torch.zeros(nb, dtype=torch.int32, device=dev)
        if self.rng == 'reference':
            iter(loader)
            ep = self._host_epoch(data, loader)
        else:
            st = self._device_stream('test')
            sample_seed = _mix64(self.seed, 0xE7A1)
        # Linear / FM score triples independently of their batch: several batches per launch (ids, scores, per-batch
        # reductions, one id-range check per group); the MLP's activations are per batch
        group = 64 if hasattr(self.net, 'table_params') else 1
        group = max(1, min(group, (1 << 22) // max(batch_size, 1)))
        for b0 in range(0, nb, group):
            b1 = min(b0 + group, nb)
            s, e = b0 * batch_size, min(b1 * batch_size, n_test)
            if self.rng == 'reference':
                ids = {k: v[s:e] for k, v in ep.items()}
            else:
                ids = ops.batch_prepare(st['user'], st['pos'], st['neg'], 0, s, e - s, self.n_items, sample_seed, s,
                                        st['item_meta'], sampler=self._eval_sampler())
            pos, neg = self.net.score_ids(ids)
            from ._lib import LOSS_ID
            ops.hinge_auc_batches(pos, neg, batch_size, loss_sums[b0:b1], auc_counts[b0:b1],
                                  loss=LOSS_ID[getattr(self, "loss", "hinge")])
        ls, ac = loss_sums.cpu().numpy(), auc_counts.cpu().numpy()
        sizes = [min((b + 1) * batch_size, n_test) - b * batch_size for b in range(nb)]
        results = {}
        if 'loss' in eval_metrics:
            results['loss'] = [float(np.float32(ls[b]) / np.float32(sizes[b])) for b in range(nb)]
        if 'auc' in eval_metrics:
            results['auc'] = [float(np.float32(ac[b]) / np.float32(sizes[b])) for b in range(nb)]
        world = tdist.world_info()[1]
        for metric in eval_metrics:
            values = results.get(metric, [])
            if world > 1:  # unweighted mean over all ranks' batches
                tot, cnt = tdist.allreduce_scalar_sum([float(sum(values)), float(len(values))], dev)
                value = tot / cnt if cnt else 0
            else:
                value = sum(values) / len(values) if values else 0
            print(f'|--- Testing {metric}: {value:.4f}')

    # ------------------------------------------------------------------------------------------------ predict
    @_host_side
    def predict(self, user_id: int, top_k: int = 10, prediction_batch_size: int = 4096):
        """Top-K item ids for one user (reference model.py:341-452): score every item, sort descending, first top_k.
        Ties are ordered by ascending item id (unspecified in the reference).  Returns an int64 CPU tensor.
        `prediction_batch_size` is accepted for compatibility; the fused kernel streams the item table once and the
        result does not depend on it."""
        self.net = self.net.eval()
        scores = self.net.score_all_items(self._dense_user(user_id), self._item_meta_dev())
        k = min(int(top_k), self.n_items)
        if k <= 0:
            return torch.empty(0, dtype=torch.int64)
        return self._original_items(ops.topk(scores, k).cpu())

    def _dense_user(self, user_id):
        """Table row of a caller's user id (identity unless the ingest re-mapped ids, TensorProcessData(remap_ids=True))."""
        idx = getattr(self.data_processor, "user_index", None)
        if idx is None:
            return int(user_id)
        pos = int(torch.searchsorted(idx, torch.tensor(int(user_id), dtype=idx.dtype, device=idx.device)))
        if pos >= idx.numel() or int(idx[pos]) != int(user_id):
            raise IndexError(f"user id {user_id} does not occur in the ingested interactions")
        return pos

    def _original_items(self, rows):
        idx = getattr(self.data_processor, "item_index", None)
        return rows if idx is None else idx.cpu()[rows].to(torch.int64)


    @_host_side
    def predict_many(self, user_ids, top_k: int = 10):
        """Extension: predict() for several users — row r of the (len(user_ids), top_k) int64 CPU tensor
        equals `predict(user_ids[r], top_k)`; the per-user kernels are queued back to back and read back once."""
        self.net = self.net.eval()
        k = min(int(top_k), self.n_items)
        users = [self._dense_user(u) for u in (user_ids.tolist() if hasattr(user_ids, "tolist") else user_ids)]
        if k <= 0 or not users:
            return torch.empty((len(users), max(k, 0)), dtype=torch.int64)
        meta = self._item_meta_dev()
        out = torch.empty((len(users), k), dtype=torch.int64, device=_device())
        for r, u in enumerate(users):
            out[r] = ops.topk(self.net.score_all_items(u, meta), k)
        return self._original_items(out.cpu())


class FitRunner:
    """One training run at a fixed batch size: owns the optimiser plan, the staging buffers and the per-epoch batch
    feed.  begin_epoch() -> run_steps(k) (any number of calls) -> end_epoch()."""

    def __init__(self, model, optimizer, batch_size):
        self.m = model
        self.batch_size = batch_size
        self.dev = _device()
        self.data = model._rank_rows(model.data_processor.train_data)
        self.sampler = model._sampler() if model.rng == 'device' else None
        # k negatives per positive: the epoch visits every training row k times (k * N positions)
        self.n_train = self.data['user_id'].shape[0] * (self.sampler.k if self.sampler else 1)
        self.loader = FastDataLoader(data=self.data, batch_size=batch_size, shuffle=True,
                                     dynamic_neg_sampling=model.dynamic_neg_sampling, n_items=model.n_items,
                                     item_to_metadata_map=model.data_processor.item_meta_table,
                                     metadata_id_cols=model.metadata_name) if model.rng == 'reference' else None
        self.num_batches = int(math.ceil(self.n_train / batch_size)) if self.n_train > 0 else 0
        self.own_batches = self.num_batches
        if model.net_type == 'mlp' and tdist.world_info()[1] > 1:
            # the MLP all-reduces its dense gradients every step: all ranks run the common (minimum) number of steps per
            # epoch; the shard stays whole, so the positions beyond are other rows every epoch (the epoch shuffle)
            import torch.distributed as _d
            dev = self.dev if _d.get_backend() == 'nccl' else torch.device('cpu')
            self.num_batches = tdist.allreduce_min_int(self.num_batches, dev)
            if self.num_batches < self.own_batches:
                print(f'|--- data parallel: rank {tdist.world_info()[0]} runs {self.num_batches} of its '
                      f'{self.own_batches} steps per epoch (about '
                      f'{self.n_train - min(self.n_train, self.num_batches * batch_size)} of {self.n_train} positions '
                      'wait for another epoch\'s shuffle)')
        self.trainer = model._make_trainer(optimizer, min(batch_size, max(self.n_train, 1)))
        if getattr(self.trainer, "M", 0) > 0:
            self.trainer.item_meta = model._item_meta_dev()  # metadata scorers: the presort groups each column too
        self.trainer.sampler = self.sampler
        self.loss_sums = torch.zeros(max(self.num_batches, 1), dtype=torch.float32, device=self.dev)
        self.next_batch = 0
        self.ep = None
        self._next_ep = None
        self.prep_out = None
        # MLP (fused embedding update, device RNG): ids AND duplicate flags of 256 batches at a time from the sparse
        # regime's presort (trs_epoch_flags: the same triples trs_batch_prepare generates) instead of a prepare launch
        # per step — with the flags, the update's rows that are alone in their batch (94-97 % at c5) take plain
        # read-modify-writes instead of float atomics.  TRS_MLP_SLICE_FLAGS=0: a prepare launch per step, all atomics.
        self._mlp_ef, self._mlp_slice = None, None
        if (model.rng == 'device' and type(self.trainer).__name__ == "MLPTrainer" and
                getattr(self.trainer, "kind", None) == "sgd" and self.trainer._fused_embed_lr() is not None and
                batch_size <= ops.EpochFlags.MAX_BATCH and os.environ.get("TRS_MLP_SLICE_FLAGS", "1") != "0" and
                self.dev.type == "cuda" and self.n_train >= batch_size):
            self._mlp_ef = ops.EpochFlags(min(256, self.n_train // batch_size), batch_size, model.n_users,
                                          model.n_items, self.dev, ordered=False)

    def begin_epoch(self):
        m = self.m
        self.loss_sums.zero_()
        self.next_batch = 0
        self._slice = None
        self._epoch_no = getattr(self, "_epoch_no", 0) + 1
        if self.num_batches == 0:
            return
        if m.rng == 'reference':
            if self._next_ep is not None:  # drawn by prepare_next_epoch() while the previous epoch's steps ran
                self.ep, self._next_ep = self._next_ep, None
            else:
                iter(self.loader)  # reshuffle: one torch.randperm per epoch (dataset.py:369-373)
                self.ep = m._host_epoch(self.data, self.loader)
        else:
            self.st = m._device_stream('train')
            self.shuffle_key, self.sample_seed = self._epoch_keys(m._fit_epochs_done)

    def prepare_next_epoch(self):
        """Reference-RNG mode: draw the NEXT epoch's batches (shuffle + sampler, in the reference's RNG order — nothing
        else consumes the generators in between) right after the current epoch's steps were enqueued, so the host work
        overlaps the GPU instead of following the epoch's loss read-back.  Only fit() calls this, and only when another
        epoch follows: the generators are left exactly where the reference leaves them."""
        if self.m.rng == 'reference' and self.num_batches > 0 and self._next_ep is None:
            iter(self.loader)
            self._next_ep = self.m._host_epoch(self.data, self.loader)

    def _epoch_keys(self, epochs_done):
        """(shuffle key, sampler seed) of the device-RNG epoch that follows `epochs_done` finished ones."""
        seed = self.m.seed + 1000003 * tdist.world_info()[0]  # every rank draws its own negatives
        return _mix64(seed, 2 * epochs_done + 1), _mix64(seed, 2 * epochs_done + 2)

    def _presort(self, s0, full, prefetch=False, next_epoch=False):
        m, B, sl = self.m, self.batch_size, self.trainer.SLICE_BATCHES
        nb = min(sl, full - s0)
        tag = (self._epoch_no + int(next_epoch), s0, nb, B)
        if m.rng == 'device':
            sk, ss = self._epoch_keys(m._fit_epochs_done + 1) if next_epoch else (self.shuffle_key, self.sample_seed)
            return self.trainer.presort_slice(nb, B, self.st, sk, ss, s0 * B, tag=tag, prefetch=prefetch)
        assert not next_epoch  # the reference-RNG epoch is drawn from the host generators at begin_epoch
        return self.trainer.presort_slice(nb, B, given_ids=[self.ep[k_][s0 * B:(s0 + nb) * B]
                                                            for k_ in ('user', 'pos', 'neg')], tag=tag,
                                          prefetch=prefetch)

    def run_steps(self, k):
        """Run the next k steps of the current epoch (stops at the epoch's end).  Returns the number of steps run."""
        m, B = self.m, self.batch_size
        done = 0
        kind = getattr(self.trainer, "fast_kind", None)
        # the C step loop: SGD on every path, SparseAdam / Adagrad on the presorted path only
        has_meta = getattr(self.trainer, "M", 0) > 0  # metadata scorers: presorted path only (SGD)
        if os.environ.get("TRS_META_FAST", "1") == "0" and has_meta:
            kind = None  # tuning / fall-back knob: metadata scorers on the generic staged path
        fast = (kind == "sgd" and not has_meta) or (kind is not None and self.trainer.wants_presort(B))
        if fast and m.rng == 'reference':
            fast = self.ep['user'].dtype == torch.int32
        ops.stamp("run_steps:setup")
        if fast:  # whole batches, step loop in C (csrc/fast_step.hip): from the resident stream, or from the epoch's
            full = self.n_train // B  # host-prepared id arrays (bit-exact reference batches)
            presort = self.trainer.wants_presort(B)
            while done < k and self.next_batch < full:
                n = min(k - done, full - self.next_batch, 64)
                b = self.next_batch
                if presort:  # item references grouped by row per slice of batches (csrc/presort.hip)
                    sl = self.trainer.SLICE_BATCHES
                    if self._slice is None or not (self._slice[0] <= b < self._slice[0] + self._slice[1].n_batches):
                        s0 = (b // sl) * sl
                        self._slice = (s0, self._presort(s0, full))
                        if s0 + sl < full:  # next slice: sorted on a side stream while this slice's steps run
                            self._presort(s0 + sl, full, prefetch=True)
                        elif m.rng == 'device' and getattr(self, 'more_epochs', True):  # last slice: the next epoch's keys
                            # are known, start on its first slice
                            self._presort(0, full, prefetch=True, next_epoch=True)
                    s0, ps = self._slice
                    n = min(n, s0 + ps.n_batches - b)
                    ops.stamp("run_steps:before_fast_sorted_steps")
                    self.trainer.fast_sorted_steps(ps, b - s0, B, n, self.loss_sums[b:b + n],
                                                   m._item_meta_dev() if has_meta else None)
                elif m.rng == 'device' and self.sampler is None:
                    self.trainer.fast_stream_steps(self.st, self.shuffle_key, self.sample_seed, b * B, B, n,
                                                   self.loss_sums[b:b + n])
                elif m.rng == 'device':  # sampler options live in the presort / batch_prepare generators
                    break
                else:
                    self.trainer.fast_array_steps(self.ep, b * B, B, n, self.loss_sums[b:b + n])
                self.next_batch += n
                done += n
        while done < k and self.next_batch < self.num_batches:
            b = self.next_batch
            s, e = b * B, min((b + 1) * B, self.n_train)
            if m.rng == 'reference':
                ids = {key: v[s:e] for key, v in self.ep.items()}
            elif self._mlp_ef is not None and e - s == B:
                ids = self._mlp_slice_ids(b)
            else:
                st = self.st
                out = self.prep_out if (self.prep_out is not None and e - s == B) else None
                ids = ops.batch_prepare(st['user'], st['pos'], st['neg'], self.shuffle_key, s, e - s, m.n_items,
                                        self.sample_seed, s, st['item_meta'], out, sampler=self.sampler)
                if e - s == B:
                    self.prep_out = ids
            self.trainer.step(ids, self.loss_sums[b:b + 1])
            self.next_batch += 1
            done += 1
        return done

    def _mlp_slice_ids(self, b):
        """ids + duplicate flags of whole batch b as views of the current 256-batch slice (generated on the launch
        stream when the epoch enters the slice: 0.8 ms per 256 steps of >= 1.5 ms each)."""
        ef, B, st = self._mlp_ef, self.batch_size, self.st
        cur = self._mlp_slice
        if cur is None or cur[0] != self._epoch_no or not (cur[1] <= b < cur[1] + cur[2]):
            s0 = (b // ef.n_batches) * ef.n_batches
            nb = min(ef.n_batches, self.n_train // B - s0)
            if "ui" not in st:
                st["ui"] = ops.interleave_stream(st["user"], st["pos"])
            ef.run(st["ui"], st["neg"], self.shuffle_key, self.sample_seed, s0 * B, self.trainer.err,
                   sampler=self.sampler, n_batches=nb)  # (the epoch's last slice may be shorter)
            meta = None
            if st.get("item_meta") is not None:
                n = nb * B
                meta = (st["item_meta"][ef.ids[1][:n].long()], st["item_meta"][ef.ids[2][:n].long()])
            cur = self._mlp_slice = (self._epoch_no, s0, nb, meta)
        o = (b - cur[1]) * B
        ids = {"user": ef.ids[0][o:o + B], "pos": ef.ids[1][o:o + B], "neg": ef.ids[2][o:o + B],
               "user_dup": ef.user_dup[o:o + B], "item_dup": ef.item_dup[o:o + B]}
        if cur[3] is not None:
            ids["pos_meta"], ids["neg_meta"] = cur[3][0][o:o + B], cur[3][1][o:o + B]
        return ids

    def touch_host_path(self):
        """Walk the host side of the next run_steps() call with ZERO steps: the same Python code and the same C entry
        point, no kernel launch and no state change (the C step loop with n_steps = 0 returns at once; a pending slice
        switch is left to the real call).  For callers that time a SHORT window right after a synchronise (bench.py with
        the driver's 20 steps): the thread wakes from the wait with cold caches and its first pass through this code
        takes ~75 us instead of ~12 (host time stamps, TRS_BENCH_TIMELINE=1) — 9 % of such a window, nothing in a
        training run of thousands of steps.  Returns True if the path was walked."""
        m, B = self.m, self.batch_size
        tr = self.trainer
        if getattr(tr, "fast_kind", None) != "sgd" or getattr(tr, "M", 0) > 0 or not tr.wants_presort(B):
            return False
        b = self.next_batch
        if self._slice is None or not (self._slice[0] <= b < self._slice[0] + self._slice[1].n_batches):
            return False
        s0, ps = self._slice
        tr.fast_sorted_steps(ps, b - s0, B, 0, self.loss_sums[b:b + 1])  # (zero steps: the slot is not written)
        return True

    def end_epoch(self):
        """Sync once, check the id-range flag, return the reference's epoch loss (unweighted mean of batch means)."""
        m, B = self.m, self.batch_size
        m._fit_epochs_done += 1
        sums = self.loss_sums.cpu().numpy()
        self.trainer.check_errors()
        total = 0.0
        for b in range(self.next_batch):
            nb = min((b + 1) * B, self.n_train) - b * B
            total += float(np.float32(sums[b]) / np.float32(nb))
        return total / self.next_batch if self.next_batch > 0 else 0
