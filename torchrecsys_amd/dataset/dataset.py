# -*- coding: utf-8 -*-
"""Data pipeline of the path (reference dataset/dataset.py): DataFrame -> id tensors, train/test split, static
negatives, and the batch iterator with the dynamic negative sampler.

Index streams are in the bit-exact parity class: the same global RNG streams are consumed in the same order as the
reference (numpy legacy `np.random` for negatives, RandomState(42) for the split, torch's CPU generator for the
shuffle), so a seeded run yields the very same batches.  The per-row Python sampler loop of the reference
(dataset.py:435-447) is replaced by a vectorised walk over the same stream.
"""
import ast
import json
import math
from typing import List

import numpy as np
import pandas as pd
import torch


def _scalar_meta(x):
    """One categorical id per row: ints, 1-element lists, or their string encodings ('3', '[3]')."""
    if isinstance(x, str):
        x = ast.literal_eval(x)
    if isinstance(x, (list, tuple, np.ndarray)):
        if len(x) != 1:
            raise NotImplementedError(
                "multi-valued metadata lists are not supported: the scorers index one embedding row per metadata "
                "column ((B, M) id contract, reference collaborative/linear.py:72-75); pass one categorical id per row")
        x = x[0]
    return int(x)


class Data:
    """reference dataset/dataset.py:15-118."""

    def __init__(self, dataset: pd.DataFrame, user_id_col: str, item_id_col: str, metadata_id_col: List[str] = None,
                 split_ratio: float = 0.8, dynamic_neg_sampling: bool = False):
        self.dataset = dataset
        self.user_id = user_id_col
        self.item_id = item_id_col
        self._users = np.asarray(dataset[user_id_col].values)
        self._items = np.asarray(dataset[item_id_col].values)
        self.num_items = len(np.unique(self._items))   # dataset.py:30
        self.num_users = len(np.unique(self._users))   # dataset.py:31
        self.dynamic_neg_sampling = dynamic_neg_sampling
        self._neg_items = None
        if not self.dynamic_neg_sampling:
            self._neg_items = self._get_negative_items()
        self.split_ratio = split_ratio
        if metadata_id_col:
            self.metadata_id = list(metadata_id_col)
            self._meta = np.stack([np.asarray([_scalar_meta(v) for v in dataset[c].values], dtype=np.int64)
                                   for c in self.metadata_id], axis=1)
            self.negative_metadata_id = self._get_negative_metadata_column_names()
            self.metadata_size = {c: len(np.unique(self._meta[:, k])) for k, c in enumerate(self.metadata_id)}

    def _get_negative_items(self):
        """Static negatives: ONE np.random.randint(0, num_items, size=N) from the global legacy stream, no rejection of
        neg == pos, drawn before the split (dataset.py:56-64).  The caller's DataFrame is not mutated."""
        return np.random.randint(low=0, high=self.num_items, size=len(self._items))

    def _get_negative_metadata_column_names(self):
        return ["neg_" + c for c in self.metadata_id]


class ProcessData(Data):
    """reference dataset/dataset.py:121-316.  `train_data` / `test_data` are dicts of int64 CPU tensors with keys
    user_id, pos_item_id[, neg_item_id][, pos_metadata_id (N,M)][, neg_metadata_id (N,M)]."""

    def __init__(self, dataset: pd.DataFrame, user_id_col: str, item_id_col: str, metadata_id_col: List[str] = None,
                 split_ratio: float = 0.9, dynamic_neg_sampling: bool = False):
        super().__init__(dataset, user_id_col, item_id_col, metadata_id_col, split_ratio, dynamic_neg_sampling)
        self.user_id_col = user_id_col
        self.item_id_col = item_id_col
        if metadata_id_col:
            self.metadata_id_col = list(metadata_id_col)

    def prepare_data(self):
        N = len(self._users)
        has_meta = hasattr(self, "metadata_id_col") and bool(self.metadata_id_col)
        self.config = {"num_users": self.num_users, "num_items": self.num_items,
                       "num_metadata": self.metadata_size if has_meta else {}}
        for name, ids, n in (("user", self._users, self.num_users), ("item", self._items, self.num_items)):
            if N and (ids.min() < 0 or ids.max() >= n):
                raise IndexError(f"{name} ids must be dense 0..{n - 1} (the tables are sized by the number of unique "
                                 f"ids, dataset.py:30-31, and indexed by the raw values, :268-269); found "
                                 f"[{ids.min()}, {ids.max()}]")
        self.item_to_metadata_map = None
        self.item_meta_table = None
        self.meta_data_df = None
        if has_meta:
            # item -> its metadata ids (first occurrence wins), the device-friendly form of item_to_metadata_map
            table = np.zeros((self.num_items, len(self.metadata_id_col)), dtype=np.int64)
            first = np.unique(self._items, return_index=True)[1]
            table[self._items[first]] = self._meta[first]
            self.item_meta_table = table
            self.meta_data_df = pd.DataFrame({"pos_item_id": np.arange(self.num_items),
                                              **{c: table[:, k] for k, c in enumerate(self.metadata_id_col)}})
            self.item_to_metadata_map = {int(i): {c: [int(table[i, k])] for k, c in enumerate(self.metadata_id_col)}
                                         for i in range(self.num_items)}
        # split: sklearn.train_test_split(test_size=1-split_ratio, random_state=42) restated (dataset.py:239-240)
        if self.split_ratio < 1:
            n_test = int(math.ceil((1 - self.split_ratio) * N))
            perm = np.random.RandomState(42).permutation(N)
            tr, te = perm[n_test:], perm[:n_test]
        else:
            tr, te = np.arange(N), np.arange(0)
        self.train_data = self._rows_to_tensor_dict(tr, has_meta)
        self.test_data = self._rows_to_tensor_dict(te, has_meta)

    def _rows_to_tensor_dict(self, rows, has_meta):
        d = {"user_id": torch.from_numpy(self._users[rows]).long(),
             "pos_item_id": torch.from_numpy(self._items[rows]).long()}
        if not self.dynamic_neg_sampling:
            d["neg_item_id"] = torch.from_numpy(self._neg_items[rows]).long()
        if has_meta:
            d["pos_metadata_id"] = torch.from_numpy(self._meta[rows]).long()
            if not self.dynamic_neg_sampling:
                d["neg_metadata_id"] = torch.from_numpy(self.item_meta_table[self._neg_items[rows]]).long()
        return d

    def write_data(self, path: str):
        with open(f"{path}/config.json", "w") as file:
            json.dump(self.config, file)
        if self.meta_data_df is not None:
            self.meta_data_df.to_csv(f"{path}/meta.csv", index=False)


class TensorProcessData:
    """Tensor-native ingest: the same `train_data` / `test_data` / `config` surface as ProcessData, built
    from id tensors without a DataFrame.  At 1e8-1e9 rows the pandas copies of the reference (dataset.py:162,240)
    dominate start-up and host memory; here ids go straight to int32 in HBM.

    user_ids, item_ids : (N,) integer tensors, CPU or GPU
    item_metadata      : optional (n_items, M) integer tensor — the metadata ids of every item ((B,M) contract)
    remap_ids          : False (default, the reference's semantics: ids index the tables raw, dataset.py:30-31,268-269 —
                         an id >= the table size raises IndexError HERE, at ingest, not in the first kernel) or True:
                         arbitrary ids are mapped to dense 0..n-1 by rank (sorted unique values; `user_index` /
                         `item_index` keep the original ids, TorchRecSys.predict translates both ways)
    split              : 'reference' (default up to 2^31-1 rows): the reference's split — sklearn train_test_split(
                         random_state=42) = np.random.RandomState(42).permutation(N), first ceil((1-ratio) N) rows test
                         (dataset.py:237-249) — for CPU AND GPU tensors: the permutation is drawn on the host (8 B per
                         row of host memory, ~0.1 s per 10M rows), uploaded as indices and applied on the device, so a
                         GPU-resident stream is cut into exactly the rows the reference would train on.  Static
                         negatives likewise come from the reference's single legacy-stream draw (dataset.py:56-64).
                         'device': a seeded torch.randperm on the GPU (same proportions, other rows; no host pass)."""

    def __init__(self, user_ids, item_ids, n_users=None, n_items=None, item_metadata=None, metadata_names=None,
                 split_ratio=0.8, dynamic_neg_sampling=False, remap_ids=False, split=None):
        assert user_ids.dim() == 1 and user_ids.shape == item_ids.shape
        assert split in (None, 'reference', 'device')
        self.dynamic_neg_sampling = dynamic_neg_sampling
        self.split_ratio = split_ratio
        N = user_ids.shape[0]
        self.split = split or ('reference' if N < 2 ** 31 else 'device')
        if self.split == 'reference' and N >= 2 ** 31:
            raise ValueError("split='reference' replays numpy's permutation on the host: at most 2^31-1 rows")
        self.user_index = self.item_index = None
        if remap_ids:  # dense ids by rank of the original value
            self.user_index, user_ids = torch.unique(user_ids, sorted=True, return_inverse=True)
            self.item_index, item_ids = torch.unique(item_ids, sorted=True, return_inverse=True)
            if item_metadata is not None and item_metadata.shape[0] != self.item_index.numel():
                item_metadata = item_metadata[self.item_index.to(item_metadata.device).long()]  # rows of the items present
            n_users = n_users if n_users is not None and n_users >= self.user_index.numel() else self.user_index.numel()
            n_items = n_items if n_items is not None and n_items >= self.item_index.numel() else self.item_index.numel()
        if N:  # one reduction per column, on the device the ids live on
            u_lo, u_hi = int(user_ids.min()), int(user_ids.max())
            i_lo, i_hi = int(item_ids.min()), int(item_ids.max())
        else:
            u_lo = i_lo = 0
            u_hi = i_hi = -1
        self.num_users = int(n_users) if n_users is not None else u_hi + 1
        self.num_items = int(n_items) if n_items is not None else i_hi + 1
        if u_lo < 0 or i_lo < 0 or u_hi >= self.num_users or i_hi >= self.num_items:
            raise IndexError(f"index out of range in self (ingest: user ids span [{u_lo}, {u_hi}] for a table of "
                             f"{self.num_users} rows, item ids [{i_lo}, {i_hi}] for {self.num_items}; ids must be dense "
                             f"0..n-1 as in the reference, dataset/dataset.py:30-31,268-269 — or pass remap_ids=True)")
        self._users, self._items = user_ids, item_ids
        self._neg = None
        if not dynamic_neg_sampling:
            if user_ids.is_cuda and self.split == 'device':
                g = torch.Generator(device=user_ids.device)
                g.manual_seed(int(np.random.randint(0, 2 ** 31 - 1)))
                self._neg = torch.randint(0, self.num_items, (N,), device=user_ids.device, dtype=item_ids.dtype,
                                          generator=g)
            else:  # the reference's single legacy-stream draw (dataset.py:56-64)
                self._neg = torch.from_numpy(np.random.randint(low=0, high=self.num_items, size=N)).to(
                    item_ids.dtype).to(item_ids.device)
        self.item_meta_table = None
        self.metadata_id_col = None
        self.metadata_size = {}
        if item_metadata is not None:
            tab = item_metadata.cpu().numpy().astype(np.int64)
            assert tab.shape[0] == self.num_items
            self.item_meta_table = tab
            self.metadata_id_col = list(metadata_names) if metadata_names else [f"meta_{m}" for m in range(tab.shape[1])]
            self.metadata_size = {c: int(tab[:, k].max()) + 1 for k, c in enumerate(self.metadata_id_col)}
        self.item_to_metadata_map = None
        self.meta_data_df = None

    def prepare_data(self):
        N = self._users.shape[0]
        self.config = {"num_users": self.num_users, "num_items": self.num_items, "num_metadata": self.metadata_size}
        if self.split_ratio < 1:
            n_test = int(math.ceil((1 - self.split_ratio) * N))
            if self.split == 'device' and self._users.is_cuda:
                g = torch.Generator(device=self._users.device)
                g.manual_seed(42)
                perm = torch.randperm(N, device=self._users.device, generator=g)
            else:  # the reference's permutation, drawn where numpy lives and applied where the ids live
                perm = torch.from_numpy(np.random.RandomState(42).permutation(N))
                if self._users.is_cuda:
                    perm = (perm.to(torch.int32) if N < 2 ** 31 else perm).to(self._users.device).long()
            tr, te = perm[n_test:], perm[:n_test]
        else:
            tr = torch.arange(N, device=self._users.device)
            te = tr[:0]
        self.train_data, self.test_data = self._rows(tr), self._rows(te)
        self._users = self._items = self._neg = None  # the split copies are the only ones kept

    def _rows(self, rows):
        d = {"user_id": self._users[rows], "pos_item_id": self._items[rows]}
        if self._neg is not None:
            d["neg_item_id"] = self._neg[rows]
        if self.item_meta_table is not None and not self._items.is_cuda:  # device streams look metadata up per batch
            tab = torch.from_numpy(self.item_meta_table).to(self._items.device)
            d["pos_metadata_id"] = tab[d["pos_item_id"].long()]
            if self._neg is not None:
                d["neg_metadata_id"] = tab[d["neg_item_id"].long()]
        return d


def sample_negatives_reference_stream(pos_item_ids: np.ndarray, n_items: int) -> np.ndarray:
    """The dynamic sampler of dataset.py:435-447 as a vectorised walk over the SAME global legacy numpy stream: row k
    takes the next stream value that differs from its own positive.  Scalar `np.random.randint(0, n)` calls consume the
    stream exactly like `randint(0, n, size=k)`, so the output and the number of draws equal the reference's."""
    pos = np.asarray(pos_item_ids, dtype=np.int64)
    B = pos.size
    out = np.empty(B, dtype=np.int64)
    k = 0
    while k < B:
        draws = np.random.randint(0, n_items, size=B - k)
        # without collisions row k+i takes draws[i]; the first collision shifts everything behind it by one
        start = 0
        while start < draws.size and k < B:
            m = min(draws.size - start, B - k)
            hit = np.nonzero(draws[start:start + m] == pos[k:k + m])[0]
            if hit.size == 0:
                out[k:k + m] = draws[start:start + m]
                k += m
                start += m
            else:
                h = int(hit[0])
                out[k:k + h] = draws[start:start + h]
                k += h
                start += h + 1  # the colliding draw is consumed, its row is not finished
    return out


class FastDataLoader:
    """Batch iterator of the reference (dataset/dataset.py:319-458): randperm shuffle (once in the constructor, once
    per __iter__), contiguous slices, last batch partial, optional dynamic negatives."""

    def __init__(self, data: dict, batch_size: int = 32, shuffle: bool = False, dynamic_neg_sampling: bool = False,
                 n_items: int = None, item_to_metadata_map=None, metadata_id_cols: List[str] = None):
        self.data = data
        self.batch_size = batch_size
        self.shuffle = shuffle
        self.dynamic_neg_sampling = dynamic_neg_sampling
        self.n_items = n_items
        self.item_to_metadata_map = item_to_metadata_map
        self.metadata_id_cols = metadata_id_cols
        if self.dynamic_neg_sampling and self.n_items is None:
            raise ValueError("n_items must be provided for dynamic negative sampling.")
        if self.dynamic_neg_sampling and self.metadata_id_cols and self.item_to_metadata_map is None:
            raise ValueError("item_to_metadata_map must be provided for dynamic negative sampling with metadata.")
        self._meta_table = self._as_meta_table(item_to_metadata_map, metadata_id_cols)
        self.dataset_len = 0
        if "user_id" in self.data and isinstance(self.data["user_id"], torch.Tensor):
            self.dataset_len = self.data["user_id"].shape[0]
        if self.shuffle and self.dataset_len > 0:
            self.shuffle_indices()
        self.num_batches = int(np.ceil(self.dataset_len / self.batch_size)) if self.dataset_len > 0 else 0

    def _as_meta_table(self, mapping, cols):
        """(n_items, M) int64 lookup from either an array/tensor or the reference's {item: {col: [id]}} dict."""
        if mapping is None or not cols:
            return None
        if isinstance(mapping, dict):
            n = max(mapping.keys()) + 1 if mapping else 0
            n = max(n, self.n_items or 0)
            table = np.zeros((n, len(cols)), dtype=np.int64)
            for item, meta in mapping.items():
                for k, c in enumerate(cols):
                    v = meta.get(c, [])
                    table[item, k] = _scalar_meta(v) if (not isinstance(v, list) or len(v)) else 0
            return table
        return np.asarray(mapping, dtype=np.int64)

    def shuffle_indices(self):
        if self.dataset_len == 0:
            return
        self.indices = torch.randperm(self.dataset_len)

    def __len__(self):
        return self.num_batches

    def __iter__(self):
        self.i = 0
        if self.shuffle and self.dataset_len > 0:
            self.shuffle_indices()
        return self

    def epoch_order(self):
        """Row positions in the order this epoch visits them (after __iter__)."""
        return self.indices if (self.shuffle and self.dataset_len > 0) else torch.arange(self.dataset_len)

    def __next__(self):
        if self.i >= self.dataset_len:
            raise StopIteration
        end = min(self.i + self.batch_size, self.dataset_len)
        sel = self.indices[self.i:end] if (self.shuffle and self.dataset_len > 0) else slice(self.i, end)
        batch = {k: t[sel] for k, t in self.data.items() if k in ("user_id", "pos_item_id", "pos_metadata_id")}
        if not self.dynamic_neg_sampling:
            for k in ("neg_item_id", "neg_metadata_id"):
                if k in self.data:
                    batch[k] = self.data[k][sel]
        else:
            neg = sample_negatives_reference_stream(batch["pos_item_id"].numpy(), self.n_items)
            batch["neg_item_id"] = torch.from_numpy(neg)
            if self._meta_table is not None and "pos_metadata_id" in batch:
                batch["neg_metadata_id"] = torch.from_numpy(self._meta_table[neg])
        self.i += self.batch_size
        return batch
