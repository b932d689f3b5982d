# -*- coding: utf-8 -*-
"""SURVEY §8f-2 on the GPU: tensor-native ingest (dataset.TensorProcessData) — the reference's RandomState(42) split
applied to GPU-resident id tensors, and dense re-mapping of arbitrary ids (remap_ids=True) through fit() / evaluate() /
predict().  Reference behaviour matched / deviated from: dataset/dataset.py:30-31,268-269 (raw ids index the tables),
:236-242 (train_test_split(random_state=42))."""
import contextlib
import io

import numpy as np
import pandas as pd
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _frame(n_u=300, n_i=120, n=6000, seed=0):
    rs = np.random.RandomState(seed)
    return pd.DataFrame({"user_id": np.concatenate([np.arange(n_u), rs.randint(0, n_u, n - n_u)]),
                         "item_id": np.concatenate([np.arange(n_i), rs.randint(0, n_i, n - n_i)])})


@pytest.mark.parametrize("dyn", [True, False])
def test_gpu_resident_split_is_the_reference_split(dyn):
    """TensorProcessData(split='reference') on GPU tensors cuts the stream into exactly the rows ProcessData (the
    DataFrame ingest, pinned to the reference's split by the G3 goldens) trains and tests on — same rows, same order,
    static negatives included (the reference's single legacy-stream draw)."""
    from torchrecsys_amd.dataset.dataset import ProcessData, TensorProcessData
    df = _frame()
    np.random.seed(1)
    a = ProcessData(df, "user_id", "item_id", split_ratio=0.8, dynamic_neg_sampling=dyn)
    a.prepare_data()
    np.random.seed(1)
    u = torch.from_numpy(df["user_id"].values).to(DEV)
    i = torch.from_numpy(df["item_id"].values).to(DEV)
    b = TensorProcessData(u, i, split_ratio=0.8, dynamic_neg_sampling=dyn, split="reference")
    b.prepare_data()
    assert b.config == a.config
    keys = ("user_id", "pos_item_id") + (() if dyn else ("neg_item_id",))
    for k in keys:
        assert b.train_data[k].is_cuda and b.test_data[k].is_cuda
        assert torch.equal(a.train_data[k], b.train_data[k].cpu().long()), k
        assert torch.equal(a.test_data[k], b.test_data[k].cpu().long()), k
    assert ("neg_item_id" in b.train_data) == (not dyn)
    # int32 ids on the device (what the resident stream holds) take the same path
    c = TensorProcessData(u.to(torch.int32), i.to(torch.int32), split_ratio=0.8, dynamic_neg_sampling=True,
                          split="reference")
    c.prepare_data()
    assert torch.equal(c.train_data["user_id"].cpu().long(), a.train_data["user_id"])
    # split='device' keeps the proportions (other rows: a GPU permutation, no host pass)
    d = TensorProcessData(u, i, split_ratio=0.8, dynamic_neg_sampling=True, split="device")
    d.prepare_data()
    assert d.train_data["user_id"].numel() == a.train_data["user_id"].numel()
    assert torch.equal(torch.cat([d.train_data["user_id"], d.test_data["user_id"]]).sort().values.cpu(),
                       torch.from_numpy(np.sort(df["user_id"].values)))


@pytest.mark.parametrize("net_type", ["fm", "linear", "mlp"])
def test_remap_ids_trains_and_predicts_in_original_ids(net_type):
    """Non-dense ids (users 3 + 7k, items 11 + 5k: the reference would index its tables with them raw and raise
    `index out of range`, dataset.py:30-31,268-269) with remap_ids=True: the run IS the dense run — same tables after
    fit(), same printed metrics — and predict() takes the ORIGINAL user id and returns ORIGINAL item ids; without
    remap_ids the ingest raises IndexError; a user id that never occurred raises at predict()."""
    from torchrecsys_amd.model import TorchRecSys
    df = _frame()
    dense_u, dense_i = torch.from_numpy(df["user_id"].values).to(DEV), torch.from_numpy(df["item_id"].values).to(DEV)
    raw_u, raw_i = dense_u * 7 + 3, dense_i * 5 + 11  # monotone: rank of a raw id == its dense id

    def run(u, i, **kw):
        np.random.seed(4)
        torch.manual_seed(4)
        buf = io.StringIO()
        extra = {"hidden_layers": [32, 16]} if net_type == "mlp" else {}
        with contextlib.redirect_stdout(buf):
            m = TorchRecSys.from_tensors(u, i, n_factors=16, net_type=net_type, dynamic_neg_sampling=True, rng="device",
                                         seed=9, **extra, **kw)
            m.fit(torch.optim.SGD(m.parameters(), lr=0.05), epochs=2, batch_size=256)
            m.evaluate(batch_size=256)
        return m, [ln for ln in buf.getvalue().splitlines() if ln.startswith("|---")]

    with pytest.raises(IndexError, match="remap_ids"):
        with contextlib.redirect_stdout(io.StringIO()):
            TorchRecSys.from_tensors(raw_u, raw_i, n_users=300, n_items=120, n_factors=16, net_type=net_type,
                                     dynamic_neg_sampling=True)  # (tables sized by the caller: 300 users, 120 items)
    dense, lines_d = run(dense_u, dense_i)
    remap, lines_r = run(raw_u, raw_i, remap_ids=True)
    assert remap.n_users == dense.n_users == 300 and remap.n_items == dense.n_items == 120
    assert torch.equal(remap.data_processor.user_index.cpu(), torch.arange(300) * 7 + 3)
    assert torch.equal(remap.data_processor.item_index.cpu(), torch.arange(120) * 5 + 11)
    # the same dense stream -> the same run, up to the order of the float atomics on duplicated rows (1e-6 on the
    # scorers without dense layers; the MLP + BatchNorm trajectory amplifies that last bit)
    tol = 2e-6 if net_type != "mlp" else 5e-3
    for (k, a), (_, b) in zip(sorted(dense.state_dict().items()), sorted(remap.state_dict().items())):
        assert torch.allclose(a.float(), b.float(), rtol=0, atol=tol), k
    assert len(lines_d) == len(lines_r) == 4
    val = lambda ln: float(ln.split(":")[-1])
    for a, b in zip(lines_d, lines_r):
        assert a.split(":")[0] == b.split(":")[0] and abs(val(a) - val(b)) <= (1e-4 if net_type != "mlp" else 5e-3), (a, b)
    remap.load_state_dict(dense.state_dict())  # identical weights: predict() must agree exactly, ids translated
    for user in (0, 17, 299):
        want = dense.predict(user, top_k=10)
        got = remap.predict(user * 7 + 3, top_k=10)
        assert got.dtype == torch.int64 and torch.equal(got, want * 5 + 11)
    many = remap.predict_many([3, 3 + 7 * 42], top_k=5)
    assert torch.equal(many[1], remap.predict(3 + 7 * 42, top_k=5))
    assert bool(((many - 11) % 5 == 0).all())
    with pytest.raises(IndexError, match="does not occur"):
        remap.predict(4)  # between two known raw ids


def test_remap_ids_with_metadata_rows_follow_the_items_present():
    """remap_ids with an item-metadata table given for the ORIGINAL id space: the rows of the items present are kept,
    in dense order, and the metadata scorer trains / predicts through them."""
    from torchrecsys_amd.model import TorchRecSys
    rs = np.random.RandomState(2)
    n_raw_items = 400
    present = np.sort(rs.choice(n_raw_items, 90, replace=False))
    items = torch.from_numpy(np.concatenate([present, present[rs.randint(0, 90, 3000)]])).to(DEV)
    users = torch.from_numpy(np.concatenate([np.arange(90) * 3, rs.randint(0, 200, 3000) * 3])).to(DEV)
    meta_raw = torch.from_numpy(rs.randint(0, 6, (n_raw_items, 1))).to(DEV)
    with contextlib.redirect_stdout(io.StringIO()):
        np.random.seed(3)
        torch.manual_seed(3)
        m = TorchRecSys.from_tensors(users, items, item_metadata=meta_raw, n_factors=16, net_type="fm",
                                     dynamic_neg_sampling=True, rng="device", remap_ids=True)
        m.fit(torch.optim.SGD(m.parameters(), lr=0.05), epochs=1, batch_size=128)
    assert m.n_items == 90
    assert np.array_equal(m.data_processor.item_meta_table[:, 0], meta_raw.cpu().numpy()[present, 0])
    top = m.predict(int(users[5]), top_k=7)
    assert set(top.tolist()) <= set(present.tolist()) and top.numel() == 7


@pytest.mark.parametrize("meta", [False, True])
def test_mlp_fit_takes_ids_and_flags_from_presort_slices(meta, monkeypatch):
    """The MLP's device-RNG fit() draws its batches and their duplicate flags per slice from trs_epoch_flags (lone
    embedding rows: plain read-modify-writes) — the same triples trs_batch_prepare generates step by step
    (TRS_MLP_SLICE_FLAGS=0: that path, float atomics for every row): same losses, same tables up to the order of the
    float atomics on duplicated rows, flags exact for these table sizes, metadata ids looked up per slice."""
    from torchrecsys_amd.model import TorchRecSys
    df = _frame(n_u=3000, n_i=1500, n=40_000, seed=5)
    u, i = torch.from_numpy(df["user_id"].values).to(DEV), torch.from_numpy(df["item_id"].values).to(DEV)
    rs = np.random.RandomState(1)
    item_meta = torch.from_numpy(rs.randint(0, 9, (1500, 2))).to(DEV) if meta else None

    def run(slices):
        monkeypatch.setenv("TRS_MLP_SLICE_FLAGS", "1" if slices else "0")
        np.random.seed(4)
        torch.manual_seed(4)
        with contextlib.redirect_stdout(io.StringIO()):
            m = TorchRecSys.from_tensors(u, i, item_metadata=item_meta, n_factors=32, net_type="mlp",
                                         hidden_layers=[64, 32], dynamic_neg_sampling=True, rng="device", seed=11)
            opt = torch.optim.SGD(m.parameters(), lr=0.05)
            r = m.make_runner(opt, 1024)
            assert (r._mlp_ef is not None) == slices
            if slices and meta:  # slices of 8 batches: 31 whole batches = 3 slices + a shorter last one
                from torchrecsys_amd import ops
                r._mlp_ef = ops.EpochFlags(8, 1024, 3000, 1500, DEV, ordered=False)
            m.net.train()
            losses = []
            for _ in range(2):
                r.begin_epoch()
                while r.run_steps(7):
                    pass
                losses.append(r.end_epoch())
        if slices:  # the slice in use: flags == "the row occurs again in this batch" (tables fit the bitmap: exact)
            ef, (_, s0, nb, _m) = r._mlp_ef, r._mlp_slice
            for b in range(nb):
                sl = slice(b * 1024, (b + 1) * 1024)
                uu, pp, nn = (t[sl].long() for t in ef.ids)
                assert torch.equal(ef.user_dup[sl].bool(), torch.bincount(uu, minlength=3000)[uu] > 1)
                cnt = torch.bincount(torch.cat([pp, nn]), minlength=1500)
                assert torch.equal(ef.item_dup[sl].bool(), torch.stack([cnt[pp] > 1, cnt[nn] > 1], 1))
        return m, losses
    a, la = run(True)
    b, lb = run(False)
    for x, y in zip(la, lb):
        assert abs(x - y) <= 5e-3 * max(abs(y), 1e-3), (la, lb)
    for (k, va), (_, vb) in zip(sorted(a.state_dict().items()), sorted(b.state_dict().items())):
        assert torch.allclose(va.float(), vb.float(), rtol=0, atol=5e-3), k
