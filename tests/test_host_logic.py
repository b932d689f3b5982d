# -*- coding: utf-8 -*-
"""Host-side logic (no GPU): data pipeline invariants of the reference's own tests, optimiser dispatch, loud failure
without a device."""
import contextlib
import io
import os

import numpy as np
import pandas as pd
import pytest
import torch

from oracle import loader as oloader
from torchrecsys_amd.dataset.dataset import (FastDataLoader, ProcessData, TensorProcessData,
                                            sample_negatives_reference_stream)

N_USERS, N_ITEMS, N = 100, 50, 1000


def dummy_df(with_meta=False, seed=0):
    rs = np.random.RandomState(seed)
    df = pd.DataFrame({"user_id": np.concatenate([np.arange(N_USERS), rs.randint(0, N_USERS, N - N_USERS)]),
                       "item_id": np.concatenate([np.arange(N_ITEMS), rs.randint(0, N_ITEMS, N - N_ITEMS)])})
    if with_meta:
        cat = rs.randint(0, 5, N_ITEMS)
        cat[:5] = np.arange(5)
        df["category"] = cat[df["item_id"].values]
        df["brand"] = [f"[{(i * 7) % 3}]" for i in df["item_id"].values]  # string-encoded 1-element lists
    return df


def test_process_data_static_neg_sampling():  # reference tests/test_model_and_features.py:52-59
    dp = ProcessData(dummy_df(), "user_id", "item_id", dynamic_neg_sampling=False)
    dp.prepare_data()
    assert "neg_item_id" in dp.train_data and dp.train_data["neg_item_id"].numel() > 0
    assert dp.train_data["user_id"].dtype == torch.int64
    assert dp.train_data["user_id"].numel() + dp.test_data["user_id"].numel() == N
    assert dp.config == {"num_users": N_USERS, "num_items": N_ITEMS, "num_metadata": {}}


def test_process_data_dynamic_neg_sampling():  # :74-80
    dp = ProcessData(dummy_df(), "user_id", "item_id", dynamic_neg_sampling=True)
    dp.prepare_data()
    assert "neg_item_id" not in dp.train_data


def test_process_data_does_not_mutate_callers_frame():
    df = dummy_df()
    cols = list(df.columns)
    ProcessData(df, "user_id", "item_id").prepare_data()
    assert list(df.columns) == cols


@pytest.mark.parametrize("dyn", [False, True])
def test_process_data_with_metadata_int_contract(dyn):  # the (B,M) contract, int and "[k]" columns
    dp = ProcessData(dummy_df(True), "user_id", "item_id", metadata_id_col=["category", "brand"],
                     dynamic_neg_sampling=dyn)
    dp.prepare_data()
    tr = dp.train_data
    assert tr["pos_metadata_id"].shape == (tr["user_id"].numel(), 2)
    assert ("neg_metadata_id" in tr) == (not dyn)
    assert dp.config["num_metadata"] == {"category": 5, "brand": 3}
    tab = dp.item_meta_table
    assert np.array_equal(tab[tr["pos_item_id"].numpy()], tr["pos_metadata_id"].numpy())
    if not dyn:
        assert np.array_equal(tab[tr["neg_item_id"].numpy()], tr["neg_metadata_id"].numpy())
    assert dp.item_to_metadata_map[3] == {"category": [int(tab[3, 0])], "brand": [int(tab[3, 1])]}


def test_multi_valued_metadata_is_rejected_loudly():
    df = dummy_df()
    df["tags"] = [[1, 2]] * len(df)
    with pytest.raises(NotImplementedError):
        ProcessData(df, "user_id", "item_id", metadata_id_col=["tags"])


def test_non_dense_ids_raise_index_error():
    df = dummy_df()
    df.loc[0, "item_id"] = 10_000
    with pytest.raises(IndexError):
        ProcessData(df, "user_id", "item_id").prepare_data()


def test_dataloader_dynamic_sampling_batch():  # :93-109
    dp = ProcessData(dummy_df(), "user_id", "item_id", dynamic_neg_sampling=True)
    dp.prepare_data()
    loader = FastDataLoader(dp.train_data, batch_size=32, shuffle=True, dynamic_neg_sampling=True, n_items=N_ITEMS)
    nb = 0
    for batch in loader:
        nb += 1
        assert "neg_item_id" in batch and batch["neg_item_id"].shape == batch["pos_item_id"].shape
        assert (batch["neg_item_id"] != batch["pos_item_id"]).all()
        assert batch["neg_item_id"].dtype == torch.int64
    assert nb == loader.num_batches == int(np.ceil(dp.train_data["user_id"].numel() / 32))


def test_dataloader_dynamic_sampling_with_metadata():  # :111-131 (fails in the reference: unhashable list)
    dp = ProcessData(dummy_df(True), "user_id", "item_id", metadata_id_col=["category", "brand"],
                     dynamic_neg_sampling=True)
    dp.prepare_data()
    for mapping in (dp.item_meta_table, dp.item_to_metadata_map):  # array form and the reference's dict form
        loader = FastDataLoader(dp.train_data, batch_size=64, shuffle=False, dynamic_neg_sampling=True,
                                n_items=N_ITEMS, item_to_metadata_map=mapping, metadata_id_cols=["category", "brand"])
        b = next(iter(loader))
        assert b["neg_metadata_id"].shape == b["pos_metadata_id"].shape == (64, 2)
        assert np.array_equal(dp.item_meta_table[b["neg_item_id"].numpy()], b["neg_metadata_id"].numpy())


def test_dataloader_argument_errors():  # dataset.py:347-351
    data = {"user_id": torch.arange(4), "pos_item_id": torch.arange(4)}
    with pytest.raises(ValueError):
        FastDataLoader(data, dynamic_neg_sampling=True)
    with pytest.raises(ValueError):
        FastDataLoader(data, dynamic_neg_sampling=True, n_items=4, metadata_id_cols=["a"])
    empty = FastDataLoader({"user_id": torch.empty(0, dtype=torch.long), "pos_item_id": torch.empty(0, dtype=torch.long)})
    assert list(empty) == [] and empty.num_batches == 0


def test_vectorised_sampler_equals_the_literal_loop():
    for n_items, B in ((2, 300), (3, 1000), (50, 4096)):
        pos = np.random.RandomState(B).randint(0, n_items, B)
        np.random.seed(4)
        a = sample_negatives_reference_stream(pos, n_items)
        s1 = np.random.randint(0, 1 << 30)
        np.random.seed(4)
        b = oloader.dynamic_negatives_loop(pos, n_items)
        s2 = np.random.randint(0, 1 << 30)
        assert np.array_equal(a, b) and s1 == s2 and (a != pos).all()


def test_tensor_ingest_matches_dataframe_ingest():
    df = dummy_df()
    np.random.seed(1)
    a = ProcessData(df, "user_id", "item_id", split_ratio=0.8)
    a.prepare_data()
    np.random.seed(1)
    b = TensorProcessData(torch.from_numpy(df["user_id"].values), torch.from_numpy(df["item_id"].values),
                          split_ratio=0.8)
    b.prepare_data()
    assert b.config == a.config
    for k in ("user_id", "pos_item_id", "neg_item_id"):
        assert torch.equal(a.train_data[k], b.train_data[k]) and torch.equal(a.test_data[k], b.test_data[k])


def test_tensor_ingest_remaps_arbitrary_ids():
    """remap_ids=True: arbitrary ids -> dense 0..n-1 by rank, original ids kept in user_index / item_index;
    without it a non-dense id raises at ingest like the reference's first embedding lookup would (dataset.py:30-31,
    268-269)."""
    df = dummy_df()
    u, i = torch.from_numpy(df["user_id"].values), torch.from_numpy(df["item_id"].values)
    with pytest.raises(IndexError, match="remap_ids"):
        TensorProcessData(u * 3 + 1, i, n_users=N_USERS, n_items=N_ITEMS)
    np.random.seed(1)
    a = TensorProcessData(u, i, split_ratio=0.8)
    a.prepare_data()
    np.random.seed(1)
    b = TensorProcessData(u * 3 + 1, i * 1000 + 7, split_ratio=0.8, remap_ids=True)
    b.prepare_data()
    assert b.config == a.config
    assert torch.equal(b.user_index, torch.arange(N_USERS) * 3 + 1)
    assert torch.equal(b.item_index, torch.arange(N_ITEMS) * 1000 + 7)
    for k in ("user_id", "pos_item_id", "neg_item_id"):
        assert torch.equal(a.train_data[k], b.train_data[k]) and torch.equal(a.test_data[k], b.test_data[k])
    # the original ids come back through the index
    assert torch.equal(b.user_index[b.train_data["user_id"]], a.train_data["user_id"] * 3 + 1)
    # ids missing from the stream leave no hole: 3 distinct users -> 3 rows
    c = TensorProcessData(torch.tensor([10, 500, 10, 7]), torch.tensor([2, 2, 9, 9]), split_ratio=1.0, remap_ids=True,
                          dynamic_neg_sampling=True)
    c.prepare_data()
    assert c.config["num_users"] == 3 and c.config["num_items"] == 2
    assert c.train_data["user_id"].tolist() == [1, 2, 1, 0] and c.train_data["pos_item_id"].tolist() == [0, 0, 1, 1]


def test_optimizer_dispatch():
    from torchrecsys_amd.engine import classify_optimizer
    ps = [torch.nn.Parameter(torch.zeros(4, 2)), torch.nn.Parameter(torch.zeros(3, 2))]
    assert classify_optimizer(torch.optim.SGD(ps, lr=0.1), ps) == "sgd"
    assert classify_optimizer(torch.optim.SGD(ps, lr=0.1, momentum=0.9), ps) == "generic"
    assert classify_optimizer(torch.optim.SGD(ps, lr=0.1, weight_decay=1e-4), ps) == "generic"
    assert classify_optimizer(torch.optim.SparseAdam(ps, lr=0.1), ps) == "sparse_adam"
    assert classify_optimizer(torch.optim.Adagrad(ps, lr=0.1), ps) == "adagrad"
    assert classify_optimizer(torch.optim.Adagrad(ps, lr=0.1, weight_decay=0.1), ps) == "generic"
    assert classify_optimizer(torch.optim.Adam(ps, lr=0.1), ps) == "sparse_adam"  # lazy rows (torch's Adam cannot)
    assert classify_optimizer(torch.optim.Adam(ps, lr=0.1, weight_decay=1e-5), ps) == "generic"
    assert classify_optimizer(torch.optim.RMSprop(ps, lr=0.1), ps) == "generic"
    assert classify_optimizer(torch.optim.SGD(ps[:1], lr=0.1), ps) == "generic"  # a parameter the optimiser lacks


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_compute_fails_loudly_without_a_gpu():
    from torchrecsys_amd.model import TorchRecSys
    with contextlib.redirect_stdout(io.StringIO()) as out:
        model = TorchRecSys(dummy_df(), "user_id", "item_id", n_factors=8, net_type="fm")
    assert "Factorization Machine" in out.getvalue()
    assert sorted(model.state_dict()) == ["net.item.weight", "net.linear_item.weight", "net.linear_user.weight",
                                          "net.user.weight"]
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    for call in (lambda: model.fit(opt, epochs=1), lambda: model.evaluate(), lambda: model.predict(0),
                 lambda: model.net.forward({"user_id": torch.tensor([0]), "pos_item_id": torch.tensor([0])},
                                           "user_id", "pos_item_id")):
        with pytest.raises(RuntimeError, match="MI355X"):
            call()


def test_unimplemented_net_types_raise():
    from torchrecsys_amd.model import TorchRecSys
    with pytest.raises(NotImplementedError):
        TorchRecSys(dummy_df(), "user_id", "item_id", net_type="neucf")
    with pytest.raises(AssertionError):
        TorchRecSys(dummy_df(), "user_id", "item_id", net_type="nope")


def test_seeded_construction_reproduces_reference_initial_weights():
    """Same RNG calls in the same order as the reference: bit-identical initial state_dict (golden G4 'init')."""
    from conftest import load_golden, sub
    from torchrecsys_amd.model import TorchRecSys
    for net_type in ("linear", "fm"):
        for dyn in (False, True):
            g = load_golden(f"g4_{net_type}_{'dyn' if dyn else 'static'}.npz")
            df = pd.DataFrame({"user": g["df_user"], "item": g["df_item"]})
            np.random.seed(7)
            torch.manual_seed(7)
            with contextlib.redirect_stdout(io.StringIO()):
                model = TorchRecSys(df, "user", "item", n_factors=16, net_type=net_type, dynamic_neg_sampling=dyn)
            sd = model.state_dict()
            ref = sub(g, "init")
            assert sorted(sd) == sorted(ref)
            for k, v in ref.items():
                assert np.array_equal(sd[k].cpu().numpy(), v), k


def test_host_thread_budget_caps_and_restores(monkeypatch):
    """helper.cuda.host_threads(): torch's intra-op pool is capped at min(CPU budget, 8) inside the block (a container's
    CPU quota is far below the host's core count torch sizes its pool from) and restored afterwards; TRS_HOST_THREADS
    overrides; entry points keep the reference's signatures through the wrapper."""
    import inspect
    import torch
    from torchrecsys_amd.helper.cuda import cpu_budget, host_threads
    from torchrecsys_amd.model import TorchRecSys
    assert 1 <= cpu_budget() <= (os.cpu_count() or 1)
    before = torch.get_num_threads()
    try:
        torch.set_num_threads(max(before, 2))
        monkeypatch.setenv("TRS_HOST_THREADS", "1")
        with host_threads():
            assert torch.get_num_threads() == 1
        assert torch.get_num_threads() == max(before, 2)
        monkeypatch.setenv("TRS_HOST_THREADS", "4096")  # never raises the count
        with host_threads():
            assert torch.get_num_threads() == max(before, 2)
    finally:
        torch.set_num_threads(before)
    assert list(inspect.signature(TorchRecSys.fit).parameters)[:4] == ["self", "optimizer", "epochs", "batch_size"]
    assert list(inspect.signature(TorchRecSys.__init__).parameters)[:4] == ["self", "dataset", "user_id_col", "item_id_col"]


def test_mlp_custom_hidden_layers_and_batch_norm_toggle():  # reference tests/test_model_and_features.py:145-185
    from torchrecsys_amd.collaborative.mlp import MLP
    from torchrecsys_amd.dataset.dataset import ProcessData
    rs = np.random.RandomState(0)
    df = pd.DataFrame({"user_id": np.concatenate([np.arange(100), rs.randint(0, 100, 900)]),
                       "item_id": np.concatenate([np.arange(50), rs.randint(0, 50, 950)])})
    counts = ProcessData(df, "user_id", "item_id")
    custom = [64, 32]
    mlp = MLP(n_users=counts.num_users, n_items=counts.num_items, n_metadata={}, n_factors=16, use_metadata=False,
              hidden_layers=custom)
    assert len(mlp.fcs) == len(custom)
    assert mlp.fcs[0].out_features == custom[0] and mlp.fcs[1].out_features == custom[1]
    assert mlp.fcs[0].in_features == 2 * 16 and mlp.output_layer.in_features == custom[-1]
    with_bn = MLP(n_users=counts.num_users, n_items=counts.num_items, n_metadata={}, n_factors=16, use_metadata=False,
                  use_batch_norm=True)
    assert hasattr(with_bn, "bns") and len(with_bn.bns) == len(with_bn.hidden_layers)
    without = MLP(n_users=counts.num_users, n_items=counts.num_items, n_metadata={}, n_factors=16, use_metadata=False,
                  use_batch_norm=False)
    assert not hasattr(without, "bns") or len(without.bns) == 0
