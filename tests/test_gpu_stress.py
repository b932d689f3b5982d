# -*- coding: utf-8 -*-
"""Randomised shapes for the presorted two-launch step (SGD / SparseAdam / Adagrad): odd batch sizes, tiny tables (every
row duplicated many times), batches smaller than one 64-reference chunk, D from 1 to 200, skewed ids — each compared
with the oracle on a few steps.  GPU only."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import nets as onets
from oracle import optim as ooptim
from oracle.nets import touched_rows
from test_gpu_kernels import DEV, make_case

pytestmark = pytest.mark.gpu


def _cases():
    rs = np.random.RandomState(2024)
    out = []
    for c in range(36):
        net = "fm" if rs.rand() < 0.6 else "linear"
        D = int(rs.choice([1, 2, 4, 7, 8, 12, 16, 24, 32, 33, 64, 96, 128, 200]))
        B = int(rs.choice([1, 3, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 255, 257, 500, 1000]))
        NU = int(rs.choice([1, 2, 5, 40, 300, 5000]))
        NI = int(rs.choice([2, 3, 9, 60, 700]))
        nb = int(rs.choice([1, 2, 5]))
        kind = ["sgd", "sparse_adam", "adagrad"][c % 3]
        if kind != "sgd":  # with a handful of users every row couples to a noise-level sign flip within a step or two
            NU = max(NU, 40)
        out.append((c, net, D, B, NU, NI, nb, kind, bool(rs.rand() < 0.4)))
    return out


@pytest.mark.parametrize("case", _cases(), ids=lambda c: f"{c[0]}-{c[1]}-D{c[2]}-B{c[3]}-U{c[4]}-I{c[5]}-n{c[6]}-{c[7]}")
def test_presorted_step_random_shapes(case):
    from torchrecsys_amd import _lib, ops
    seed_, net, D, B, NU, NI, nb, kind, skew = case
    rs = np.random.RandomState(seed_)
    p, _, _ = make_case(net, D, 0, 8, NU=NU, NI=NI, seed=seed_)
    u, i, j = rs.randint(0, NU, nb * B), rs.randint(0, NI, nb * B), rs.randint(0, NI, nb * B)
    if skew:
        i[rs.rand(nb * B) < 0.5] = NI - 1
    lin = ("user_bias.weight", "item_bias.weight") if net == "linear" else ("linear_user.weight", "linear_item.weight")
    names = ["user.weight", "item.weight", lin[0], lin[1]]
    t = {k: torch.from_numpy(v.copy()).to(DEV) for k, v in p.items()}
    T, keep = ops.make_tables(*(t[k] for k in names))
    err = torch.zeros(1, dtype=torch.int32, device=DEV)
    ps = ops.EpochPresort(nb, B, NU, NI, DEV)
    ps.run(None, None, 0, 0, 0, err, given_ids=[torch.from_numpy(a.astype(np.int32)).to(DEV) for a in (u, i, j)])
    ids, sk, sv, udup, usorted, idup = ps.step_args(0)
    gz, du = torch.empty((2, B), device=DEV), torch.empty((B, D), device=DEV)
    losses = torch.zeros(nb, device=DEV)
    lr, b1, b2, eps, lr_decay = {"sgd": (0.05, 0, 0, 0, 0), "sparse_adam": (0.01, 0.9, 0.999, 1e-8, 0.0),
                                 "adagrad": (0.05, 0, 0, 1e-10, 0.01)}[kind]
    s1 = {k: torch.zeros_like(t[k]) for k in names}
    s2 = {k: torch.zeros_like(t[k]) for k in names}
    o = None
    if kind != "sgd":
        gacc, gacc_lin = torch.zeros_like(t["item.weight"]), torch.zeros_like(t[lin[1]])
        cut_rows = torch.empty(2 * B // 64 + 64, dtype=torch.int32, device=DEV)
        cut_count = torch.zeros(2, dtype=torch.int32, device=DEV)
        o = _lib.TrsOpt()
        o.kind, o.lr, o.beta1, o.beta2, o.eps, o.lr_decay, o.step0 = (1 if kind == "sparse_adam" else 2), lr, b1, b2, eps, lr_decay, 0
        o.user_s1, o.item_s1, o.user_lin_s1, o.item_lin_s1 = (ops.ptr(s1[k]) for k in names)
        if kind == "sparse_adam":
            o.user_s2, o.item_s2, o.user_lin_s2, o.item_lin_s2 = (ops.ptr(s2[k]) for k in names)
        o.gacc, o.gacc_lin, o.cut_rows, o.cut_count = ops.ptr(gacc), ops.ptr(gacc_lin), ops.ptr(cut_rows), ops.ptr(cut_count)
        o.cut_capacity = cut_rows.numel()
    if kind == "sgd" and seed_ % 2 == 0:  # every other SGD case: the sparse regime's flag mode (no sorted runs)
        ef = ops.EpochFlags(nb, B, NU, NI, DEV)
        ef.run(None, None, 0, 0, 0, err, given_ids=[torch.from_numpy(a.astype(np.int32)).to(DEV) for a in (u, i, j)])
        fids, fu, fi = ef.step_args(0)
        ops.train_steps_sgd(net, T, None, None, 0, 0, 0, B, nb, lr, *fids, gz, du, losses, err,
                            ops.train_scratch(NU, NI, B, D, DEV), 1, None, user_dup=fu, item_dup=fi,
                            ustage=torch.empty((B, D), device=DEV))
    else:
        ops.train_steps_sgd(net, T, None, None, 0, 0, 0, B, nb, lr, *ids, gz, du, losses, err,
                            ops.train_scratch(NU, NI, B, D, DEV), 1, None, sk, sv, ps.key_bytes, udup,
                            torch.empty((B, D), device=DEV), usorted, o,
                            item_dup=idup if kind == "sgd" else None)  # plain SGD: K1 also takes item rows referenced once
    torch.cuda.synchronize()
    ref = {k: v.copy() for k, v in p.items()}
    r1 = {k: np.zeros_like(v) for k, v in p.items()}
    r2 = {k: np.zeros_like(v) for k, v in p.items()}
    for b in range(nb):
        batch = {"user_id": u[b * B:(b + 1) * B], "pos_item_id": i[b * B:(b + 1) * B], "neg_item_id": j[b * B:(b + 1) * B]}
        _, _, loss, grads = onets.train_forward_backward(net, ref, batch)
        if kind == "sgd":
            ooptim.sgd_step(ref, grads, lr)
        else:
            rows = touched_rows(net, ref, batch)
            for k in names:
                if kind == "sparse_adam":
                    ooptim.sparse_adam_rows(ref[k], grads[k], rows[k], r1[k], r2[k], b + 1, lr, b1, b2, eps)
                else:
                    ooptim.adagrad_rows(ref[k], grads[k], rows[k], r1[k], b + 1, lr, lr_decay, eps)
        # adaptive rules on tiny tables: a coalesced gradient that cancels exactly in one summation order and to 1e-10 in
        # another becomes a +-lr step (see tests/test_gpu_kernels.py), after which the trajectories part: the loss is
        # compared on the first step only, the weights by the bulk criterion below
        if kind == "sgd" or b == 0:
            assert abs(losses[b].item() / B - float(loss)) <= 3e-5 * max(abs(float(loss)), 1e-3), (b, losses[b].item() / B, loss)
    for k in names:
        got = t[k].cpu().numpy()
        if kind == "sgd":
            assert rel_err(got, ref[k]) < 3e-5, k
        else:  # bulk criterion of the adaptive rules (noise-level gradients become +-lr steps: tests/test_gpu_kernels.py)
            ok = (np.abs(got - ref[k]) <= 2e-3 * max(np.abs(ref[k]).max(), 1e-6)).mean()  # fraction of elements
            assert ok >= 0.8, (k, ok)
            assert np.isfinite(got).all()
    assert err.item() == 0
    if kind != "sgd":
        assert float(gacc.abs().max()) == 0.0


@pytest.mark.parametrize("n,batch,slice_batches,net_type", [(1000, 256, 2, "fm"), (1000, 999, 4, "linear"),
                                                            (777, 64, 3, "fm"), (300, 512, 8, "fm"),
                                                            (5000, 100, 7, "linear"), (2049, 1024, 1, "fm"),
                                                            (130, 1, 16, "fm")])
def test_fit_presorted_path_equals_ownership_path_random_sizes(n, batch, slice_batches, net_type, monkeypatch):
    """fit() over odd stream lengths / batch sizes / slice lengths (partial last batch, batch larger than the stream,
    batch of one, one-batch slices): the presorted two-launch path and the four-launch ownership path train the same
    model on the same reference-RNG batches."""
    import contextlib
    import io
    import pandas as pd
    from torchrecsys_amd.engine import SparseScorerTrainer
    from torchrecsys_amd.model import TorchRecSys
    rs = np.random.RandomState(n + batch)
    n_u, n_i = 50, 23
    df = pd.DataFrame({"user": np.concatenate([np.arange(n_u), rs.randint(0, n_u, n - n_u)]),
                       "item": np.concatenate([np.arange(n_i), rs.randint(0, n_i, n - n_i)])})
    out = {}
    monkeypatch.setattr(SparseScorerTrainer, "SLICE_BATCHES", slice_batches)
    for path in ("presorted", "ownership"):
        if path == "ownership":
            monkeypatch.setattr(SparseScorerTrainer, "wants_presort", lambda self, b: False)
        torch.manual_seed(3)
        np.random.seed(3)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            model = TorchRecSys(dataset=df, user_id_col="user", item_id_col="item", n_factors=8, net_type=net_type,
                                dynamic_neg_sampling=True)
            model.fit(optimizer=torch.optim.SGD(model.parameters(), lr=0.1), epochs=3, batch_size=batch)
        out[path] = ({k: v.cpu().numpy() for k, v in model.state_dict().items()}, buf.getvalue())
    assert out["presorted"][1] == out["ownership"][1]  # printed epoch losses (4 decimals)
    for k, v in out["ownership"][0].items():
        assert rel_err(out["presorted"][0][k], v) < 1e-5, k


@pytest.mark.parametrize("net_type,n_meta_cols", [("fm", 1), ("linear", 2), ("fm", 3)])
def test_fit_with_metadata_presorted_path_equals_generic_path(net_type, n_meta_cols, monkeypatch):
    """fit() of a metadata scorer (DataFrame front-end, reference RNG): the presorted step with the scorer's staging mode
    trains the same model as the generic staged path (TRS_META_FAST=0) on the same batches."""
    import contextlib
    import io
    import pandas as pd
    from torchrecsys_amd.model import TorchRecSys
    rs = np.random.RandomState(11)
    n, n_u, n_i = 3000, 80, 37
    items = np.concatenate([np.arange(n_i), rs.randint(0, n_i, n - n_i)])
    df = pd.DataFrame({"user": np.concatenate([np.arange(n_u), rs.randint(0, n_u, n - n_u)]), "item": items})
    cols = []
    for c in range(n_meta_cols):
        cat_of_item = rs.randint(0, 5 + c, n_i)
        cat_of_item[:5 + c] = np.arange(5 + c)
        df[f"cat{c}"] = cat_of_item[items]
        cols.append(f"cat{c}")
    out = {}
    for path in ("presorted", "generic"):
        monkeypatch.setenv("TRS_META_FAST", "1" if path == "presorted" else "0")
        torch.manual_seed(3)
        np.random.seed(3)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            model = TorchRecSys(dataset=df, user_id_col="user", item_id_col="item", metadata_id_col=cols, n_factors=16,
                                net_type=net_type, dynamic_neg_sampling=True)
            model.fit(optimizer=torch.optim.SGD(model.parameters(), lr=0.1), epochs=3, batch_size=128)
        out[path] = ({k: v.cpu().numpy() for k, v in model.state_dict().items()}, buf.getvalue())
    assert out["presorted"][1] == out["generic"][1]
    for k, v in out["generic"][0].items():
        assert rel_err(out["presorted"][0][k], v) < 1e-5, k


@pytest.mark.parametrize("net_type,n_meta_cols", [("fm", 1), ("linear", 2), ("fm", 3)])
@pytest.mark.parametrize("oname", ["adam", "adagrad"])
def test_fit_with_metadata_and_adaptive_rule_equals_generic_path(net_type, n_meta_cols, oname, monkeypatch):
    """fit() of a metadata scorer with Adam (lazy / SparseAdam semantics on the tables) or Adagrad: the presorted step
    with the rule fused in trains the same model as the generic staged path (TRS_META_FAST=0): printed losses, weights,
    optimiser state under torch's key names."""
    import contextlib
    import io
    import re
    import pandas as pd
    from torchrecsys_amd.model import TorchRecSys
    rs = np.random.RandomState(12)
    n, n_u, n_i = 4000, 120, 37
    items = np.concatenate([np.arange(n_i), rs.randint(0, n_i, n - n_i)])
    df = pd.DataFrame({"user": np.concatenate([np.arange(n_u), rs.randint(0, n_u, n - n_u)]), "item": items})
    cols = []
    for c in range(n_meta_cols):
        cat_of_item = rs.randint(0, 5 + c, n_i)
        cat_of_item[:5 + c] = np.arange(5 + c)
        df[f"cat{c}"] = cat_of_item[items]
        cols.append(f"cat{c}")
    out = {}
    from torchrecsys_amd.engine import SparseScorerTrainer
    calls, inner = [0], SparseScorerTrainer.fast_sorted_steps

    def counted(self, *a, **k):
        calls[0] += 1
        return inner(self, *a, **k)

    monkeypatch.setattr(SparseScorerTrainer, "fast_sorted_steps", counted)
    for path in ("presorted", "generic"):
        monkeypatch.setenv("TRS_META_FAST", "1" if path == "presorted" else "0")
        torch.manual_seed(3)
        np.random.seed(3)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            model = TorchRecSys(dataset=df, user_id_col="user", item_id_col="item", metadata_id_col=cols, n_factors=32,
                                net_type=net_type, dynamic_neg_sampling=True)
            opt = (torch.optim.Adam(model.parameters(), lr=0.01) if oname == "adam"
                   else torch.optim.Adagrad(model.parameters(), lr=0.05))
            model.fit(optimizer=opt, epochs=2, batch_size=128)
        assert (calls[0] > 0) == (path == "presorted")  # the C step loop ran / did not run
        calls[0] = 0
        st = opt.state[model.net.metadata[0].weight]
        losses = [float(x) for x in re.findall(r"Training Loss: ([0-9.]+)", buf.getvalue())]
        out[path] = ({k: v.cpu().numpy() for k, v in model.state_dict().items()}, losses,
                     (st["exp_avg_sq"] if oname == "adam" else st["sum"]).cpu().numpy(), int(st["step"]))
    assert out["presorted"][1] == pytest.approx(out["generic"][1], abs=2.01e-4) and len(out["generic"][1]) == 2
    assert out["presorted"][3] == out["generic"][3] == 2 * -(-int(n * 0.8) // 128)
    for k, v in out["generic"][0].items():
        d = np.abs(out["presorted"][0][k] - v).max(axis=1)
        assert (d <= 1e-3 * np.abs(v).max()).mean() >= 0.95, k  # bulk (see test_presorted_adaptive_rules_match_the_oracle)
    assert rel_err(out["presorted"][2], out["generic"][2]) < 1e-2
