# -*- coding: utf-8 -*-
"""API-level parity on the GPU: torchrecsys_amd.TorchRecSys / nets / engine against golden vectors recorded from the
real reference (tests/golden/make_golden.py).  GPU only."""
import contextlib
import io
import re

import numpy as np
import pandas as pd
import pytest
import torch

from conftest import load_golden, rel_err, sub
from oracle import nets as onets

pytestmark = pytest.mark.gpu
TOL = 1e-5
DEV = "cuda:0"
META_SIZES = [5, 7, 4]


def seed(s):
    np.random.seed(s)
    torch.manual_seed(s)


def build_net(name, M, g, prefix="init"):
    """Our module with the reference's initial weights loaded through state_dict (same parameter names)."""
    from torchrecsys_amd.collaborative.fm import FM
    from torchrecsys_amd.collaborative.linear import Linear
    net_type = name.split("_")[0]
    n_meta = {f"m{m}": META_SIZES[m] for m in range(M)}
    kw = {}
    if net_type == "mlp":
        from torchrecsys_amd.collaborative.mlp import MLP
        cls = MLP
        kw = {"hidden_layers": [32, 16], "use_batch_norm": name == "mlp"}
    else:
        cls = {"linear": Linear, "fm": FM}[net_type]
    net = cls(n_users=40, n_items=30, n_metadata=n_meta, n_factors=8, use_metadata=M > 0, **kw)
    sd = {k: torch.from_numpy(np.array(v)) for k, v in sub(g, prefix).items()}
    net.load_state_dict(sd)
    return net.to(DEV)


def golden_batch(g):
    b = {k: torch.from_numpy(v).long() for k, v in sub(g, "batch").items()}
    for k in ("pos_metadata_id", "neg_metadata_id"):
        if k in b and b[k].dim() == 1:
            b[k] = b[k].reshape(-1, 1)
    return b


def dev_ids(net, b):
    ids = {"user": b["user_id"], "pos": b["pos_item_id"], "neg": b["neg_item_id"]}
    if "pos_metadata_id" in b:
        ids["pos_meta"], ids["neg_meta"] = b["pos_metadata_id"], b["neg_metadata_id"]
    return {k: v.to(DEV).contiguous() for k, v in ids.items()}


SPARSE_NETS = ["linear", "fm"]
ALL_NETS = ["linear", "fm", "mlp", "mlp_nobn"]


@pytest.mark.parametrize("name", ALL_NETS)
@pytest.mark.parametrize("M", [0, 1, 3])
def test_g1_autograd_bridge(name, M):
    """net.forward x2 + hinge_loss + loss.backward() through the HIP autograd Functions == reference autograd."""
    from torchrecsys_amd.helper.loss import hinge_loss
    from torchrecsys_amd.evaluate.metrics import Metrics
    g = load_golden(f"g1_{name}_M{M}.npz")
    net, b = build_net(name, M, g), golden_batch(g)
    mk = ("pos_metadata_id", "neg_metadata_id") if M else (None, None)
    pos = net.forward(b, "user_id", "pos_item_id", mk[0])
    neg = net.forward(b, "user_id", "neg_item_id", mk[1])
    assert pos.shape == g["pos"].shape and neg.shape == g["neg"].shape
    loss = hinge_loss(pos, neg)
    loss.backward()
    assert rel_err(pos.detach().cpu().numpy(), g["pos"]) < TOL
    assert rel_err(neg.detach().cpu().numpy(), g["neg"]) < TOL
    assert abs(loss.item() - float(g["loss"])) <= TOL * abs(float(g["loss"]))
    assert float(Metrics().auc_score(pos, neg).item()) == pytest.approx(float(g["auc"]), abs=1e-6)
    ref = sub(g, "grad")
    is_mlp = name.startswith("mlp")
    for k, p in net.named_parameters():
        assert p.grad is not None, k
        gd = p.grad.to_dense() if p.grad.is_sparse else p.grad
        if name == "mlp" and k.startswith("fcs") and k.endswith("bias"):
            assert float(gd.abs().max()) < 1e-6  # bias in front of train-mode BN: mathematically zero gradient
            continue
        assert rel_err(gd.cpu().numpy(), ref[k]) < (2 * TOL if is_mlp else TOL), k
    if name == "mlp":  # BatchNorm running statistics after the two training passes (positive first)
        after = sub(g, "after_fwd")
        sd = net.state_dict()
        for k, v in after.items():
            if "running" in k:
                assert rel_err(sd[k].cpu().numpy(), v) < TOL, k
            if "num_batches_tracked" in k:
                assert int(sd[k]) == int(v) == 2
        net.eval()  # G6: eval-mode scores use the running statistics
        pe = net.forward(b, "user_id", "pos_item_id", mk[0])
        assert rel_err(pe.detach().cpu().numpy(), g["pos_eval"]) < TOL
        net.train()
    if not is_mlp:  # fused pair forward gives the same scores (both passes run the same code: bitwise)
        p2, n2 = net.forward_pair(b)
        assert torch.equal(p2, pos.detach()) and torch.equal(n2, neg.detach())


@pytest.mark.parametrize("name", SPARSE_NETS)
@pytest.mark.parametrize("M", [0, 1, 3])
@pytest.mark.parametrize("oname", ["sgd", "sgdm", "adagrad", "sparseadam"])
def test_g2_engine_optimizer_trajectories(name, M, oname):
    """Three fused training steps honouring a user-built torch optimiser == the reference's trajectories."""
    from torchrecsys_amd.engine import SparseScorerTrainer
    g = load_golden(f"g2_{name}_M{M}_{oname}.npz")
    net, b = build_net(name, M, g), golden_batch(g)
    ps = list(net.parameters())
    opt = {"sgd": lambda: torch.optim.SGD(ps, lr=0.05), "sgdm": lambda: torch.optim.SGD(ps, lr=0.05, momentum=0.9),
           "adagrad": lambda: torch.optim.Adagrad(ps, lr=0.05),
           "sparseadam": lambda: torch.optim.SparseAdam(ps, lr=0.01)}[oname]()
    tr = SparseScorerTrainer(net, opt, 64)
    assert tr.kind == {"sgd": "sgd", "sgdm": "generic", "adagrad": "adagrad", "sparseadam": "sparse_adam"}[oname]
    ids = dev_ids(net, b)
    losses = torch.zeros(3, dtype=torch.float32, device=DEV)
    for t in range(3):
        tr.step(ids, losses[t:t + 1])
        ref = sub(g, f"step{t}")
        sd = net.state_dict()
        for k, v in ref.items():
            assert rel_err(sd[k].cpu().numpy(), v) < 5 * TOL, (k, t)
    tr.check_errors()
    assert np.allclose(losses.cpu().numpy() / 64, g["losses"], rtol=2e-5, atol=1e-7)
    if oname == "sparseadam":  # optimizer.state stays truthful
        st = opt.state[net.user.weight]
        assert st["step"] == 3 and st["exp_avg"].shape == net.user.weight.shape


@pytest.mark.parametrize("name", ["mlp", "mlp_nobn"])
@pytest.mark.parametrize("M", [0, 1, 3])
@pytest.mark.parametrize("oname", ["sgd", "sgdm", "adagrad", "adam"])
def test_g2_mlp_trainer_trajectories(name, M, oname):
    """Fused MLP training steps (stacked passes, MFMA GEMMs, per-pass BN) vs the reference's trajectories."""
    from torchrecsys_amd.mlp_engine import MLPTrainer
    g = load_golden(f"g2_{name}_M{M}_{oname}.npz")
    net, b = build_net(name, M, g), golden_batch(g)
    net.train()
    ps = list(net.parameters())
    opt = {"sgd": lambda: torch.optim.SGD(ps, lr=0.05), "sgdm": lambda: torch.optim.SGD(ps, lr=0.05, momentum=0.9),
           "adagrad": lambda: torch.optim.Adagrad(ps, lr=0.05), "adam": lambda: torch.optim.Adam(ps, lr=0.01)}[oname]()
    tr = MLPTrainer(net, opt, 64)
    assert tr.kind == {"sgd": "sgd", "sgdm": "generic", "adagrad": "adagrad", "adam": "sparse_adam"}[oname]
    ids = dev_ids(net, b)
    losses = torch.zeros(3, dtype=torch.float32, device=DEV)
    adaptive = oname in ("adagrad", "adam")
    for t in range(3):
        tr.step(ids, losses[t:t + 1])
        sd = net.state_dict()
        for k, v in sub(g, f"step{t}").items():
            if v.dtype != np.float32:
                continue
            got = sd[k].cpu().numpy()
            if adaptive:
                # adaptive optimisers turn rounding-noise gradients into +-lr steps (in the reference too): compare the
                # bulk, skip the pure-noise parameters (tests/test_oracle_golden.py explains)
                if k.endswith("bias") or k.endswith("running_mean"):
                    continue
                d = np.abs(got.astype(np.float64) - v) / max(np.abs(v).max(), 1e-12)
                assert np.quantile(d, 0.95) < (1e-4 if t == 0 else 1e-2), (k, t)
            else:
                assert rel_err(got, v) < 5 * TOL, (k, t)
    tr.check_errors()
    ltol = 1e-3 if adaptive else 2e-5
    assert np.allclose(losses.cpu().numpy() / 64, g["losses"], rtol=ltol, atol=1e-6)


@pytest.mark.parametrize("name", SPARSE_NETS)
def test_adam_on_sparse_tables_gets_lazy_adam_semantics(name):
    """torch.optim.Adam (the README quick-start optimiser, which the reference cannot run on sparse gradients) follows
    the SparseAdam trajectory of the golden run, with its moments in optimizer.state under Adam's key names."""
    from torchrecsys_amd.engine import SparseScorerTrainer
    g = load_golden(f"g2_{name}_M1_sparseadam.npz")
    net, b = build_net(name, 1, g), golden_batch(g)
    opt = torch.optim.Adam(list(net.parameters()), lr=0.01)
    tr = SparseScorerTrainer(net, opt, 64)
    assert tr.kind == "sparse_adam"
    ids = dev_ids(net, b)
    losses = torch.zeros(3, dtype=torch.float32, device=DEV)
    for t in range(3):
        tr.step(ids, losses[t:t + 1])
        for k, v in sub(g, f"step{t}").items():
            assert rel_err(net.state_dict()[k].cpu().numpy(), v) < 5 * TOL, (k, t)
    st = opt.state[net.user.weight]
    assert float(st["step"]) == 3.0 and set(st) == {"step", "exp_avg", "exp_avg_sq"}


@pytest.mark.parametrize("name", SPARSE_NETS)
def test_generic_backward_path_with_any_optimizer(name):
    """TorchRecSys.forward + hinge_loss + TorchRecSys.backward (reference model.py:171-200) with SparseAdam."""
    from torchrecsys_amd.helper.loss import hinge_loss
    from torchrecsys_amd.model import TorchRecSys
    g = load_golden(f"g2_{name}_M1_sparseadam.npz")
    net, b = build_net(name, 1, g), golden_batch(g)
    opt = torch.optim.SparseAdam(list(net.parameters()), lr=0.01)
    for t in range(3):
        pos, neg = TorchRecSys.forward(None, net, b)
        loss = hinge_loss(pos, neg)
        val = TorchRecSys.backward(None, loss, opt)
        assert abs(val - float(g["losses"][t])) <= 2e-5 * max(abs(float(g["losses"][t])), 1e-3)
        for k, v in sub(g, f"step{t}").items():
            assert rel_err(net.state_dict()[k].cpu().numpy(), v) < 5 * TOL, (k, t)


def run_model(net_type, dyn, g, rng="reference", **kw):
    from torchrecsys_amd.model import TorchRecSys
    df = pd.DataFrame({"user": g["df_user"], "item": g["df_item"]})
    if "item_cat" in g:  # the metadata front-end shape the reference can run: one column of "[k]" strings
        df["cat"] = [f"[{g['item_cat'][i]}]" for i in df["item"].values]
        kw = dict(kw, metadata_id_col=["cat"])
    seed(7)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        model = TorchRecSys(dataset=df, user_id_col="user", item_id_col="item", n_factors=16, net_type=net_type,
                            dynamic_neg_sampling=dyn, rng=rng, **kw)
        init = {k: v.cpu().numpy().copy() for k, v in model.state_dict().items()}
        opt = torch.optim.SGD(model.parameters(), lr=0.05)
        model.fit(optimizer=opt, epochs=2, batch_size=256)
        final = {k: v.cpu().numpy().copy() for k, v in model.state_dict().items()}
        if not dyn:
            model.evaluate(batch_size=256)
        top = model.predict(user_id=3, top_k=10, prediction_batch_size=37)
    return model, init, final, top, buf.getvalue()


@pytest.mark.parametrize("net_type", SPARSE_NETS + ["mlp"])
@pytest.mark.parametrize("dyn", [False, True])
@pytest.mark.parametrize("fixture", ["g4", "g4m"])
def test_g4_end_to_end_fit_evaluate_predict(net_type, dyn, fixture):
    """fit() / evaluate() / predict() through the public API against the REFERENCE's own run of the same script
    (tests/golden/make_golden.py).  g4m: with one metadata column of string-encoded one-element lists — the only
    metadata front-end shape the reference can run end to end; its predict() raises with metadata
    (model.py:401), so the top-10 fixture is the reference's net-level eval-mode scores sorted the reference's way."""
    g = load_golden(f"{fixture}_{net_type}_{'dyn' if dyn else 'static'}.npz")
    model, init, final, top, txt = run_model(net_type, dyn, g)
    for k, v in sub(g, "init").items():  # seeded construction: bit-identical initial weights
        assert np.array_equal(init[k], v), k
    losses = [float(x) for x in re.findall(r"Training Loss: ([0-9.]+)", txt)]
    is_mlp = net_type == "mlp"
    # printed with 4 decimals.  The MLP trajectory (BatchNorm, 156 steps) amplifies summation-order differences — the
    # reference itself differs between 1 and 8 BLAS threads, and the numpy oracle deviates from the
    # golden run by the same 1% (tests/test_oracle_golden.py::test_g4_oracle_end_to_end).  The float atomics of the
    # embedding update make the trajectory differ from run to run as well: six runs of the g4m dynamic case (two update
    # kernels x three repeats) printed second-epoch losses 0.9890 .. 0.9896 around the golden 0.9892 — so the MLP is held
    # to 6e-4 (losses) / 0.2 (weights, max-norm; oracle-vs-golden is 0.1).
    assert losses == pytest.approx(list(g["epoch_losses"]), abs=6e-4 if is_mlp else 1.01e-4)
    for k, v in sub(g, "final").items():
        if v.dtype != np.float32:
            assert int(final[k]) == int(v), k  # BatchNorm num_batches_tracked: two per step
            continue
        # 78 SGD steps of accumulated fp32 rounding; the MLP's GEMMs sum in a different order than MKL's
        if is_mlp and not k.endswith(("user.weight", "item.weight")):
            continue  # near-zero BN biases etc. have no meaningful relative scale on a chaotic trajectory
        assert rel_err(final[k], v) < (0.2 if is_mlp else 2e-5), k
    assert isinstance(top, torch.Tensor) and top.dtype == torch.int64 and top.device.type == "cpu"
    if not is_mlp:
        assert np.array_equal(top.numpy(), g["top10_user3"])  # bit-exact top-k (tie-free fixture)
    if not dyn:
        tol = 5e-3 if is_mlp else 1.01e-4  # (MLP: a chaotic 156-step trajectory, see above; 2.6e-3 observed with metadata)
        assert float(re.findall(r"Testing loss: ([0-9.]+)", txt)[0]) == pytest.approx(float(g["eval_loss"]), abs=tol)
        # the MLP's test AUC sits at 0.50 on 2000 pairs after a chaotic 156-step run whose embedding updates use float
        # atomics (order varies with the allocator's addresses): observed 0.497-0.509 against the golden 0.4975
        assert float(re.findall(r"Testing auc: ([0-9.]+)", txt)[0]) == pytest.approx(float(g["eval_auc"]),
                                                                                    abs=0.025 if is_mlp else 5 * tol)
    # predict()/evaluate() parity on IDENTICAL weights: load the reference's final state_dict, then the eval-mode scores
    # of user 3 match to 1e-5 and the top-10 is bit-exact (all three nets)
    model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sub(g, "final").items()})
    model.net.eval()
    sc = model.net.score_all_items(3, model._item_meta_dev())
    assert rel_err(sc.cpu().numpy(), g["scores_user3"]) < TOL
    with contextlib.redirect_stdout(io.StringIO()):
        assert np.array_equal(model.predict(user_id=3, top_k=10).numpy(), g["top10_user3"])
    # same banner / line formats as the reference
    ref_lines = [l for l in str(g["stdout"]).splitlines() if l.strip()]
    got_lines = [l for l in txt.splitlines() if l.strip()]
    assert [re.sub(r"[0-9.]+$", "", l) for l in got_lines] == [re.sub(r"[0-9.]+$", "", l) for l in ref_lines]


@pytest.mark.parametrize("net_type", SPARSE_NETS)
@pytest.mark.parametrize("rng", ["reference", "device"])
def test_presort_slices_prefetched_on_the_side_stream(net_type, rng, monkeypatch):
    """Several presort slices per epoch (double-buffered, the next one sorted on a side stream while the current one's
    steps run, short tail slice): same weights as with one slice per epoch."""
    from torchrecsys_amd.engine import SparseScorerTrainer
    g = load_golden(f"g4_{net_type}_dyn.npz")
    kw = dict(rng="device", seed=5) if rng == "device" else {}
    monkeypatch.setattr(SparseScorerTrainer, "SLICE_BATCHES", 4)
    model, _, sliced, _, txt_s = run_model(net_type, True, g, **kw)
    monkeypatch.setattr(SparseScorerTrainer, "SLICE_BATCHES", 256)
    _, _, whole, _, txt_w = run_model(net_type, True, g, **kw)
    ls, lw = ([float(x) for x in re.findall(r"Training Loss: ([0-9.]+)", t)] for t in (txt_s, txt_w))
    assert len(ls) == 2 and ls == pytest.approx(lw, abs=1.01e-4)
    for k, v in whole.items():
        assert rel_err(sliced[k], v) < 1e-6, k  # cut hot-row runs use atomics: order-dependent rounding only
    if rng == "reference":
        for k, v in sub(g, "final").items():
            assert rel_err(sliced[k], v) < 2e-5, k


@pytest.mark.parametrize("net_type", SPARSE_NETS)
@pytest.mark.parametrize("oname", ["sparseadam", "adagrad"])
def test_adaptive_optimisers_fused_presorted_path_equals_generic_path(net_type, oname, monkeypatch):
    """fit() with SparseAdam / Adagrad runs the presorted two-launch step with the rule fused in (state in
    optimizer.state[p]); forcing the generic staged path (accumulate + elected-owner apply) on the same batches must
    give the same training: printed losses, weights and optimiser state."""
    from torchrecsys_amd.engine import SparseScorerTrainer
    from torchrecsys_amd.model import TorchRecSys
    g = load_golden(f"g4_{net_type}_dyn.npz")
    df = pd.DataFrame({"user": g["df_user"], "item": g["df_item"]})
    out = {}
    for path in ("fused", "generic"):
        if path == "generic":
            monkeypatch.setattr(SparseScorerTrainer, "wants_presort", lambda self, batch: False)
        seed(7)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            model = TorchRecSys(dataset=df, user_id_col="user", item_id_col="item", n_factors=16, net_type=net_type,
                                dynamic_neg_sampling=True)
            opt = (torch.optim.SparseAdam(list(model.parameters()), lr=0.01) if oname == "sparseadam"
                   else torch.optim.Adagrad(model.parameters(), lr=0.05))
            model.fit(optimizer=opt, epochs=2, batch_size=256)
        losses = [float(x) for x in re.findall(r"Training Loss: ([0-9.]+)", buf.getvalue())]
        st = opt.state[model.net.item.weight]
        out[path] = (losses, {k: v.cpu().numpy() for k, v in model.state_dict().items()},
                     (st["exp_avg_sq"] if oname == "sparseadam" else st["sum"]).cpu().numpy(), int(st["step"]))
    assert out["fused"][0] == pytest.approx(out["generic"][0], abs=2.01e-4) and len(out["fused"][0]) == 2
    n_batches = -(-int(len(df) * 0.8) // 256)
    assert out["fused"][3] == out["generic"][3] == 2 * n_batches  # every step counted once (C loop + partial last batch)
    for k, v in out["generic"][1].items():
        d = np.abs(out["fused"][1][k] - v).max(axis=1)
        assert (d <= 1e-3 * np.abs(v).max()).mean() >= 0.97, k  # bulk: see test_presorted_adaptive_rules_match_the_oracle
    assert rel_err(out["fused"][2], out["generic"][2]) < 1e-2


@pytest.mark.parametrize("net_type", SPARSE_NETS)
def test_device_rng_mode_trains(net_type):
    """rng='device': on-GPU shuffle + sampler; not the reference's stream, but it must learn the same problem."""
    g = load_golden(f"g4_{net_type}_dyn.npz")
    model, init, final, top, txt = run_model(net_type, True, g, rng="device", seed=3)
    losses = [float(x) for x in re.findall(r"Training Loss: ([0-9.]+)", txt)]
    assert len(losses) == 2 and abs(losses[0] - g["epoch_losses"][0]) < 0.02 and losses[1] <= losses[0] + 1e-3
    assert top.shape == (10,) and len(set(top.tolist())) == 10 and max(top.tolist()) < 100


def test_reference_api_invariants():
    """The structural invariants the reference's own tests assert (tests/test_model_and_features.py)."""
    from torchrecsys_amd.dataset.dataset import FastDataLoader, ProcessData
    from torchrecsys_amd.model import TorchRecSys
    rs = np.random.RandomState(0)
    n_u, n_i, n = 100, 50, 1000
    df = pd.DataFrame({"user_id": np.concatenate([np.arange(n_u), rs.randint(0, n_u, n - n_u)]),
                       "item_id": np.concatenate([np.arange(n_i), rs.randint(0, n_i, n - n_i)])})
    with contextlib.redirect_stdout(io.StringIO()):
        model = TorchRecSys(df, "user_id", "item_id", n_factors=16, net_type="linear")
        # test_batched_prediction / _consistency (:189-215)
        a = model.predict(user_id=0, top_k=5, prediction_batch_size=7)
        b = model.predict(user_id=0, top_k=5, prediction_batch_size=n_i + 1)
    assert a.shape == (5,) and (a < n_i).all() and torch.equal(a, b)
    with pytest.raises(IndexError):
        model.predict(user_id=n_u + 5)
    # ids outside the tables raise IndexError like aten::embedding does
    bad = {"user_id": torch.tensor([0, n_u + 3]), "pos_item_id": torch.tensor([1, 2])}
    with pytest.raises(IndexError):
        model.net.forward(bad, "user_id", "pos_item_id")


def test_mlp_bf16_amp_tracks_fp32():
    """use_amp=True: bf16 GEMM inputs / fp32 accumulate (the reference's fp16 autocast is CUDA-only).  Scores and
    gradients follow the fp32 path at bf16 tolerance; training reduces the loss."""
    from torchrecsys_amd.helper.loss import hinge_loss
    g = load_golden("g1_mlp_M1.npz")
    b = golden_batch(g)
    out = {}
    for amp in (False, True):
        net = build_net("mlp", 1, g)
        net.use_bf16 = amp
        net.train()
        pos, neg = net.forward_pair(b)
        loss = hinge_loss(pos, neg)
        loss.backward()
        out[amp] = (pos.detach().cpu().numpy(), loss.item(),
                    {k: (p.grad.to_dense() if p.grad.is_sparse else p.grad).cpu().numpy() for k, p in net.named_parameters()})
    assert rel_err(out[True][0], out[False][0]) < 3e-2
    assert abs(out[True][1] - out[False][1]) < 2e-2
    for k in ("user.weight", "fcs.0.weight", "output_layer.weight"):  # same direction (a few hinge flags may flip)
        a, c = out[True][2][k].reshape(-1).astype(np.float64), out[False][2][k].reshape(-1).astype(np.float64)
        assert a @ c / (np.linalg.norm(a) * np.linalg.norm(c)) > 0.97, k
    assert not np.array_equal(out[True][0], out[False][0])  # the bf16 path really ran


@pytest.mark.parametrize("use_bn", [True, False])
def test_mlp_bf16_resident_path_tracks_fp32(use_bn):
    """use_amp on tile-aligned shapes: layer inputs and BN-backward outputs live in HBM as bf16 and the nine GEMMs of a
    step run on the bf16-resident kernels (forward / dgrad through weight images, wgrad through the transposing LDS
    read).  A training step matches the fp32 step at bf16 tolerance, and the two bf16 implementations (resident vs
    fp32-staged, same rounding points except the stored activations) agree closely."""
    from torchrecsys_amd.model import TorchRecSys
    rs = np.random.RandomState(0)
    n_u, n_i, n, B = 500, 300, 4096, 256
    users = torch.from_numpy(np.concatenate([np.arange(n_u), rs.randint(0, n_u, n - n_u)]).astype(np.int64))
    items = torch.from_numpy(np.concatenate([np.arange(n_i), rs.randint(0, n_i, n - n_i)]).astype(np.int64))
    res = {}
    for mode in ("fp32", "amp"):
        seed(11)
        with contextlib.redirect_stdout(io.StringIO()):
            model = TorchRecSys.from_tensors(users, items, n_users=n_u, n_items=n_i, n_factors=64, net_type="mlp",
                                             hidden_layers=[256, 128], use_batch_norm=use_bn, use_amp=(mode == "amp"),
                                             dynamic_neg_sampling=True, rng="reference")
        net = model.net
        net.train()
        dev = net.user.weight.device
        ids = {"user": torch.arange(B, device=dev) % n_u, "pos": (torch.arange(B, device=dev) * 7) % n_i,
               "neg": (torch.arange(B, device=dev) * 13 + 5) % n_i}
        assert net.compute._resident(2 * B, True) == (mode == "amp")
        scores, ctx = net.compute.forward(ids, 2, True)
        g = torch.cat([torch.full((B,), -1.0 / B, device=dev), torch.full((B,), 1.0 / B, device=dev)])
        grads, dx0 = net.compute.backward(ctx, g)
        res[mode] = (scores.cpu().numpy(), dx0.cpu().numpy(),
                     {n_: grads[p].cpu().numpy() for n_, p in net.named_parameters() if p in grads})
    assert rel_err(res["amp"][0], res["fp32"][0]) < 3e-2
    for a, c in [(res["amp"][1], res["fp32"][1])] + [(res["amp"][2][k], res["fp32"][2][k])
                                                       for k in ("fcs.0.weight", "fcs.1.weight", "output_layer.weight")]:
        a, c = a.reshape(-1).astype(np.float64), c.reshape(-1).astype(np.float64)
        cos = a @ c / (np.linalg.norm(a) * np.linalg.norm(c))
        assert cos > 0.97, cos  # y_l, dx_l, x_l and dy_l all live in bf16 between the GEMMs


@pytest.mark.parametrize("use_bn,M,shape", [(True, 0, (64, [256, 128], 256)), (False, 0, (64, [256, 128], 256)),
                                             (True, 2, (32, [128, 128, 128], 384)), (True, 1, (128, [512, 256], 1024))])
def test_mlp_bf16_resident_step_matches_the_bf16_oracle(use_bn, M, shape):
    """The bf16-RESIDENT training step (use_amp on tile-aligned nets) against oracle/nets.py::mlp_train_step_bf16, which
    rounds to bf16 at the same points (x_0, W_l, y_l, x_l, dy_l, dx_l) and is pinned to the reference's golden vectors
    with the rounding switched off: scores, loss, every dense gradient, the embedding gradient rows and the BatchNorm
    running statistics within 2e-3 norm-wise (the fp32-accumulation order and the occasional element that rounds to the
    neighbouring bf16 value are all that differs) — not "same direction as the fp32 path"."""
    from torchrecsys_amd import ops
    from torchrecsys_amd.model import TorchRecSys
    D, hidden, B = shape
    rs = np.random.RandomState(5)
    n_u, n_i, n = 500, 300, 4096
    users = torch.from_numpy(np.concatenate([np.arange(n_u), rs.randint(0, n_u, n - n_u)]).astype(np.int64))
    items = torch.from_numpy(np.concatenate([np.arange(n_i), rs.randint(0, n_i, n - n_i)]).astype(np.int64))
    cats = [17, 5, 9][:M]
    meta = torch.from_numpy(np.stack([rs.randint(0, c, n_i) for c in cats], axis=1)) if M else None
    seed(11)
    with contextlib.redirect_stdout(io.StringIO()):
        model = TorchRecSys.from_tensors(users, items, n_users=n_u, n_items=n_i, item_metadata=meta, n_factors=D,
                                         net_type="mlp", hidden_layers=hidden, use_batch_norm=use_bn, use_amp=True,
                                         dynamic_neg_sampling=True, rng="reference")
    net = model.net
    net.train()
    dev = net.user.weight.device
    params = {k: v.detach().cpu().numpy().copy() for k, v in net.state_dict().items()}
    h = {"user": rs.randint(0, n_u, B), "pos": rs.randint(0, n_i, B), "neg": rs.randint(0, n_i, B)}
    ids = {k: torch.from_numpy(v).to(dev) for k, v in h.items()}
    batch = {"user_id": h["user"], "pos_item_id": h["pos"], "neg_item_id": h["neg"]}
    if M:
        mt = meta.numpy()
        batch["pos_metadata_id"], batch["neg_metadata_id"] = mt[h["pos"]], mt[h["neg"]]
        ids["pos_meta"] = torch.from_numpy(batch["pos_metadata_id"]).to(dev).contiguous()
        ids["neg_meta"] = torch.from_numpy(batch["neg_metadata_id"]).to(dev).contiguous()
    assert net.compute._resident(2 * B, True)
    scores, ctx = net.compute.forward(ids, 2, True)
    pos, neg = scores[:B], scores[B:]
    gp, gn = ops.hinge_backward(pos, neg)
    grads, dx0 = net.compute.backward(ctx, torch.cat([gp, gn]))
    sp, sn, loss, og = onets.mlp_train_step_bf16(params, batch)
    TOL16 = 2e-3
    assert rel_err(pos.cpu().numpy(), sp.reshape(-1)) < TOL16 and rel_err(neg.cpu().numpy(), sn.reshape(-1)) < TOL16
    named = dict(net.named_parameters())
    for k, p_ in named.items():
        if p_ in grads:
            if use_bn and k.startswith("fcs") and k.endswith("bias"):
                continue  # mathematically zero in front of train-mode BatchNorm: rounding noise on both sides
            # BatchNorm's gamma / beta gradients are cancelling sums over the batch (sum d*xhat, sum d): an element of
            # y_l that rounds to the neighbouring bf16 value moves them a little further than the GEMM outputs
            # (likewise the output layer's: g = -+1/B, the positive and the negative pass nearly cancel)
            assert rel_err(grads[p_].cpu().numpy(), og[k]) < (5e-3 if k.startswith(("bns", "output_layer")) else TOL16), k
    dx = dx0.cpu().numpy().astype(np.float64)
    gu = np.zeros((n_u, D))
    np.add.at(gu, h["user"], dx[:B, :D] + dx[B:, :D])
    gi = np.zeros((n_i, D))
    np.add.at(gi, h["pos"], dx[:B, D:2 * D])
    np.add.at(gi, h["neg"], dx[B:, D:2 * D])
    assert rel_err(gu, og["user.weight"]) < TOL16 and rel_err(gi, og["item.weight"]) < TOL16
    for m in range(M):
        gm = np.zeros((cats[m], D))
        np.add.at(gm, batch["pos_metadata_id"][:, m], dx[:B, (2 + m) * D:(3 + m) * D])
        np.add.at(gm, batch["neg_metadata_id"][:, m], dx[B:, (2 + m) * D:(3 + m) * D])
        assert rel_err(gm, og[f"metadata_embeddings.{m}.weight"]) < TOL16, m
    if use_bn:
        sd = net.state_dict()
        for l in range(len(hidden)):
            for stat in ("running_mean", "running_var"):
                assert rel_err(sd[f"bns.{l}.{stat}"].cpu().numpy(), params[f"bns.{l}.{stat}"]) < 1e-4, (l, stat)
            assert int(sd[f"bns.{l}.num_batches_tracked"]) == 2
    # plain SGD folded into the weight-gradient GEMM (MLPCompute.backward's sgd_lr): W - lr * dW with the dW checked
    # above, bit for bit (same kernels and summation order; the reduce computes (-lr * sum) + W), dW not returned, d x0
    # unchanged (the input-gradient GEMM reads the W^T image of the forward pass, not the stepped master weights)
    lrs = [0.05 * (l + 1) for l in range(len(hidden))]
    W0 = [fc.weight.data.clone() for fc in net.fcs]
    scores2, ctx2 = net.compute.forward(ids, 2, True)
    assert torch.equal(scores2, scores)
    grads2, dx02 = net.compute.backward(ctx2, torch.cat([gp, gn]), sgd_lr=lrs, g_antisymmetric=True)
    # (g_antisymmetric: the output layer's bias gradient written as the exact 0 the two per-pass sums cancel to)
    assert torch.equal(grads2[net.output_layer.bias], grads[net.output_layer.bias])
    assert float(grads2[net.output_layer.bias].abs().max()) == 0.0
    for l, fc in enumerate(net.fcs):
        assert fc.weight not in grads2 and fc.bias in grads2
        assert torch.equal(fc.weight.data, W0[l] - lrs[l] * grads[fc.weight]), l
    assert torch.equal(dx02, dx0)
    assert torch.equal(grads2[net.output_layer.weight], grads[net.output_layer.weight])


@pytest.mark.parametrize("amp", [True, False])
def test_mlp_trainer_steps_the_weights_in_the_gradient_gemm(amp, monkeypatch):
    """MLPTrainer.step (bf16-resident path / fp32 path): with momentum-free torch.optim.SGD the hidden layers' weights are
    stepped by their weight-gradient GEMMs (no .grad, the optimiser skips them) and end where the torch optimiser puts
    them (TRS_MLP_FUSED_DENSE=0: same kernels, dW materialised, torch's foreach step); with momentum, or another
    optimiser, the fold is off."""
    from torchrecsys_amd.mlp_engine import MLPTrainer
    from torchrecsys_amd.model import TorchRecSys
    rs = np.random.RandomState(3)
    n_u, n_i, n, B, D = 500, 300, 4096, 256, 64
    users = torch.from_numpy(np.concatenate([np.arange(n_u), rs.randint(0, n_u, n - n_u)]).astype(np.int64))
    items = torch.from_numpy(np.concatenate([np.arange(n_i), rs.randint(0, n_i, n - n_i)]).astype(np.int64))
    h = [{"user": rs.randint(0, n_u, B), "pos": rs.randint(0, n_i, B), "neg": rs.randint(0, n_i, B)} for _ in range(3)]
    out = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("TRS_MLP_FUSED_DENSE", fused)
        seed(11)
        with contextlib.redirect_stdout(io.StringIO()):
            model = TorchRecSys.from_tensors(users, items, n_users=n_u, n_items=n_i, n_factors=D, net_type="mlp",
                                             hidden_layers=[256, 128], use_amp=amp, dynamic_neg_sampling=True,
                                             rng="reference")
        net = model.net
        net.train()
        dev = net.user.weight.device
        assert net.compute._resident(2 * B, True) == amp
        def groups():  # (fresh dicts: an optimiser writes its defaults into the ones it is given)
            return [{"params": [p for n_, p in net.named_parameters() if not n_.startswith("fcs.0")]},
                    {"params": list(net.fcs[0].parameters()), "lr": 0.02}]
        assert MLPTrainer(net, torch.optim.SGD(groups(), lr=0.05, momentum=0.9), B)._fused_weight_lrs() is None
        assert MLPTrainer(net, torch.optim.Adam(groups(), lr=0.05), B)._fused_weight_lrs() is None
        opt = torch.optim.SGD(groups(), lr=0.05)
        tr = MLPTrainer(net, opt, B)
        assert tr._fused_weight_lrs() == ([0.02, 0.05] if fused == "1" else None)
        slot = torch.zeros(1, device=dev)
        for hb in h:
            tr.step({k: torch.from_numpy(v).to(dev) for k, v in hb.items()}, slot)
        tr.check_errors()
        torch.cuda.synchronize()
        assert (net.fcs[0].weight.grad is None) == (fused == "1")
        assert net.fcs[0].bias.grad is not None
        out[fused] = ({k: v.detach().cpu().numpy().copy() for k, v in net.state_dict().items()}, slot.item())
    for k, v in out["1"][0].items():
        if v.dtype.kind == "f":
            # the same gradients; W + (-lr) * dW is one fused multiply-add in torch's kernel and two roundings in the
            # GEMM's reduce, and three steps of bf16 GEMMs carry that last bit along
            assert rel_err(v, out["0"][0][k]) < 2e-4, k
        else:
            assert np.array_equal(v, out["0"][0][k]), k
    assert abs(out["1"][1] - out["0"][1]) <= 1e-3 * abs(out["0"][1])


@pytest.mark.parametrize("net_type", ["linear", "mlp"])
def test_amp_training_run_and_profiling_run(net_type):
    """The reference's smoke tests (tests/test_model_and_features.py:136-143, 219-227): fit() with use_amp=True and
    with profile_epochs=1 runs to the end with torch.optim.Adam at batch_size 32 (here use_amp is bf16 GEMM inputs on the
    MLP and a no-op on the scorers without GEMMs; the profiler prints its table after the first epoch)."""
    from torchrecsys_amd.model import TorchRecSys
    rs = np.random.RandomState(1)
    n_u, n_i, n = 60, 40, 1000
    df = pd.DataFrame({"user_id": np.concatenate([np.arange(n_u), rs.randint(0, n_u, n - n_u)]),
                       "item_id": np.concatenate([np.arange(n_i), rs.randint(0, n_i, n - n_i)])})
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        model = TorchRecSys(dataset=df, user_id_col="user_id", item_id_col="item_id", n_factors=16, net_type=net_type,
                            use_amp=True, use_cuda=True)
        model.fit(torch.optim.Adam(model.parameters(), lr=1e-3), epochs=1, batch_size=32)
        model = TorchRecSys(dataset=df, user_id_col="user_id", item_id_col="item_id", n_factors=16, net_type=net_type)
        model.fit(torch.optim.Adam(model.parameters(), lr=1e-3), epochs=1, batch_size=32, profile_epochs=1)
    out = buf.getvalue()
    assert out.count("Training Loss") == 2 and "Profiler Results" in out
    assert all(np.isfinite(v.cpu().numpy()).all() for v in model.state_dict().values())


@pytest.mark.parametrize("net_type,M", [("linear", 0), ("fm", 0), ("fm", 1), ("mlp", 1)])
def test_bpr_training_step_matches_the_oracle(net_type, M):
    """loss='bpr' (BASELINE.json north_star; the reference has hinge only): one fused training step on the golden G1
    batch == the oracle's step with BPR's score gradients — scores, loss, every updated parameter — on the generic staged
    path and, without metadata, on the presorted fast path too; helper.loss.bpr_loss is differentiable like hinge_loss."""
    from torchrecsys_amd.engine import SparseScorerTrainer
    from torchrecsys_amd.helper.loss import bpr_loss
    from torchrecsys_amd.mlp_engine import MLPTrainer
    from oracle import optim as ooptim
    g = load_golden(f"g1_{net_type}_M{M}.npz")
    b = golden_batch(g)
    B = b["user_id"].shape[0]
    lr = 0.05
    params = {k: v.copy() for k, v in sub(g, "init").items()}
    batch = {k: v.numpy() for k, v in b.items()}
    _, _, oloss, ograds = onets.train_forward_backward(net_type, params, batch, loss="bpr")
    ooptim.sgd_step(params, {k: v for k, v in ograds.items()}, lr)
    for path in (("generic", "fast") if (net_type != "mlp" and M == 0) else ("generic",)):
        net = build_net(net_type, M, g)
        net.train()
        opt = torch.optim.SGD(net.parameters(), lr=lr)
        dev = net.user.weight.device
        idt = torch.int32 if path == "fast" else torch.int64
        ids = {"user": b["user_id"].to(dev).to(idt), "pos": b["pos_item_id"].to(dev).to(idt),
               "neg": b["neg_item_id"].to(dev).to(idt)}
        if M:
            ids["pos_meta"] = b["pos_metadata_id"].reshape(B, M).to(dev).contiguous()
            ids["neg_meta"] = b["neg_metadata_id"].reshape(B, M).to(dev).contiguous()
        tr = (MLPTrainer if net_type == "mlp" else SparseScorerTrainer)(net, opt, B)
        tr.loss_id = 1
        slot = torch.zeros(1, device=dev)
        tr.step(ids, slot)
        tr.check_errors()
        assert abs(slot.item() / B - float(oloss)) < TOL * max(float(oloss), 1e-3), path
        for k, p_ in net.state_dict().items():
            if p_.dtype == torch.float32 and "running" not in k:
                if net_type == "mlp" and k.startswith("fcs") and k.endswith("bias"):
                    continue
                assert rel_err(p_.cpu().numpy(), params[k]) < TOL, (path, k)
    # the differentiable helper: same value, same gradients as torch's own formula
    net = build_net(net_type, M, g)
    net.train()
    pos, neg = net.forward_pair(b)
    l = bpr_loss(pos, neg)
    l.backward()
    assert abs(l.item() - float(oloss)) < TOL * max(float(oloss), 1e-3)


@pytest.mark.parametrize("net_type", ["fm", "mlp"])
def test_fit_with_sampler_options(net_type):
    """neg_sampling through the public API: k = 2 doubles the epoch's steps, reject_seen keeps the user's
    training positives out of the negatives fit() sees, training still converges; the reference-RNG mode refuses the
    options (it replays the reference's sampler)."""
    from torchrecsys_amd import ops
    from torchrecsys_amd.model import TorchRecSys
    rs = np.random.RandomState(4)
    n_u, n_i, n = 80, 500, 6000  # ~60 training positives per user out of 500 items
    users = torch.from_numpy(np.concatenate([np.arange(n_u), rs.randint(0, n_u, n - n_u)]).astype(np.int64)).to(DEV)
    items = torch.from_numpy(np.concatenate([np.arange(n_i), rs.randint(0, n_i, n - n_i)]).astype(np.int64)).to(DEV)
    seed(3)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        kw = dict(hidden_layers=[32, 16]) if net_type == "mlp" else {}
        model = TorchRecSys.from_tensors(users, items, n_users=n_u, n_items=n_i, n_factors=16, net_type=net_type,
                                         dynamic_neg_sampling=True, seed=1,
                                         neg_sampling=dict(reject_seen=True, k=2, max_tries=16), **kw)
        opt = torch.optim.SGD(model.parameters(), lr=0.1)
        runner = model.make_runner(opt, 128)
        n_train = model.data_processor.train_data["user_id"].shape[0]
        assert runner.n_train == 2 * n_train and runner.num_batches == -(-2 * n_train // 128)
        model.fit(opt, epochs=3, batch_size=128)
        model.evaluate(batch_size=128)
    losses = [float(x) for x in re.findall(r"Training Loss: ([0-9.]+)", buf.getvalue())]
    assert len(losses) == 3 and all(np.isfinite(losses))
    # the negatives of a training batch avoid the user's training positives
    st = model._device_stream("train")
    out = ops.batch_prepare(st["user"], st["pos"], None, 0x55, 0, 512, n_i, 9, 0, sampler=model._sampler())
    u, neg = out["user"].cpu().numpy(), out["neg"].cpu().numpy()
    su, si = st["user"].cpu().numpy(), st["pos"].cpu().numpy()
    pos_of = {uu: set(si[su == uu].tolist()) for uu in np.unique(u)}
    assert sum(int(nn in pos_of[uu]) for uu, nn in zip(u, neg)) == 0
    with pytest.raises(ValueError):
        with contextlib.redirect_stdout(io.StringIO()):
            TorchRecSys.from_tensors(users.cpu(), items.cpu(), n_users=n_u, n_items=n_i, n_factors=8,
                                     dynamic_neg_sampling=True, neg_sampling=dict(k=2))  # CPU tensors -> reference RNG


def test_bench_line_contract():
    """`python bench.py` prints ONE JSON line with the contract's keys, a roofline object (bound / achieved / peak / unit /
    frac / traffic) and a cpu_baseline object (value / unit / cores / kind / sample).  Run on the small c1 workload."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "c1", "--steps", "64", "--warmup", "8"],
                         capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout[-2000:]  # nothing else on stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 64 and d["warmup"] == 8 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["scaling"] == "weak" and d["data"] == "synthetic" and d["dtype"] == "f32"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] == pytest.approx(2 * 1024 * 64 / (d["ms_per_step"] * 64 * 1e-3), rel=1e-6)
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and "traffic" in r
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"]) and 0.0 < r["frac"] < 1.0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == d["unit"] and c["cores"] >= 1 and c["value"] > 0 and c["sample"]


@pytest.mark.parametrize("net_type", ["linear", "fm", "mlp"])
def test_predict_many_equals_predict(net_type):
    """predict_many (extension): every row is predict() of that user, bit for bit."""
    from torchrecsys_amd.model import TorchRecSys
    rs = np.random.RandomState(2)
    n_u, n_i, n = 50, 333, 2000
    df = pd.DataFrame({"user_id": np.concatenate([np.arange(n_u), rs.randint(0, n_u, n - n_u)]),
                       "item_id": np.concatenate([np.arange(n_i), rs.randint(0, n_i, n - n_i)])})
    with contextlib.redirect_stdout(io.StringIO()):
        model = TorchRecSys(df, "user_id", "item_id", n_factors=16, net_type=net_type)
        users = [0, 7, 49, 7]
        many = model.predict_many(users, top_k=12)
        single = [model.predict(u, top_k=12) for u in users]
    assert many.shape == (4, 12) and many.dtype == torch.int64 and not many.is_cuda
    for r in range(4):
        assert torch.equal(many[r], single[r])
    assert model.predict_many([], top_k=5).shape == (0, 5)
    with pytest.raises(IndexError):
        model.predict_many([0, n_u], top_k=3)


def test_touch_host_path_changes_nothing():
    """FitRunner.touch_host_path (bench.py calls it between the warm-up's synchronise and the clock): the host side of the
    next run_steps() call walked with ZERO steps — no kernel launched, no state advanced: the run that follows is the run
    without it, bit for bit (same batches, same tables, same losses)."""
    from torchrecsys_amd.engine import SparseScorerTrainer
    from torchrecsys_amd.model import TorchRecSys
    g = torch.Generator(device=DEV)
    g.manual_seed(1)
    n_users, n_items, n, B = 50_000, 60_000, 600_000, 2048  # sparse regime: the flag-mode step
    users = torch.randint(0, n_users, (n,), device=DEV, dtype=torch.int32, generator=g)
    items = torch.randint(0, n_items, (n,), device=DEV, dtype=torch.int32, generator=g)
    old = SparseScorerTrainer.SLICE_BATCHES
    SparseScorerTrainer.SLICE_BATCHES = 16
    try:
        res = []
        for touch in (False, True):
            seed(3)
            with contextlib.redirect_stdout(io.StringIO()):
                m = TorchRecSys.from_tensors(users, items, n_users=n_users, n_items=n_items, n_factors=32, net_type="fm",
                                             dynamic_neg_sampling=True, rng="device", seed=2)
            r = m.make_runner(torch.optim.SGD(m.parameters(), lr=0.05), B)
            m.net.train()
            r.begin_epoch()
            r.run_steps(5)
            torch.cuda.synchronize()
            if touch:
                assert r.touch_host_path() and r.next_batch == 5
                torch.cuda.synchronize()
            r.run_steps(20)  # crosses a slice boundary
            torch.cuda.synchronize()
            r.trainer.check_errors()
            res.append(({k: v.clone() for k, v in m.state_dict().items()}, r.loss_sums[:25].clone()))
    finally:
        SparseScorerTrainer.SLICE_BATCHES = old
    assert torch.allclose(res[0][1], res[1][1], rtol=1e-5, atol=0)
    for k in res[0][0]:
        # (float atomics on the few shared rows may reorder sums from run to run: last-bit differences only)
        assert torch.allclose(res[0][0][k], res[1][0][k], rtol=0, atol=1e-6), k
