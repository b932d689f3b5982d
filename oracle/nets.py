# -*- coding: utf-8 -*-
"""Closed-form forward / backward of the three scorers and the hinge loss (numpy fp32).  TEST INFRASTRUCTURE.

Parameter dictionaries use the reference's state_dict names without the leading "net." (SURVEY.md §5):
  Linear: user.weight, item.weight, user_bias.weight, item_bias.weight, metadata.{m}.weight
  FM:     user.weight, item.weight, linear_user.weight, linear_item.weight, metadata.{m}.weight,
          linear_metadata.{m}.weight
  MLP:    user.weight, item.weight, metadata_embeddings.{m}.weight, fcs.{l}.weight/bias,
          bns.{l}.weight/bias/running_mean/running_var/num_batches_tracked, output_layer.weight/bias
Batches: user (B,), item (B,), meta (B,M) int64 — the (B,M) contract of the reference nets
(collaborative/linear.py:72-75, fm.py:75-79, mlp.py:98-102).
Gradients are returned as DENSE-EQUIVALENT arrays (what `p.grad.to_dense()` holds in the reference after
loss.backward(), model.py:197).
"""
import numpy as np

F32 = np.float32


def _n_meta(params, prefix):
    m = 0
    while f"{prefix}.{m}.weight" in params:
        m += 1
    return m


# ----------------------------------------------------------------------------------------------- hinge
def hinge_loss(pos, neg):
    """helper/loss.py:5-9: mean(clamp(neg - pos + 1, min=0))."""
    h = (neg.astype(F32) - pos.astype(F32) + F32(1.0)).astype(F32)
    return np.maximum(h, F32(0)).mean(dtype=F32)


def hinge_grad(pos, neg):
    """d mean-hinge / d pos, d neg.  torch.clamp(min=0) passes the gradient where the input is >= 0."""
    h = (neg.astype(F32) - pos.astype(F32) + F32(1.0)).astype(F32)
    a = (h >= 0).astype(F32) / F32(h.size)
    return (-a).astype(F32), a.astype(F32)


def auc_score(pos, neg):
    """evaluate/metrics.py:23-31: (pos > neg).sum() / len(pos)."""
    return float((pos > neg).sum()) / float(len(pos))


# ----------------------------------------------------------------------------------------------- Linear
def linear_forward(params, user, item, meta=None):
    """collaborative/linear.py:54-80 -> (B,1)."""
    u = params["user.weight"][user]
    it = params["item.weight"][item].copy()
    for m in range(_n_meta(params, "metadata")):
        it += params[f"metadata.{m}.weight"][meta[:, m]]  # linear.py:75
    dot = (u * it).sum(axis=1, dtype=F32).reshape(-1, 1)
    return (dot + params["user_bias.weight"][user] + params["item_bias.weight"][item]).astype(F32)


def linear_backward(params, user, item, meta, g):
    """Accumulate d/d params of sum(g * score) for one pass into a dict of dense arrays.  g: (B,) or (B,1)."""
    g = np.asarray(g, F32).reshape(-1, 1)
    M = _n_meta(params, "metadata")
    grads = {k: np.zeros_like(v) for k, v in params.items()}
    u = params["user.weight"][user]
    it = params["item.weight"][item].copy()
    for m in range(M):
        it += params[f"metadata.{m}.weight"][meta[:, m]]
    np.add.at(grads["user.weight"], user, g * it)
    np.add.at(grads["item.weight"], item, g * u)
    for m in range(M):
        np.add.at(grads[f"metadata.{m}.weight"], meta[:, m], g * u)
    np.add.at(grads["user_bias.weight"], user, g)
    np.add.at(grads["item_bias.weight"], item, g)
    return grads


# ----------------------------------------------------------------------------------------------- FM
def _sigmoid(z):
    return (F32(1.0) / (F32(1.0) + np.exp(-z, dtype=F32))).astype(F32)


def _fm_fields(params, user, item, meta):
    M = _n_meta(params, "metadata")
    fields = [params["user.weight"][user], params["item.weight"][item]]
    lins = [params["linear_user.weight"][user], params["linear_item.weight"][item]]
    for m in range(M):
        fields.append(params[f"metadata.{m}.weight"][meta[:, m]])
        lins.append(params[f"linear_metadata.{m}.weight"][meta[:, m]])
    return np.stack(fields, axis=1), np.concatenate(lins, axis=1)  # (B,F,D), (B,F)


def fm_forward(params, user, item, meta=None):
    """collaborative/fm.py:60-101 -> (B,) sigmoid(linear + pairwise)."""
    emb, lin = _fm_fields(params, user, item, meta)
    power_of_sum = emb.sum(axis=1, dtype=F32) ** 2          # fm.py:83
    sum_of_power = (emb ** 2).sum(axis=1, dtype=F32)        # fm.py:84
    pairwise = (power_of_sum - sum_of_power).sum(axis=1, dtype=F32) * F32(0.5)  # fm.py:86
    linear = lin.sum(axis=1, dtype=F32)                     # fm.py:97
    return _sigmoid((linear + pairwise).astype(F32))


def fm_backward(params, user, item, meta, g):
    """d/d params of sum(g * score) for one pass.  SURVEY App. A.3: g_z = g*s*(1-s); dv_f = g_z*(S - v_f); dw_f = g_z."""
    g = np.asarray(g, F32).reshape(-1)
    M = _n_meta(params, "metadata")
    emb, _ = _fm_fields(params, user, item, meta)
    s = fm_forward(params, user, item, meta)
    gz = (g * ((F32(1.0) - s) * s)).astype(F32).reshape(-1, 1)
    S = emb.sum(axis=1, dtype=F32)
    grads = {k: np.zeros_like(v) for k, v in params.items()}
    ids = [("user", user), ("item", item)] + [(f"metadata.{m}", meta[:, m]) for m in range(M)]
    lin_names = ["linear_user", "linear_item"] + [f"linear_metadata.{m}" for m in range(M)]
    for f, ((name, idx), lname) in enumerate(zip(ids, lin_names)):
        np.add.at(grads[f"{name}.weight"], idx, gz * (S - emb[:, f, :]))
        np.add.at(grads[f"{lname}.weight"], idx, gz)
    return grads


# ----------------------------------------------------------------------------------------------- MLP
BN_EPS = F32(1e-5)
BN_MOMENTUM = F32(0.1)


def _mlp_dims(params):
    L = 0
    while f"fcs.{L}.weight" in params:
        L += 1
    return L, _n_meta(params, "metadata_embeddings"), "bns.0.weight" in params


def mlp_forward(params, user, item, meta=None, training=True, update_running=True):
    """collaborative/mlp.py:88-115 -> (B,1).  Returns (score, cache); in training mode the BatchNorm1d running
    statistics inside `params` are updated in place exactly once per call (so twice per training step: positive pass
    first, then negative pass — SURVEY App. A.4)."""
    L, M, use_bn = _mlp_dims(params)
    cols = [params["user.weight"][user], params["item.weight"][item]]
    for m in range(M):
        cols.append(params[f"metadata_embeddings.{m}.weight"][meta[:, m]])
    x = np.concatenate(cols, axis=1).astype(F32)
    cache = {"x": [x], "y": [], "xhat": [], "invstd": [], "pre_relu": []}
    B = x.shape[0]
    for l in range(L):
        y = (x @ params[f"fcs.{l}.weight"].T + params[f"fcs.{l}.bias"]).astype(F32)
        cache["y"].append(y)
        if use_bn:
            if training:
                mu = y.mean(axis=0, dtype=F32)
                var = y.var(axis=0, dtype=F32)  # biased
                if update_running:
                    unb = var * F32(B) / F32(max(B - 1, 1))
                    rm, rv = params[f"bns.{l}.running_mean"], params[f"bns.{l}.running_var"]
                    rm[...] = (F32(1) - BN_MOMENTUM) * rm + BN_MOMENTUM * mu
                    rv[...] = (F32(1) - BN_MOMENTUM) * rv + BN_MOMENTUM * unb
                    params[f"bns.{l}.num_batches_tracked"] += 1
            else:
                mu, var = params[f"bns.{l}.running_mean"], params[f"bns.{l}.running_var"]
            invstd = (F32(1) / np.sqrt(var + BN_EPS)).astype(F32)
            xhat = ((y - mu) * invstd).astype(F32)
            y = (xhat * params[f"bns.{l}.weight"] + params[f"bns.{l}.bias"]).astype(F32)
            cache["xhat"].append(xhat)
            cache["invstd"].append(invstd)
        cache["pre_relu"].append(y)
        x = np.maximum(y, F32(0))
        cache["x"].append(x)
    out = (x @ params["output_layer.weight"].T + params["output_layer.bias"]).astype(F32)
    return out, cache


def mlp_backward(params, user, item, meta, g, cache, training=True):
    """d/d params of sum(g * score) for one pass, given the forward cache."""
    L, M, use_bn = _mlp_dims(params)
    g = np.asarray(g, F32).reshape(-1, 1)
    grads = {k: np.zeros_like(v) for k, v in params.items() if v.dtype == np.float32 and "running" not in k}
    x = cache["x"][L]
    grads["output_layer.weight"] = (g.T @ x).astype(F32)
    grads["output_layer.bias"] = g.sum(axis=0, dtype=F32)
    dx = (g @ params["output_layer.weight"]).astype(F32)
    B = g.shape[0]
    for l in reversed(range(L)):
        dy = dx * (cache["pre_relu"][l] > 0)
        if use_bn:
            xhat, invstd = cache["xhat"][l], cache["invstd"][l]
            gamma = params[f"bns.{l}.weight"]
            grads[f"bns.{l}.weight"] = (dy * xhat).sum(axis=0, dtype=F32)
            grads[f"bns.{l}.bias"] = dy.sum(axis=0, dtype=F32)
            if training:
                dxhat = dy * gamma
                dy = (invstd / F32(B)) * (F32(B) * dxhat - dxhat.sum(axis=0, dtype=F32)
                                           - xhat * (dxhat * xhat).sum(axis=0, dtype=F32))
            else:
                dy = dy * gamma * invstd
            dy = dy.astype(F32)
        grads[f"fcs.{l}.weight"] = (dy.T @ cache["x"][l]).astype(F32)
        grads[f"fcs.{l}.bias"] = dy.sum(axis=0, dtype=F32)
        dx = (dy @ params[f"fcs.{l}.weight"]).astype(F32)
    D = params["user.weight"].shape[1]
    np.add.at(grads["user.weight"], user, dx[:, 0:D])
    np.add.at(grads["item.weight"], item, dx[:, D:2 * D])
    for m in range(M):
        np.add.at(grads[f"metadata_embeddings.{m}.weight"], meta[:, m], dx[:, (2 + m) * D:(3 + m) * D])
    return grads


# ----------------------------------------------------------------------------------------------- MLP, bf16-resident
def bf16_round(a):
    """Round-to-nearest-even to bfloat16, returned as float32 (what the kernels store in a bf16 image)."""
    a = np.ascontiguousarray(a, F32)
    u = a.view(np.uint32).astype(np.uint64)
    r = ((u >> np.uint64(16)) & np.uint64(1)) + np.uint64(0x7FFF)
    return ((u + r) & np.uint64(0xFFFF0000)).astype(np.uint32).view(F32).reshape(a.shape)


def _mm(a, b):
    """fp32-accumulated product of bf16-valued operands: products are exact in fp32, the sum is taken in float64 and
    rounded once (the MFMA's fp32 accumulation order is not restated; it differs at the 1e-7 level)."""
    return (a.astype(np.float64) @ b.astype(np.float64)).astype(F32)


def mlp_train_step_bf16(params, batch, rnd=bf16_round, y_bf16=True, dx0_bf16=False):
    """use_amp=True on tile-aligned nets (torchrecsys_amd/mlp_engine.py, the bf16-RESIDENT path): the arithmetic of
    mlp_forward / mlp_backward with a rounding to bf16 at exactly the points where the product path keeps a bf16 image
    in HBM — the gathered input x_0, every weight image W_l (forward and input gradient), the pre-BN outputs y_l (when
    their statistics come from the fp32 accumulators of the same launch: y_bf16; statistics are taken BEFORE the
    rounding), the layer inputs x_l (l < L; the last one feeds the fp32 H -> 1 dot), the BN-backward outputs dy_l and
    the input gradients dx_l (l > 0; dx_0 = the embedding gradient stays fp32 unless dx0_bf16: the fused SGD embedding
    update of the trainer reads a bf16 d x0 — what autocast's gradient of the half-precision x_0 is).  Bias / gamma / beta gradients and dW
    are fp32 sums.  rnd=identity restates the fp32 path (tests/test_oracle_golden.py pins that against the fp32
    functions above, which the reference's golden vectors pin).  The reference's own AMP is fp16 autocast + GradScaler
    and CUDA-only (model.py:86-88,192-195): there is no reference output to pin the ROUNDED variant to — it is this
    repository's statement of where it rounds.
    Returns (pos, neg, loss, grads) like train_forward_backward; running statistics in `params` are updated."""
    L, M, use_bn = _mlp_dims(params)
    u, p, n = batch["user_id"], batch["pos_item_id"], batch["neg_item_id"]
    pm, nm = batch.get("pos_metadata_id"), batch.get("neg_metadata_id")
    W = [rnd(params[f"fcs.{l}.weight"]) for l in range(L)]

    def forward(item, meta):
        cols = [params["user.weight"][u], params["item.weight"][item]]
        for m in range(M):
            cols.append(params[f"metadata_embeddings.{m}.weight"][meta[:, m]])
        x = rnd(np.concatenate(cols, axis=1))
        c = {"x": [x], "y": [], "mu": [], "invstd": []}
        B = x.shape[0]
        for l in range(L):
            y = (_mm(x, W[l].T) + params[f"fcs.{l}.bias"]).astype(F32)
            mu = invstd = None
            if use_bn:
                y64 = y.astype(np.float64)
                mu = y64.mean(axis=0).astype(F32)
                var = y64.var(axis=0).astype(F32)
                unb = (var * F32(B) / F32(max(B - 1, 1))).astype(F32)
                rm, rv = params[f"bns.{l}.running_mean"], params[f"bns.{l}.running_var"]
                rm[...] = (F32(1) - BN_MOMENTUM) * rm + BN_MOMENTUM * mu
                rv[...] = (F32(1) - BN_MOMENTUM) * rv + BN_MOMENTUM * unb
                params[f"bns.{l}.num_batches_tracked"] += 1
                invstd = (F32(1) / np.sqrt(var + BN_EPS)).astype(F32)
            if y_bf16 or not use_bn:
                y = rnd(y)
            c["y"].append(y)
            c["mu"].append(mu)
            c["invstd"].append(invstd)
            h = ((y - mu) * invstd * params[f"bns.{l}.weight"] + params[f"bns.{l}.bias"]).astype(F32) if use_bn else y
            x = np.maximum(h, F32(0))
            if l < L - 1:
                x = rnd(x)
            c["x"].append(x)
        out = (x.astype(np.float64) @ params["output_layer.weight"].astype(np.float64).T).astype(F32) \
            + params["output_layer.bias"]
        return out.astype(F32), c

    def backward(item, meta, g, c):
        g = np.asarray(g, F32).reshape(-1, 1)
        grads = {k: np.zeros_like(v) for k, v in params.items() if v.dtype == np.float32 and "running" not in k}
        xL = c["x"][L]
        grads["output_layer.weight"] = (g.astype(np.float64).T @ xL.astype(np.float64)).astype(F32)
        grads["output_layer.bias"] = g.sum(axis=0, dtype=np.float64).astype(F32)
        dx = (g @ params["output_layer.weight"]).astype(F32)
        B = g.shape[0]
        for l in reversed(range(L)):
            y = c["y"][l]
            if use_bn:
                mu, invstd, gamma = c["mu"][l], c["invstd"][l], params[f"bns.{l}.weight"]
                xhat = ((y - mu) * invstd).astype(F32)
                d = np.where(xhat * gamma + params[f"bns.{l}.bias"] > 0, dx, F32(0)).astype(F32)
                s1 = d.sum(axis=0, dtype=np.float64).astype(F32)
                s2 = (d.astype(np.float64) * xhat).sum(axis=0).astype(F32)
                grads[f"bns.{l}.weight"], grads[f"bns.{l}.bias"] = s2, s1
                dy = ((gamma * invstd) * (d - s1 / F32(B) - xhat * (s2 / F32(B)))).astype(F32)
            else:
                dy = np.where(y > 0, dx, F32(0)).astype(F32)
            grads[f"fcs.{l}.bias"] = dy.sum(axis=0, dtype=np.float64).astype(F32)
            dy16 = rnd(dy)
            grads[f"fcs.{l}.weight"] = _mm(dy16.T, c["x"][l])
            dx = _mm(dy16, W[l])
            if (l > 0 and (y_bf16 or not use_bn)) or (l == 0 and dx0_bf16):  # the layer below keeps a bf16 y: its dx image is bf16 too
                dx = rnd(dx)
        D = params["user.weight"].shape[1]
        np.add.at(grads["user.weight"], u, dx[:, 0:D])
        np.add.at(grads["item.weight"], item, dx[:, D:2 * D])
        for m in range(M):
            np.add.at(grads[f"metadata_embeddings.{m}.weight"], meta[:, m], dx[:, (2 + m) * D:(3 + m) * D])
        return grads

    sp, cp = forward(p, pm)
    sn, cn = forward(n, nm)
    gp, gn = hinge_grad(sp, sn)
    return sp, sn, hinge_loss(sp, sn), _add(backward(p, pm, gp, cp), backward(n, nm, gn, cn))


# ----------------------------------------------------------------------------------------------- one step
def _add(a, b):
    return {k: (a[k] + b[k]).astype(F32) for k in a}


def bpr_loss(pos, neg):
    """mean(-log sigmoid(pos - neg)) = mean softplus(neg - pos).  NOT in the reference (helper/loss.py holds hinge_loss
    only): BASELINE.json's north_star names it beside hinge; pinned by its formula against torch autograd
    (tests/test_oracle_golden.py), there is no reference output for it."""
    x = (np.asarray(neg, np.float64) - np.asarray(pos, np.float64)).reshape(-1)
    return F32(np.mean(np.maximum(x, 0) + np.log1p(np.exp(-np.abs(x)))))


def bpr_grad(pos, neg):
    """d bpr_loss / d pos, d neg: -+ sigmoid(neg - pos) / B."""
    x = (np.asarray(neg, F32) - np.asarray(pos, F32)).astype(F32)
    s = (F32(1) / (F32(1) + np.exp(-x))).astype(F32) / F32(x.size)
    return (-s).astype(F32).reshape(np.shape(pos)), s.astype(F32).reshape(np.shape(neg))


def train_forward_backward(net_type, params, batch, loss="hinge"):
    """TorchRecSys.forward + hinge_loss + loss.backward() (model.py:171-185, 280-284, 188-197) on one batch dict
    with keys user_id, pos_item_id, neg_item_id[, pos_metadata_id, neg_metadata_id].
    Returns (pos_score, neg_score, loss, dense-equivalent grads).  MLP: `params` running stats are updated in place."""
    u, p, n = batch["user_id"], batch["pos_item_id"], batch["neg_item_id"]
    pm, nm = batch.get("pos_metadata_id"), batch.get("neg_metadata_id")
    _lg = bpr_grad if loss == "bpr" else hinge_grad
    if net_type == "linear":
        sp, sn = linear_forward(params, u, p, pm), linear_forward(params, u, n, nm)
        gp, gn = _lg(sp, sn)
        grads = _add(linear_backward(params, u, p, pm, gp), linear_backward(params, u, n, nm, gn))
    elif net_type == "fm":
        sp, sn = fm_forward(params, u, p, pm), fm_forward(params, u, n, nm)
        gp, gn = _lg(sp, sn)
        grads = _add(fm_backward(params, u, p, pm, gp), fm_backward(params, u, n, nm, gn))
    elif net_type == "mlp":
        sp, cp = mlp_forward(params, u, p, pm, training=True)
        sn, cn = mlp_forward(params, u, n, nm, training=True)
        gp, gn = _lg(sp, sn)
        grads = _add(mlp_backward(params, u, p, pm, gp, cp), mlp_backward(params, u, n, nm, gn, cn))
    else:
        raise ValueError(net_type)
    return sp, sn, (bpr_loss if loss == "bpr" else hinge_loss)(sp, sn), grads


def touched_rows(net_type, params, batch):
    """Rows 'present in the batch' per embedding table (they appear in the sparse COO gradient even when their
    gradient is zero — SURVEY App. A.5).  Returns {param name: sorted unique row ids}."""
    u, p, n = batch["user_id"], batch["pos_item_id"], batch["neg_item_id"]
    pm, nm = batch.get("pos_metadata_id"), batch.get("neg_metadata_id")
    items = np.unique(np.concatenate([p, n]))
    users = np.unique(u)
    out = {"user.weight": users, "item.weight": items}
    if net_type == "linear":
        out["user_bias.weight"], out["item_bias.weight"] = users, items
        mp = "metadata"
    elif net_type == "fm":
        out["linear_user.weight"], out["linear_item.weight"] = users, items
        mp = "metadata"
    else:
        mp = "metadata_embeddings"
    for m in range(_n_meta(params, mp)):
        rows = np.unique(np.concatenate([pm[:, m], nm[:, m]]))
        out[f"{mp}.{m}.weight"] = rows
        if net_type == "fm":
            out[f"linear_metadata.{m}.weight"] = rows
    return out


def topk(scores, k):
    """predict(): torch.sort(scores, descending=True)[1][:k] (model.py:447-450); ties broken by ascending index
    (the reference's order for ties is unspecified — SURVEY §3.5)."""
    scores = np.asarray(scores, F32).reshape(-1)
    order = np.lexsort((np.arange(scores.size), -scores.astype(np.float64)))
    return order[:k].astype(np.int64)
