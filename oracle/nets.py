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


# ----------------------------------------------------------------------------------------------- one step
def _add(a, b):
    return {k: (a[k] + b[k]).astype(F32) for k in a}


def train_forward_backward(net_type, params, batch):
    """TorchRecSys.forward + hinge_loss + loss.backward() (model.py:171-185, 280-284, 188-197) on one batch dict
    with keys user_id, pos_item_id, neg_item_id[, pos_metadata_id, neg_metadata_id].
    Returns (pos_score, neg_score, loss, dense-equivalent grads).  MLP: `params` running stats are updated in place."""
    u, p, n = batch["user_id"], batch["pos_item_id"], batch["neg_item_id"]
    pm, nm = batch.get("pos_metadata_id"), batch.get("neg_metadata_id")
    if net_type == "linear":
        sp, sn = linear_forward(params, u, p, pm), linear_forward(params, u, n, nm)
        gp, gn = hinge_grad(sp, sn)
        grads = _add(linear_backward(params, u, p, pm, gp), linear_backward(params, u, n, nm, gn))
    elif net_type == "fm":
        sp, sn = fm_forward(params, u, p, pm), fm_forward(params, u, n, nm)
        gp, gn = hinge_grad(sp, sn)
        grads = _add(fm_backward(params, u, p, pm, gp), fm_backward(params, u, n, nm, gn))
    elif net_type == "mlp":
        sp, cp = mlp_forward(params, u, p, pm, training=True)
        sn, cn = mlp_forward(params, u, n, nm, training=True)
        gp, gn = hinge_grad(sp, sn)
        grads = _add(mlp_backward(params, u, p, pm, gp, cp), mlp_backward(params, u, n, nm, gn, cn))
    else:
        raise ValueError(net_type)
    return sp, sn, hinge_loss(sp, sn), grads


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
