# -*- coding: utf-8 -*-
"""torch twin of oracle/nets.py::mlp_train_step_bf16 for sizes the numpy oracle cannot finish in seconds (c5: 65 536
rows x [1280, 1024, 512, 256]): the same arithmetic and the same bf16 rounding points, products and sums in float64 on
the GPU.  TEST INFRASTRUCTURE; tests/test_gpu_fullsize.py pins it to the numpy oracle at a small size before using it."""
import torch

BN_EPS, BN_MOMENTUM = 1e-5, 0.1


def rnd(a):
    """fp32 -> bf16 (round-to-nearest-even) -> fp32."""
    return a.float().bfloat16().float()


WIDE = torch.float64  # accumulation type of every reduction; torch.float32 = "a plain fp32 implementation of the same
#                       rounding points", used to calibrate how ill-conditioned a gradient is at a given size


def mm(a, b):
    return (a.to(WIDE) @ b.to(WIDE)).float()


def train_step_bf16(P, ids, y_bf16=True, wide=torch.float64, dx0_bf16=False, score_grad=None, relu_masks=None):
    """wide: accumulation dtype of the reductions (float64 = the judge; float32 = the calibration run).
    score_grad: optional (g_pos, g_neg), each (B,): d loss / d score driving the backward instead of the hinge's (the
    hinge VALUE is still returned as loss) — MLPTrainer.step(score_grad=...).
    relu_masks: optional {(pass, layer): bool (B, H)} — the ReLU decisions to take instead of this function's own
    `pre-activation > 0` (the decisions of the implementation under test: at a pre-activation within rounding of the
    kink either subgradient is valid, and one flipped element moves a gradient by 1/sqrt(rows) at full size); the number
    of elements where the given decision differs from this function's own is returned in grads["relu_mask_diffs"].
    P: dict name -> fp32 GPU tensor (state_dict layout; embedding tables may be compacted); ids: dict user/pos/neg
    [/pos_meta/neg_meta] of int64 GPU tensors.  Returns (pos, neg, loss, grads, dx0 rows per pass) — running statistics
    in P are updated in place."""
    global WIDE
    WIDE = wide
    L = 0
    while f"fcs.{L}.weight" in P:
        L += 1
    M = 0
    while f"metadata_embeddings.{M}.weight" in P:
        M += 1
    use_bn = "bns.0.weight" in P
    u = ids["user"]
    W = [rnd(P[f"fcs.{l}.weight"]) for l in range(L)]

    diffs = [0]

    def forward(item, meta, ps):
        cols = [P["user.weight"][u], P["item.weight"][item]] + [P[f"metadata_embeddings.{m}.weight"][meta[:, m]]
                                                                for m in range(M)]
        x = rnd(torch.cat(cols, dim=1))
        c = {"x": [x], "y": [], "mu": [], "invstd": [], "mask": []}
        B = x.shape[0]
        for l in range(L):
            y = mm(x, W[l].T) + P[f"fcs.{l}.bias"]
            mu = invstd = None
            if use_bn:
                y64 = y.to(WIDE)
                mu = y64.mean(0).float()
                var = y64.var(0, unbiased=False).float()
                unb = var * (B / max(B - 1, 1))
                P[f"bns.{l}.running_mean"].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mu)
                P[f"bns.{l}.running_var"].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * unb)
                invstd = 1.0 / torch.sqrt(var + BN_EPS)
            if y_bf16 or not use_bn:
                y = rnd(y)
            c["y"].append(y)
            c["mu"].append(mu)
            c["invstd"].append(invstd)
            h = ((y - mu) * invstd * P[f"bns.{l}.weight"] + P[f"bns.{l}.bias"]) if use_bn else y
            mask = h > 0
            if relu_masks is not None:
                diffs[0] += int((mask != relu_masks[(ps, l)]).sum())
                mask = relu_masks[(ps, l)]
            c["mask"].append(mask)
            x = torch.where(mask, h, torch.zeros_like(h))
            if l < L - 1:
                x = rnd(x)
            c["x"].append(x)
        out = (x.to(WIDE) @ P["output_layer.weight"].to(WIDE).T).float() + P["output_layer.bias"]
        return out.reshape(-1), c

    def backward(g, c):
        g = g.reshape(-1, 1)
        gr = {}
        xL = c["x"][L]
        gr["output_layer.weight"] = (g.to(WIDE).T @ xL.to(WIDE)).float()
        gr["absbound.output_layer.weight"] = (g.to(WIDE).abs().T @ xL.to(WIDE).abs()).float()
        gr["output_layer.bias"] = g.to(WIDE).sum(0).float()
        dx = g @ P["output_layer.weight"]
        B = g.shape[0]
        for l in reversed(range(L)):
            y = c["y"][l]
            if use_bn:
                mu, invstd, gamma = c["mu"][l], c["invstd"][l], P[f"bns.{l}.weight"]
                xhat = (y - mu) * invstd
                d = torch.where(c["mask"][l], dx, torch.zeros_like(dx))
                s1 = d.to(WIDE).sum(0).float()
                s2 = (d.to(WIDE) * xhat.to(WIDE)).sum(0).float()
                gr[f"bns.{l}.weight"], gr[f"bns.{l}.bias"] = s2, s1
                gr[f"absbound.bns.{l}.bias"] = d.to(WIDE).abs().sum(0).float()
                gr[f"absbound.bns.{l}.weight"] = (d.to(WIDE) * xhat.to(WIDE)).abs().sum(0).float()
                dy = (gamma * invstd) * (d - s1 / B - xhat * (s2 / B))
            else:
                dy = torch.where(c["mask"][l], dx, torch.zeros_like(dx))
            gr[f"fcs.{l}.bias"] = dy.to(WIDE).sum(0).float()
            dy16 = rnd(dy)
            gr[f"fcs.{l}.weight"] = mm(dy16.T, c["x"][l])
            # sum of |products| per element: the scale of the fp32-accumulation error of this reduction over all rows
            gr[f"absbound.fcs.{l}.weight"] = mm(dy16.abs().T, c["x"][l].abs())
            dx = mm(dy16, W[l])
            if (l > 0 and (y_bf16 or not use_bn)) or (l == 0 and dx0_bf16):
                dx = rnd(dx)
        return gr, dx

    sp, cp = forward(ids["pos"], ids.get("pos_meta"), 0)
    sn, cn = forward(ids["neg"], ids.get("neg_meta"), 1)
    h = sn - sp + 1.0
    B = h.shape[0]
    act = (h >= 0).float() / B
    loss = torch.clamp(h, min=0).mean()
    g_pos, g_neg = (-act, act) if score_grad is None else (score_grad[0].float(), score_grad[1].float())
    gp, dxp = backward(g_pos, cp)
    gn, dxn = backward(g_neg, cn)
    grads = {k: gp[k] + gn[k] for k in gp}  # (absbound.*: the two passes' bounds add up as well)
    if relu_masks is not None:
        grads["relu_mask_diffs"] = diffs[0]
    return sp, sn, loss, grads, (dxp, dxn)
