# -*- coding: utf-8 -*-
"""Forward / backward / training step of the MLP scorer on the HIP kernels (reference collaborative/mlp.py:88-115,
model.py:171-200).

Layout: both scoring passes of a step are stacked — rows [0,B) positive, rows [B,2B) negative — so every dense layer
is ONE fp32-MFMA GEMM over 2B rows while BatchNorm statistics stay per pass (as the reference's two net.forward calls
produce them).  Saved for backward per layer l: the layer input x_l and the pre-BN output y_l; the normalised /
rectified values are recomputed from y_l and the batch statistics.

Dense parameters (fcs / bns / output_layer) get ordinary dense gradients in ONE flat buffer (dist.FlatGradBucket: a
single RCCL all-reduce per step under data parallelism) and are stepped by the user's torch optimiser; embedding
tables take the fused sparse-row paths of engine.py from the column blocks of d x0.
"""
import torch

from . import dist as tdist
from . import ops
from .engine import RowState, _group_of, apply_rows, classify_optimizer

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


class MLPCompute:
    """Kernel orchestration for one MLP module.  forward() returns (scores, ctx); backward(ctx, g) returns
    ({dense param: grad}, d x0)."""

    def __init__(self, net):
        self.net = net
        self.gemm_events = None  # bench.py: list of (start event, end event, flops) per GEMM launch when not None
        self.GEMM_EVENT_EVERY = 5   # ... of one training step in 5 (18 event records cost a step ~0.1 ms)
        self.gemm_steps_seen = self.gemm_steps_timed = 0
        self._time_gemms = False
        # Synchronised BatchNorm under data parallelism: train-mode statistics over the GLOBAL batch — per
        # layer and pass one all-reduce of (E[y], E[y^2]) in the forward and of (sum d, sum d*xhat) in the backward,
        # 2*H floats each — so N ranks with batch B compute what one process computes with batch N*B.  Off (default):
        # per-replica statistics.  Only read when torch.distributed has more than one rank.
        self.sync_bn = False
        # the embedding gradient d x0 as a bf16 image on the bf16-resident path (what autocast's gradient of the
        # half-precision x0 is): set by MLPTrainer when its fused SGD embedding update consumes it; the autograd bridge
        # and the per-table optimiser paths keep fp32
        self.dx0_bf16 = False
        # layer 0 as one launch (gather inside the GEMM); TRS_MLP_FUSED_GATHER=0: gather-concat + GEMM (A/B knob, tests)
        import os
        self.fused_gather = os.environ.get("TRS_MLP_FUSED_GATHER", "1") != "0"

    def _resident(self, rows, training):
        """bf16-resident path: use_amp, training step, every GEMM of the net made of interior tiles."""
        net = self.net
        if not (net.use_bf16 and training):
            return False
        dims = [net.input_shape] + [fc.out_features for fc in net.fcs]
        return rows % 128 == 0 and all(d % 128 == 0 for d in dims)

    def _refresh_weight_images(self):
        """bf16 images of the fp32 master weights, (H_out, H_in) for the forward and transposed (H_in, H_out) for the
        input gradient; rewritten every step (the optimiser updates the fp32 weights)."""
        net = self.net
        if getattr(self, "w16", None) is None:
            dev = net.fcs[0].weight.device
            self.w16 = [torch.empty(fc.weight.shape, dtype=torch.bfloat16, device=dev) for fc in net.fcs]
            self.w16t = [torch.empty(fc.weight.shape[::-1], dtype=torch.bfloat16, device=dev) for fc in net.fcs]
        srcs = [fc.weight.data for fc in net.fcs]
        if len(srcs) > ops.WeightImages.MAX:
            for src, w, wt in zip(srcs, self.w16, self.w16t):
                ops.f32_to_bf16(src, w, wt)
            return
        wi = getattr(self, "_wimg", None)
        if wi is None or wi.key != tuple(t.data_ptr() for t in srcs):  # (weights re-allocated: load_state_dict, .to())
            wi = self._wimg = ops.WeightImages(srcs, self.w16, self.w16t)
        wi.refresh()  # every layer's two images, one launch

    def _gemm16(self, tn, A, B, **kw):
        ev = self.gemm_events
        if ev is None or not self._time_gemms:
            return ops.gemm_bf16in(tn, A, B, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = ops.gemm_bf16in(tn, A, B, **kw)
        e1.record()
        ev.append((e0, e1, 2.0 * out.shape[0] * out.shape[1] * (A.shape[0] if tn else A.shape[1])))
        return out

    def _gemm(self, *a, **kw):
        ev = self.gemm_events
        if ev is None or not self._time_gemms:
            return ops.gemm(*a, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = ops.gemm(*a, **kw)
        e1.record()
        A = a[2]
        K = A.shape[0] if a[0] else A.shape[1]
        ev.append((e0, e1, 2.0 * out.shape[0] * out.shape[1] * K))
        return out

    def _dims(self):
        net = self.net
        return net.n_factors, net.n_meta_tables(), len(net.fcs), net.use_batch_norm

    def forward(self, ids, passes, training, bucket=None):
        """ids: dict user/pos[/neg][/pos_meta/neg_meta] of GPU tensors; passes = 2 scores (pos, neg) stacked."""
        net = self.net
        if self.gemm_events is not None and training:  # a training step begins: is it one of the timed ones?
            self._time_gemms = self.gemm_steps_seen % self.GEMM_EVENT_EVERY == 0
            self.gemm_steps_seen += 1
            self.gemm_steps_timed += int(self._time_gemms)
        D, M, L, use_bn = self._dims()
        dev = net.user.weight.device
        B = ids["user"].shape[0]
        rows = passes * B
        err = net._err_flag()
        Bt, keep = ops.make_batch(ids["user"], ids["pos"], ids.get("neg") if passes == 2 else None,
                                  ids.get("pos_meta"), ids.get("neg_meta") if passes == 2 else None, err)
        # use_amp on shapes the bf16-resident kernels take: layer inputs x_l live in HBM as bf16 only (what the forward
        # and weight-gradient GEMMs read), pre-BN outputs y_l stay fp32 (statistics, backward recompute)
        res = self._resident(rows, training)
        # decided ONCE per forward and recorded in ctx: the backward must branch on what the forward did (x_L stored or
        # not, statistics of the global batch or of the replica), not on flags that may have changed since
        sync_fwd = bool(training and use_bn and self.sync_bn and tdist.world_info()[1] > 1)
        if res:
            self._refresh_weight_images()
        # x0 = the concatenated embedding rows.  Layer 0 runs as ONE launch when the shape allows (the gather inside the
        # GEMM's A-operand load, ops.mlp_gather_gemm1 — it writes the x0 image the weight-gradient GEMM needs as a
        # by-product, and only when there will be a backward pass); otherwise gather-concat, then the GEMM
        x_dtype = torch.bfloat16 if res else torch.float32
        need_x0 = training or L == 0
        x = torch.empty((rows, net.input_shape), dtype=x_dtype, device=dev) if need_x0 else None
        gathered = False  # x holds x0
        ctx = {"ids": ids, "B": B, "passes": passes, "x": [x], "y": [], "mean": [], "var": [], "training": training,
               "resident": res, "Bt": (Bt, keep), "sync": sync_fwd}
        tracked = []
        out = torch.empty(rows, dtype=torch.float32, device=dev)  # the scores
        for l in range(L):
            fc = net.fcs[l]
            # train-mode BN: the batch statistics come out of the GEMM epilogue (one partial per 128-row tile) when no
            # tile straddles the two passes
            fuse_stats = use_bn and training and (passes == 1 or B % ops.GEMM_TILE_ROWS == 0)
            part = None
            if fuse_stats:
                n_tiles = (rows + ops.GEMM_TILE_ROWS - 1) // ops.GEMM_TILE_ROWS
                part = torch.empty((n_tiles, 2, fc.out_features), dtype=torch.float32, device=dev)
            y = None
            if l == 0 and self.fused_gather and ids["user"].dtype == torch.int32 and (res or not net.use_bf16):
                # (int32 ids: what the device loader and the presort produce; the autograd bridge's int64 ids fall through)
                y = torch.empty((rows, fc.out_features), device=dev,
                                dtype=torch.bfloat16 if (res and (fuse_stats or not use_bn)) else torch.float32)
                timed = self.gemm_events is not None and self._time_gemms
                if timed:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                if not ops.mlp_gather_gemm1(net.tables(), Bt, passes, self.w16[0] if res else fc.weight.data, fc.bias.data,
                                            y, part, x):
                    y = None
                elif timed:
                    e1.record()
                    self.gemm_events.append((e0, e1, 2.0 * rows * fc.out_features * net.input_shape))
            if y is not None:
                gathered = need_x0  # the fused launch wrote the x0 image (when one was asked for)
            else:
                if l == 0 and not gathered:
                    if x is None:
                        x = torch.empty((rows, net.input_shape), dtype=x_dtype, device=dev)
                    ops.mlp_gather_concat(net.tables(), Bt, passes, **({"x16": x} if res else {"x": x}))
                    gathered = True
                    ctx["x"][0] = x
                if res:
                    # y_l rounded to bf16 when its statistics come from the fp32 accumulators of the same launch (or
                    # there is no BatchNorm): what autocast's half-precision linear output is
                    y = self._gemm16(False, x, self.w16[l], bias=fc.bias.data, bn_part=part,
                                     out_bf16=fuse_stats or not use_bn)
                else:
                    y = self._gemm(False, True, x, fc.weight.data, bias=fc.bias.data, bf16=net.use_bf16, bn_part=part)
            ctx["y"].append(y)
            mean = var = gamma = beta = None
            stat_passes = 1
            run = {}
            if use_bn:
                bn = net.bns[l]
                gamma, beta = bn.weight.data, bn.bias.data
                if training:
                    H = y.shape[1]
                    mean = torch.empty((passes, H), dtype=torch.float32, device=dev)
                    var = torch.empty((passes, H), dtype=torch.float32, device=dev)
                    sync = sync_fwd
                    # the running statistics' momentum update rides in the BN+ReLU launch below
                    if not sync:
                        run = {"momentum": BN_MOMENTUM, "running_mean": bn.running_mean, "running_var": bn.running_var,
                               "tracked": bn.num_batches_tracked}
                    if fuse_stats:
                        ops.bn_stats_finalize(part, B, ops.GEMM_TILE_ROWS, H, passes, BN_MOMENTUM, mean, var, None, None)
                    else:
                        ops.bn_batch_stats(y, B, passes, BN_MOMENTUM, mean, var, None, None)
                    if sync:
                        self._sync_stats(mean, var, bn, B)
                    if sync:  # (otherwise the counter rides with the running update)
                        tracked.append(bn.num_batches_tracked)
                    stat_passes = passes
                else:
                    mean, var = bn.running_mean, bn.running_var
            ctx["mean"].append(mean)
            ctx["var"].append(var)
            if res and l < L - 1:  # the next layer's input: bf16 only (the last layer's output feeds the fp32 H -> 1 dot)
                xn = torch.empty(y.shape, dtype=torch.bfloat16, device=dev)
                ops.bn_relu_forward(y, B, passes, use_bn, stat_passes, mean, var, gamma, beta, BN_EPS, out16=xn, **run)
            else:
                xn = torch.empty(y.shape, dtype=torch.float32, device=dev)
                if l == L - 1:  # the H -> 1 output layer rides in the last hidden layer's BN + ReLU launch
                    run = dict(run, dot=(net.output_layer.weight.data.reshape(-1), net.output_layer.bias.data, out))
                    # ... and with BatchNorm the last activations are not stored at all: the backward's reduce kernel
                    # recomputes them for the output layer's weight gradient (outer_xw); without BatchNorm there is no
                    # reduce pass, under sync-BN it is a separate phase: those keep x_L
                    if use_bn and ops.bn_relu_forward_forms_dot(y) and not sync_fwd:
                        xn = None
                ops.bn_relu_forward(y, B, passes, use_bn, stat_passes, mean, var, gamma, beta, BN_EPS, xn, **run)
            x = xn
            ctx["x"].append(x)
        if tracked:  # BatchNorm1d.num_batches_tracked of every layer: + passes, one launch
            torch._foreach_add_(tracked, passes)
        if L == 0:
            ops.mlp_gather_concat(net.tables(), Bt, passes, x)
            ops.rowdot(x, net.output_layer.weight.data.reshape(-1), net.output_layer.bias.data, out)
        return out, ctx

    @staticmethod
    def _sync_stats(mean, var, bn, B):
        """mean, var (passes, H): this rank's batch statistics -> the statistics of the global batch (equal B on every
        rank), in place; running statistics updated from them pass by pass like one process with batch world*B would."""
        import torch.distributed as dist
        world = tdist.world_info()[1]
        m64 = mean.double()
        st = torch.stack([m64, var.double() + m64 * m64])  # E[y], E[y^2]
        dist.all_reduce(st, op=dist.ReduceOp.SUM)
        st /= world
        gm = st[0]
        gv = (st[1] - gm * gm).clamp_(min=0.0)
        mean.copy_(gm)
        var.copy_(gv)
        n = float(world * B)
        unb = gv * (n / max(n - 1.0, 1.0))
        for ps in range(mean.shape[0]):
            bn.running_mean.mul_(1.0 - BN_MOMENTUM).add_(gm[ps].float(), alpha=BN_MOMENTUM)
            bn.running_var.mul_(1.0 - BN_MOMENTUM).add_(unb[ps].float(), alpha=BN_MOMENTUM)

    def backward(self, ctx, g, grad_of=None, on_group_done=None, sgd_lr=None, g_antisymmetric=False):
        """g: (passes*B,) = dL/dscore.  grad_of(param) -> tensor to write that dense parameter's gradient into
        (default: fresh tensors).  on_group_done(i): called once the kernels that write the dense gradients of group i
        are enqueued — i = L for the output layer (first), then L-1 ... 0 for the hidden layers (Linear + BatchNorm
        parameters of layer i): the data-parallel trainer starts that group's all-reduce while the layers below are
        still being differentiated.  Returns (grads dict keyed by parameter, d x0).
        sgd_lr (list, one learning rate per hidden Linear layer, or None): plain SGD folded into the weight-gradient
        GEMM — W_l = W_l - lr_l * dW_l written by the GEMM's reduce, dW_l never stored and W_l absent from `grads` (single
        process only: nothing to all-reduce; on the bf16-resident path the input-gradient GEMM reads the W^T image taken in
        the forward pass, on the fp32 path it runs before the weight-gradient GEMM).
        g_antisymmetric: the caller guarantees g[B + t] == -g[t] (the pairwise losses: hinge, BPR) — the output layer's bias
        gradient sum(g) is then the exact +0.0 the two per-pass sums cancel to, and is written as such."""
        net = self.net
        D, M, L, use_bn = self._dims()
        B, passes = ctx["B"], ctx["passes"]
        if use_bn and not ctx["training"]:
            raise RuntimeError("backward through eval-mode BatchNorm is not implemented (the reference trains in "
                               "train mode, model.py:232)")
        grads = {}

        def slot(p):
            t = grad_of(p) if grad_of else torch.empty_like(p.data)
            grads[p] = t
            return t

        xL = ctx["x"][L]  # None: not stored (forward) — the last hidden layer's backward gives the output layer's dW
        ol = net.output_layer
        xw = None
        if xL is None:
            xw = slot(ol.weight).reshape(-1)
        else:
            ops.colsum(xL, slot(ol.weight).reshape(-1), row_weight=g, passes=passes)
        if g_antisymmetric and passes == 2:
            slot(ol.bias).zero_()  # (-S) + S of identically ordered per-pass sums: what the colsum below returns, bit for bit
        else:
            ops.colsum(g.reshape(-1, 1), slot(ol.bias), passes=passes)
        # the output layer's input gradient dx[r][c] = g[r] * w[c]: formed inside the last hidden layer's backward kernels
        # (never stored) when their 4-column form applies; else materialised
        w_out = ol.weight.data.reshape(-1)
        HL = ctx["y"][L - 1].shape[1] if L > 0 else 0
        outer = (g, w_out) if (L > 0 and HL % 4 == 0 and ctx["y"][L - 1].stride(0) % 4 == 0) else None
        assert outer is not None or xL is not None
        dx = None
        if outer is None:
            dx = torch.empty_like(xL)
            ops.outer(g, w_out, dx)
        if on_group_done and xw is None:  # (with xw the output layer's dW is complete after the last hidden layer's reduce)
            on_group_done(L)
        res = ctx.get("resident", False)
        sync = ctx["sync"]  # the forward's decision (ctx), not the live flag
        assert not (sync and xw is not None), "sync-BN keeps x_L: the output layer's dW is a column sum, not outer_xw"
        for l in reversed(range(L)):
            fc = net.fcs[l]
            y = ctx["y"][l]
            if res:
                dy = None
                dy16 = torch.empty(y.shape, dtype=torch.bfloat16, device=y.device)
            else:
                dy, dy16 = torch.empty_like(y), None
            ou = outer if l == L - 1 else None  # (dx is None exactly then)
            if use_bn and sync:  # statistics of the global batch: reduce, all-reduce the 2*passes*H sums, apply
                import torch.distributed as dist
                bn = net.bns[l]
                sums = torch.empty((passes, 2, y.shape[1]), dtype=torch.float32, device=y.device)
                args = (y, dx, B, passes, True, ctx["mean"][l], ctx["var"][l], bn.weight.data, bn.bias.data, BN_EPS, dy,
                        slot(bn.weight), slot(bn.bias))
                ops.bn_relu_backward(*args, dy_colsum=slot(fc.bias), dy16=dy16, phase=1, sums=sums, outer=ou)
                dist.all_reduce(sums, op=dist.ReduceOp.SUM)
                ops.bn_relu_backward(*args, dy_colsum=grads[fc.bias], dy16=dy16, phase=2, sums=sums,
                                     stat_rows=tdist.world_info()[1] * B, outer=ou)
            elif use_bn:
                bn = net.bns[l]
                ops.bn_relu_backward(y, dx, B, passes, True, ctx["mean"][l], ctx["var"][l], bn.weight.data, bn.bias.data,
                                     BN_EPS, dy, slot(bn.weight), slot(bn.bias), dy_colsum=slot(fc.bias), dy16=dy16,
                                     outer=ou, outer_xw=xw if l == L - 1 else None)
                if on_group_done and xw is not None and l == L - 1:
                    on_group_done(L)
            else:
                ops.bn_relu_backward(y, dx, B, passes, False, None, None, None, None, BN_EPS, dy, None, None,
                                     dy_colsum=slot(fc.bias), dy16=dy16, outer=ou)  # db = column sums of dy, same kernel
            if res:
                if sgd_lr is not None:  # W -= lr * dy^T x in the GEMM's reduce
                    self._gemm16(True, dy16, ctx["x"][l], out=fc.weight.data, alpha=-float(sgd_lr[l]), beta=1.0)
                else:
                    self._gemm16(True, dy16, ctx["x"][l], out=slot(fc.weight))   # dW = dy^T x (transposing LDS reads)
                if on_group_done:
                    on_group_done(l)
                # dx = dy W through the W^T image: bf16 between layers (when the layer below keeps a bf16 y), fp32 for the
                # embedding gradient d x0
                dx = self._gemm16(False, dy16, self.w16t[l],
                                  out_bf16=(l > 0 and ctx["y"][l - 1].dtype == torch.bfloat16)
                                  or (l == 0 and self.dx0_bf16))
            else:
                if sgd_lr is None:
                    self._gemm(True, False, dy, ctx["x"][l], out=slot(fc.weight), bf16=net.use_bf16)  # dW = dy^T x (split-K)
                if on_group_done:
                    on_group_done(l)  # before the input-gradient GEMM: the collective overlaps it and the layers below
                # dx = dy W.  Tile-aligned fp32 shapes go NT through a transposed copy of W (tiny: the weights) so that the
                # input-gradient GEMM takes the LDS-DMA kernel like the forward one (csrc/gemm.hip gemm32_nt_glds_kernel)
                H_out, H_in = fc.weight.shape
                if (not net.use_bf16 and dy.shape[0] % 256 == 0 and H_in % 128 == 0 and H_out % 32 == 0
                        and (dy.shape[0] // 256) * (H_in // (256 if H_in % 256 == 0 else 128)) >= 256):
                    dx = self._gemm(False, True, dy, fc.weight.data.t().contiguous())
                else:
                    dx = self._gemm(False, False, dy, fc.weight.data, bf16=net.use_bf16)
                if sgd_lr is not None:  # the weight step AFTER the input gradient read the pre-step weights
                    self._gemm(True, False, dy, ctx["x"][l], out=fc.weight.data, alpha=-float(sgd_lr[l]), beta=1.0,
                               bf16=net.use_bf16)
        return grads, dx


class _MLPScore(torch.autograd.Function):
    """Autograd bridge for callers that drive net.forward / forward_pair + their own loss."""

    @staticmethod
    def forward(ctx, net, ids, passes, *params):
        out, c = net.compute.forward(ids, passes, net.training)
        net._check_err("forward")
        ctx.net, ctx.c = net, c
        return out

    @staticmethod
    def backward(ctx, g):
        net, c = ctx.net, ctx.c
        grads, dx0 = net.compute.backward(c, g.contiguous().float())
        D, M = net.n_factors, net.n_meta_tables()
        ids, B, passes = c["ids"], c["B"], c["passes"]

        def coo(idx, f, p):
            return torch.sparse_coo_tensor(idx.reshape(1, -1).long(), dx0[:, f * D:(f + 1) * D], size=p.shape)

        out = []
        for p in net.all_params():
            if p in grads:
                out.append(grads[p])
            elif p is net.user.weight:
                out.append(coo(torch.cat([ids["user"]] * passes), 0, p))
            elif p is net.item.weight:
                out.append(coo(torch.cat([ids["pos"], ids["neg"]]) if passes == 2 else ids["pos"], 1, p))
            else:
                m = [i for i, q in enumerate(net.metadata_embeddings) if q.weight is p][0]
                idx = torch.cat([ids["pos_meta"][:, m], ids["neg_meta"][:, m]]) if passes == 2 else ids["pos_meta"][:, m]
                out.append(coo(idx, 2 + m, p))
        return (None, None, None, *out)


class MLPTrainer:
    """One fused training step of the MLP scorer (the MLP counterpart of engine.SparseScorerTrainer)."""

    def __init__(self, net, optimizer, batch_capacity):
        self.net, self.opt = net, optimizer
        self.emb_params = net.embedding_params()
        self.dense_params = net.dense_params()
        self.kind = classify_optimizer(optimizer, self.emb_params)  # torch.optim.Adam -> lazy rows + dense Adam
        self.dev = self.emb_params[0].device
        self.bucket = tdist.FlatGradBucket(self.dense_params)
        # one segment of the flat gradient buffer per layer (Linear + BatchNorm parameters), the output layer last: a
        # layer's gradients are all-reduced as soon as its backward kernels are enqueued
        per = 4 if net.use_batch_norm else 2
        L = len(net.fcs)
        self.segs = self.bucket.segments([self.dense_params[l * per:(l + 1) * per] for l in range(L)]
                                         + [self.dense_params[L * per:]])
        self.err = net._err_flag()
        self.row_state = {id(p): RowState(p) for p in self.emb_params} if self.kind in ("sparse_adam", "adagrad") else {}
        self.kernel_events = None
        self.loss_id = 0  # _lib.LOSS_ID: hinge (the reference) | bpr; set by fit(loss=...)
        self.keep_ctx = False  # tests: keep the last step's forward context (y_l, batch statistics) in self.last_ctx
        self.last_ctx = None

    def step(self, ids, loss_slot, auc_slot=None, score_grad=None):
        """One training step.  score_grad (optional, (2B,) fp32: d loss / d score of the positive rows, then of the
        negative rows) replaces the gradient of the built-in pairwise loss — a caller's own loss, and what the full-size
        parity tests drive the backward with; the built-in loss value is still accumulated into loss_slot."""
        net, opt = self.net, self.opt
        B = ids["user"].shape[0]
        D, M = net.n_factors, net.n_meta_tables()
        self.kind = classify_optimizer(opt, self.emb_params)  # every step: param_groups may have changed
        if self.kind in ("sparse_adam", "adagrad") and not self.row_state:
            self.row_state = {id(p): RowState(p) for p in self.emb_params}
        fused_lr = self._fused_embed_lr() if self.kind == "sgd" else None
        net.compute.dx0_bf16 = fused_lr is not None  # (only read on the bf16-resident path)
        scores, ctx = net.compute.forward(ids, 2, True)
        self.last_ctx = ctx if self.keep_ctx else None
        pos, neg = scores[:B], scores[B:]
        gp, gn = ops.hinge_auc_backward(pos, neg, loss_slot, auc_slot, loss=self.loss_id)
        g = gp._base  # (2B,): positive half, negative half
        antisym = score_grad is None  # hinge / BPR: g[B + t] == -g[t]
        if score_grad is not None:
            assert score_grad.shape == (2 * B,) and score_grad.dtype == torch.float32 and score_grad.is_contiguous()
            g = score_grad
        # ---- dense parameters under data parallelism: RCCL all-reduce per layer, started as the backward produces the
        # layer's gradients (it runs on the collective stream beside the remaining backward GEMMs and the embedding-row
        # updates below) and awaited right before the user's optimiser steps the dense parameters
        works = []
        dp = tdist.world_info()[1] > 1
        grads, dx0 = net.compute.backward(ctx, g, grad_of=self.bucket.grad_of,
                                          on_group_done=(lambda i: works.append(
                                              self.bucket.allreduce_segment_async(self.segs[i]))) if dp else None,
                                          sgd_lr=None if dp else self._fused_weight_lrs(), g_antisymmetric=antisym)
        net.compute.dx0_bf16 = False
        tables = []
        if fused_lr is None:  # per-table paths: one index vector per table over the 2B rows of d x0
            idx_user = torch.cat([ids["user"], ids["user"]])
            idx_item = torch.cat([ids["pos"], ids["neg"]])
            tables = [(net.user.weight, idx_user, 0), (net.item.weight, idx_item, 1)]
            for m in range(M):
                tables.append((net.metadata_embeddings[m].weight,
                               torch.cat([ids["pos_meta"][:, m], ids["neg_meta"][:, m]]).contiguous(), 2 + m))
        ld = dx0.stride(0)
        if self.kind == "generic":
            opt.zero_grad()
            for p, idx, f in tables:
                p.grad = torch.sparse_coo_tensor(idx.reshape(1, -1).long(), dx0[:, f * D:(f + 1) * D], size=p.shape)
        else:
            for p in self.emb_params:
                p.grad = None
        for p in self.dense_params:  # (a weight already stepped by its gradient GEMM has no gradient: the optimiser skips it)
            p.grad = self.bucket.grad_of(p) if p in grads else None
        # ---- embedding tables: fused sparse-row updates from the column blocks of d x0 (independent of the dense
        # gradients still being reduced)
        if fused_lr is not None:
            # all tables in two launches: user / item rows (the user's two passes summed in registers; references the
            # epoch's duplicate flags call alone are plain read-modify-writes) + the owner-computes metadata update
            ops.mlp_embed_sgd_update(net.tables(), ctx["Bt"][0], dx0, fused_lr, ids.get("user_dup"), ids.get("item_dup"))
        elif self.kind == "sgd":
            for p, idx, f in tables:
                ops.rows_scatter_add(p.data, idx, dx0[:, f * D:], -_group_of(opt, p)["lr"], ld=ld, err_flag=self.err)
        elif self.kind in ("sparse_adam", "adagrad"):
            for p, idx, f in tables:
                self._rows(p, idx, dx0[:, f * D:], ld)
        self.bucket.finish_segments(works)
        opt.step()

    def _fused_weight_lrs(self):
        """Per hidden Linear layer, the learning rate of the plain SGD step folded into its weight-gradient GEMM
        (MLPCompute.backward's sgd_lr), or None: the optimiser is not momentum-free, decay-free torch.optim.SGD for
        those weights, or TRS_MLP_FUSED_DENSE=0."""
        import os
        if os.environ.get("TRS_MLP_FUSED_DENSE", "1") == "0":
            return None
        ws = [fc.weight for fc in self.net.fcs]
        # classified every step (a few dict look-ups): a caller may change param_groups between steps (momentum, decay)
        if not ws or classify_optimizer(self.opt, ws) != "sgd":
            return None
        return [_group_of(self.opt, w)["lr"] for w in ws]

    def _fused_embed_lr(self):
        """The one learning rate of the fused embedding update, or None when it does not apply (tables in parameter
        groups with different rates, n_factors not a multiple of 4, metadata tables too large for the owner-computes
        kernel, or TRS_MLP_FUSED_EMBED=0)."""
        import os
        if os.environ.get("TRS_MLP_FUSED_EMBED", "1") == "0":
            return None
        lrs = {float(_group_of(self.opt, p)["lr"]) for p in self.emb_params}
        if len(lrs) != 1 or self.net.n_factors % 4 != 0:
            return None
        if getattr(self, "_fused_ok", None) is None:
            self._fused_ok = ops.mlp_embed_sgd_supported(self.net.tables())
        return lrs.pop() if self._fused_ok else None

    def _rows(self, p, idx, vals, ld):
        apply_rows(self.kind, self.opt, p, self.row_state[id(p)], idx, vals, ld)

    def check_errors(self):
        self.net._check_err("fit")
