# -*- coding: utf-8 -*-
"""bench.py — throughput of the north-star hot path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the training hot path over one batch of synthetic interactions already resident in HBM:
epoch-shuffle slice + dynamic negative sampling -> fused embedding gather + FM pairwise scoring + hinge + backward ->
sparse embedding-row SGD update.  Default workload at EVERY N (BASELINE.json's metric is quoted on it: configs[3],
SURVEY 8d "c4"): net_type='fm', 10M users x 1M items, dim=128, dynamic_neg_sampling=True, per-GPU batch 32 768 (global
batch 262 144 at 8 GPUs), every rank an independent 125M-interaction shard of the 1B stream (weak scaling),
torch.optim.SGD(lr=1e-2), fp32.  --config c2 | c1 | c3 | c5 select BASELINE.json's other configurations.
metric = training interactions/s (pos+neg) = 2 x triples/s.

What the timed region contains: exactly K steps through FitRunner.run_steps — the object fit() is built on — incl. the
per-slice presort (sparse regime, c4: trs_epoch_flags = ids + duplicate flags; dense regime, c2: trs_epoch_presort =
ids + item references grouped by row, + user flags; prefetched on a side stream; slices are sized to the run so a
short run carries its proportional share).  NO kernel events are recorded in it.  `host_enqueue_ms` = the host time
that went into enqueueing the K steps (the launches run ahead of the GPU; the rest of the window is the GPU finishing).
After it, untimed: (1) an instrumented window of the same steps with HIP events around K1 / K2 on the launch stream
(>= 8 samples whatever --steps is) -> `roofline` of the dominant kernel; (2) the north-star pass alone
(trs_score_forward = fused pos+neg gather + FM pairwise score) at the per-GPU batch and at the global batch 262 144
-> `roofline_pass`; (3) rank 0, N=1: the CPU port of the reference's loop on a bounded sample -> `cpu_baseline`.

Multi-GPU: one process per GPU; tables replicated; FM/Linear have no dense parameters, so a step has no
data-path collective.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md "Chip-level parameters")

MFMA_PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0}  # dense matrix peaks (MI355X_MICROARCH.md), no sparsity

CONFIGS = {
    # name: net, n_users, n_items, interactions generated per rank, D, per-GPU batch, [metadata category counts],
    #       [hidden layers], bf16 GEMM inputs, description
    "c2": dict(net="fm", n_users=1_000_000, n_items=100_000, n=100_000_000, D=64, B=65_536, meta=[], hidden=None,
               amp=False,
               desc="c2: net_type='fm', 1M users x 100K items x 100M interactions, dim=64, dynamic_neg_sampling=True, "
                    "batch 65536, SGD(lr=1e-2), fp32"),
    "c1": dict(net="linear", n_users=3_000, n_items=1_000, n=100_000, D=32, B=1_024, meta=[], hidden=None, amp=False,
               desc="c1: net_type='linear', 3000 users x 1000 items x 100000 interactions, dim=32, batch 1024, "
                    "static negatives"),
    "c4": dict(net="fm", n_users=10_000_000, n_items=1_000_000, n=125_000_000, D=128, B=32_768, meta=[], hidden=None,
               amp=False,
               desc="c4 per-GPU shard: net_type='fm', 10M users x 1M items, 125M-interaction shard of 1B, dim=128, "
                    "per-GPU batch 32768 (global 262144 at 8 GPUs)"),
    "c3": dict(net="mlp", n_users=1_000_000, n_items=100_000, n=100_000_000, D=128, B=65_536, meta=[10_000],
               hidden=[512, 256, 128], amp=False,
               desc="c3: net_type='mlp' hidden [512,256,128] + BatchNorm, 1M users x 100K items + 1 metadata column "
                    "(10K categories) x 100M interactions, dim=128, dynamic_neg_sampling=True, batch 65536, "
                    "SGD(lr=1e-2), fp32"),
    "c5": dict(net="mlp", n_users=10_000_000, n_items=1_000_000, n=125_000_000, D=256, B=32_768,
               meta=[10_000, 10_000, 10_000], hidden=[1024, 512, 256], amp=True,
               desc="c5 per-GPU shard: net_type='mlp' hidden [1024,512,256] + BatchNorm, 10M users x 1M items + 3 "
                    "metadata columns (10K categories each), 125M-interaction shard of 1B, dim=256, bf16 GEMM inputs / "
                    "fp32 accumulate, per-GPU batch 32768 (global 262144 at 8 GPUs), SGD(lr=1e-2)"),
}


def synth_stream(n_users, n_items, n, device, seed, rank=0, world=1):
    """Uniform synthetic interactions with guaranteed dense id coverage, generated in HBM.  world > 1: this
    rank's shard of a stream partitioned by user (fit()'s dp_partition='user'): the users u with u % world == rank, each
    of them present, uniform otherwise; items uniform over the whole catalogue."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    own = torch.arange(rank, n_users, world, device=device, dtype=torch.int32)
    n_own = own.numel()
    users = torch.cat([own, torch.randint(0, n_own, (n - n_own,), device=device, dtype=torch.int32, generator=g)
                       * world + rank])
    reps = -(-n_own // n_items)
    items = torch.cat([torch.arange(n_items, device=device, dtype=torch.int32).repeat(reps)[:n_own],
                       torch.randint(0, n_items, (n - n_own,), device=device, dtype=torch.int32, generator=g)])
    perm = torch.randperm(n, device=device, generator=g)
    return users[perm].contiguous(), items[perm].contiguous()


def slice_fractions(runner):
    """Shares of the row references of the presort slice in use that are alone on their row inside their batch (K1
    updates them in place) — what the exact per-kernel byte counts below depend on.  None off the presorted path."""
    tr = runner.trainer
    ps = tr._ps_sets[tr._ps_cur] if getattr(tr, "_ps_sets", None) else None
    if ps is None:
        return None
    n = ps.n_batches * ps.batch
    return {"k1_takes_lone_items": ps.key_bytes == 0,  # EpochFlags (accumulate mode); the sorted runs take every item reference
            "one_launch": bool(getattr(tr, "one_launch", False)),  # flag mode with the flagged references applied by K1's launch
            "users_alone": 1.0 - float(ps.user_dup[:n].float().mean()),
            # (the sorted-run presort computes item flags only when K1 reads them: None = not computed)
            "item_refs_alone": (1.0 - float(ps.item_dup[:n].float().mean())) if getattr(ps, "item_flags", True) else None,
            "note": "shares of the references whose row no other reference of the batch names (flags of the presort)"}


def kernel_algorithmic_bytes(R, row, state_rows, seen, frac):
    """Algorithmic bytes per triple of each step kernel (DESIGN.md "Kernels"): every row a kernel must read counted
    once, every row it must write counted once, ids and per-triple scalars; no staging traffic is credited."""
    inline_user = "sorted_updates_fused_kernel" in seen or "sorted_item_update_kernel" in seen
    ua = frac["users_alone"] if frac else 1.0       # K1 writes these user rows itself
    ia = frac["item_refs_alone"] if (frac and frac.get("k1_takes_lone_items")) else 0.0  # ... and these item rows
    k1 = 16 + R * row + 8 + ((ua + 2 * ia) * (1 + 2 * state_rows) * row if inline_user else 0)
    # K2 (sorted runs): reads ids/coefficients, reads + writes the item rows K1 left (at most one row per reference:
    # an upper bound of the distinct rows) and the user rows of duplicated users
    k2 = 12 + ((2 * (1 - ia)) * 2 + (1 - ua) * 2) * (1 + state_rows) * row
    acc_apply = 0.0
    if "flagged_update_kernel" in seen or (frac and frac.get("one_launch")):
        # flag mode (sparse regime): K1 reads the triple's R rows and writes the rows of the references that are alone in
        # the batch; the second launch reads + writes (atomically) one table row per flagged reference.  Staging (gz,
        # the old user row / gradient row of flagged references) is not credited.
        k1 = 16 + R * row + (ua + 2 * ia) * row
        acc_apply = 15 + ((1 - ua) + 2 * (1 - ia)) * 2 * row
        if frac and frac.get("one_launch"):  # ... or K1's own workgroups do, after their grid-wide wait: the whole step.
            # Every row read once and written once = SURVEY 8d's step figure; the flagged rows' second access (the atomic
            # read-modify-write at the memory side) is not credited
            k1 = 16 + R * row + R * row
            acc_apply = 0.0
    return {"score_kernel<fwd_bwd>": 16 + R * row + 8, "score_sgd_update_kernel": 12 + R * row,
            "fwd_stage_kernel": k1, "item_update_kernel": 12 + 3 * row, "sorted_item_update_kernel": 12 + 3 * row,
            "sorted_updates_fused_kernel": k2, "user_update_kernel": 4 + row, "sorted_user_dup_update_kernel": 4 + row,
            "flagged_update_kernel": acc_apply}


def time_pass(model, runner, B, D, R, dev):
    """The north-star pass by itself at this config's tables: trs_score_forward (pair_scores_kernel: fused positive +
    negative embedding gather + FM / Linear pairwise score, read-only).  Every launch scores DIFFERENT triples (fresh
    rows: nothing is served from a previous launch's cache footprint); 8 groups of 10 back-to-back launches, one HIP
    event pair per group on the launch stream, at the per-GPU batch and at BASELINE's global batch 262 144."""
    st = runner.st
    n_items = model.n_items
    per = 16 + R * (4 * D + 4) + 8  # ids, R rows + 1-wide terms, two scores out
    res = {"kernel": "pair_scores_kernel", "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "algorithmic_bytes_per_triple": per, "shapes": {}}
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    # (triples per launch, label, event groups, launches per group): the per-GPU batch as ONE launch, BASELINE's global
    # batch, and the per-GPU batch the way evaluate() issues the pass — up to 64 batches per launch (model.py::evaluate)
    ev_group = max(1, min(64, (1 << 22) // B))
    shapes = [(B, f"B={B}", 8, 10), (262_144, "B=262144", 8, 10)]
    if ev_group > 1 and B * ev_group != 262_144:
        shapes.append((B * ev_group, f"B={B} x {ev_group} batches per launch (evaluate)", 4, 3))
    seen_bp = set()
    for Bp, label, groups, per_group in shapes:
        need = (groups * per_group + 3) * Bp
        if need > st["user"].shape[0] or Bp in seen_bp:
            continue
        seen_bp.add(Bp)
        neg = torch.randint(0, n_items, (need,), device=dev, dtype=torch.int32, generator=g)
        import ctypes as C
        from torchrecsys_amd import _lib, ops
        lib, T = _lib.load(), model.net.tables()
        err = torch.zeros(1, dtype=torch.int32, device=dev)
        pos_s, neg_s = torch.empty(Bp, device=dev), torch.empty(Bp, device=dev)
        bts = [ops.make_batch(st["user"][k * Bp:(k + 1) * Bp], st["pos"][k * Bp:(k + 1) * Bp],
                              neg[k * Bp:(k + 1) * Bp], None, None, err) for k in range(groups * per_group + 3)]
        net_id = ops.NET_ID[model.net.NET]

        def launch(k):  # one C-ABI call = one kernel launch on torch's current stream; no allocation, no sync
            _lib.check(lib.trs_score_forward(net_id, C.byref(T), C.byref(bts[k][0]), ops.ptr(pos_s), ops.ptr(neg_s),
                                             ops._stream()), "trs_score_forward")
        for k in range(3):
            launch(k)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(groups)]
        k = 3
        for e0, e1 in evs:
            e0.record()
            for _ in range(per_group):
                launch(k)
                k += 1
            e1.record()
        torch.cuda.synchronize()
        us = [1e3 * e0.elapsed_time(e1) / per_group for e0, e1 in evs]
        mean_us = sum(us) / len(us)
        ach = per * Bp / (mean_us * 1e-6) / 1e9
        res["shapes"][label] = {"mean_launch_us": mean_us, "triples_per_launch": Bp, "min_group_us": min(us), "max_group_us": max(us),
                                    "achieved": ach, "frac": ach / HBM_PEAK_GBS, "samples": groups,
                                    "launches": groups * per_group}
        assert int(err.item()) == 0, "an id outside its table in the pass measurement"
        del neg, bts
    if f"B={B}" in res["shapes"]:
        res["achieved"] = res["shapes"][f"B={B}"]["achieved"]
        res["frac"] = res["shapes"][f"B={B}"]["frac"]
    res["note"] = "launch-to-launch intervals of back-to-back C-ABI calls (launch gaps and 1/10 event record included)"
    return res


def build_model(config, n_inter, dev, rank=0, world=1):
    """TorchRecSys over full-size tables of BASELINE configuration `config` with `n_inter` synthetic interactions of this
    rank's shard resident in HBM."""
    import contextlib
    import io
    from torchrecsys_amd.model import TorchRecSys
    cfg = CONFIGS[config]
    net, n_users, n_items, D = cfg["net"], cfg["n_users"], cfg["n_items"], cfg["D"]
    users, items = synth_stream(n_users, n_items, n_inter, dev, seed=1000 + rank, rank=rank, world=world)
    meta = None
    if cfg["meta"]:  # one categorical id per item and column, every category present
        g = torch.Generator(device=dev)
        g.manual_seed(5)
        cols = []
        for nc in cfg["meta"]:
            c = torch.randint(0, nc, (n_items,), device=dev, dtype=torch.int32, generator=g)
            c[:nc] = torch.arange(nc, device=dev, dtype=torch.int32)
            cols.append(c)
        meta = torch.stack(cols, dim=1).contiguous()
    with contextlib.redirect_stdout(io.StringIO()):
        torch.manual_seed(7)
        kw = dict(hidden_layers=cfg["hidden"]) if net == "mlp" else {}
        model = TorchRecSys.from_tensors(users, items, n_users=n_users, n_items=n_items, item_metadata=meta,
                                         n_factors=D, net_type=net, split_ratio=0.8, dynamic_neg_sampling=config != "c1",
                                         use_amp=cfg["amp"], rng="device", seed=7 + rank, pre_sharded=True,
                                         dp_partition="user", **kw)
        # pre_sharded: every rank generated its own interaction shard, cut by user (each user row has one writer)
    return model


def mlp_gemm_roofline(cfg, gemm_events, timed_steps, ms_per_step):
    """`roofline` of an MLP configuration: the GEMMs are the dominant kernels, bound by the matrix cores."""
    M = len(cfg["meta"])
    dims = [(2 + M) * cfg["D"]] + list(cfg["hidden"])
    P = sum(a_ * b_ for a_, b_ in zip(dims[:-1], dims[1:])) + dims[-1]  # MACs per sample
    ms = sum(e0.elapsed_time(e1) for e0, e1, _ in gemm_events)
    fl = sum(f for _, _, f in gemm_events)
    ach = fl / (ms * 1e-3) / 1e12
    dtype = "bf16" if cfg["amp"] else "f32"
    peak = MFMA_PEAK_TFLOPS[dtype]
    return {"bound": "mfma", "kernel": ("gemm16_nt_glds_kernel (forward / input-gradient GEMMs) + gemm_bf16in_kernel (weight-gradient GEMMs)"
                                        if cfg["amp"] else "gemm32_nt_glds_kernel / gemm_f32_kernel"),
            "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": None,
            "gemm_launches_per_step": len(gemm_events) / timed_steps,
            "gemm_ms_per_step": ms / timed_steps,
            "gemm_share_of_step": (ms / timed_steps) / ms_per_step,
            "timed_steps": timed_steps,
            "algorithmic_flops_per_triple": 12 * P,
            "whole_step_TFLOPs": 12.0 * P * cfg["B"] / (ms_per_step * 1e-3) / 1e12,
            "note": "event intervals around the GEMM launches (HIP events on the launch stream) in an "
                    "instrumented window right after the timed region; they include one event record each"}


def mlp_leg(config, dev, steps=24, warmup=6, n_extra=6_000_000):
    """MFMA evidence beside the HBM-bound default line (north_star: "MFMA utilisation (MLP GEMM) against gfx950 peak"): a
    short window of BASELINE's MLP configuration `config` on full-size tables — `steps` timed steps through
    FitRunner.run_steps (same bracket as the main line: synchronise on both sides), then an instrumented window with HIP
    events around the GEMM launches.  The interaction shard is short (the step's cost does not depend on its length)."""
    cfg = CONFIGS[config]
    model = build_model(config, cfg["n_users"] + n_extra, dev)  # (every user once + n_extra uniform interactions)
    opt = torch.optim.SGD(model.parameters(), lr=1e-2)
    runner = model.make_runner(opt, cfg["B"])
    model.net.train()
    full = runner.n_train // cfg["B"]
    assert full >= warmup + steps + 16, "interaction shard too short for the MLP window"
    runner.begin_epoch()
    runner.run_steps(warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    runner.run_steps(steps)
    torch.cuda.synchronize()
    ms_per_step = 1e3 * (time.perf_counter() - t0) / steps
    comp = model.net.compute
    comp.gemm_events, comp.gemm_steps_seen, comp.gemm_steps_timed = [], 0, 0
    runner.run_steps(15)  # one step in 5 carries events
    torch.cuda.synchronize()
    ev, timed = comp.gemm_events, max(comp.gemm_steps_timed, 1)
    comp.gemm_events = None
    runner.end_epoch()
    r = mlp_gemm_roofline(cfg, ev, timed, ms_per_step)
    peak = MFMA_PEAK_TFLOPS["bf16" if cfg["amp"] else "f32"]
    return {"config": cfg["desc"], "dtype": "bf16" if cfg["amp"] else "f32", "steps": steps, "warmup": warmup,
            "ms_per_step": ms_per_step, "interactions_per_s": 2.0 * cfg["B"] / (ms_per_step * 1e-3),
            "gemm_TFLOPs": r["achieved"], "peak_TFLOPs": peak, "gemm_frac": r["frac"],
            "whole_step_TFLOPs": r["whole_step_TFLOPs"], "whole_step_frac": r["whole_step_TFLOPs"] / peak,
            "gemm_ms_per_step": r["gemm_ms_per_step"], "gemm_launches_per_step": r["gemm_launches_per_step"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1024)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--config", default="c4", choices=sorted(CONFIGS),
                    help="c4 (default at every N): the workload BASELINE.json's metric and scaling target are quoted on")
    ap.add_argument("--optimizer", default="sgd", choices=["sgd", "sparse_adam", "adagrad", "adam"],
                    help="sgd = BASELINE.json's primary optimiser; sparse_adam (lazy Adam) / adagrad = secondary; adam = "
                         "torch.optim.Adam as in the reference's README (lazy rows on the tables, dense Adam on the MLP)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true", help="skip the instrumented window (no roofline)")
    ap.add_argument("--no-pass", action="store_true", help="skip the north-star pass measurement (no roofline_pass)")
    ap.add_argument("--no-mlp", action="store_true",
                    help="skip the MLP legs (roofline_mlp: short c5-shard and c3 windows after the default c4 line)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (there is no CPU fallback)"
    # (rehearsal on a one-GPU box: TRS_BENCH_SHARE_DEVICE=1 puts every rank on the devices that exist and
    # TRS_DIST_BACKEND=gloo moves the few collectives through the host; a real node runs one rank per GPU over RCCL)
    if os.environ.get("TRS_BENCH_SHARE_DEVICE") == "1":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("TRS_DIST_BACKEND", "nccl")  # "nccl" IS RCCL over xGMI on ROCm
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))

    from torchrecsys_amd import _lib
    from torchrecsys_amd.model import TorchRecSys
    _lib.check(_lib.load().trs_check_device(), "trs_check_device")

    cfg = CONFIGS[args.config]
    net, n_users, n_items, n_inter, D, B, desc = (cfg[k] for k in ("net", "n_users", "n_items", "n", "D", "B", "desc"))
    dynamic = args.config != "c1"
    model = build_model(args.config, n_inter, dev, rank, world)
    if args.optimizer == "sgd":
        opt = torch.optim.SGD(model.parameters(), lr=1e-2)
    elif args.optimizer == "sparse_adam":
        assert net != "mlp", "SparseAdam takes no dense parameters: use torch.optim.Adam for the MLP"
        opt = torch.optim.SparseAdam(list(model.parameters()), lr=1e-3)
    elif args.optimizer == "adam":
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    else:
        opt = torch.optim.Adagrad(model.parameters(), lr=1e-2)
    if args.optimizer != "sgd":
        desc = desc.replace("SGD(lr=1e-2)", {"sparse_adam": "SparseAdam(lr=1e-3)", "adagrad": "Adagrad(lr=1e-2)",
                                             "adam": "Adam(lr=1e-3)"}[args.optimizer])
    # Presort slices sized to the run: a slice's grouping work is prefetched on a side stream while the previous slice's
    # steps run, so a run much shorter than the production slice (512 batches) would otherwise be charged a whole
    # slice's sort; with slices no longer than the run the timed steps carry their proportional share.
    is_mlp = net == "mlp"
    if not is_mlp and "TRS_SLICE_BATCHES" not in os.environ:
        from torchrecsys_amd.engine import SparseScorerTrainer
        sl = 8
        while sl * 2 <= min(512, max(args.steps, 8)):
            sl *= 2
        SparseScorerTrainer.SLICE_BATCHES = sl
    runner = model.make_runner(opt, B)
    model.net.train()

    full = runner.n_train // B  # only full batches are timed; an epoch's partial last batch is skipped
    assert full > 0, "stream shorter than one batch"
    state = {"started": False}

    def run(k):
        """k steps, rolling into the next epoch (new shuffle key, loss read-back) when the current one ends."""
        done = 0
        while done < k:
            if not state["started"] or runner.next_batch >= full:
                if state["started"]:
                    runner.end_epoch()
                runner.begin_epoch()
                state["started"] = True
            done += runner.run_steps(min(k - done, full - runner.next_batch))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    timeline = None
    if os.environ.get("TRS_BENCH_TIMELINE") == "1":  # diagnostic: host time stamps of every C call of the timed region
        from torchrecsys_amd import ops as _ops
        timeline = _ops.TIMELINE = []
        _orig_call = _ops.FlagStepCall.__call__

        def _stamped(self_, *a_, **k_):
            timeline.append(("call", time.perf_counter()))
            r_ = _orig_call(self_, *a_, **k_)
            timeline.append(("ret", time.perf_counter()))
            return r_
        _ops.FlagStepCall.__call__ = _stamped
        _orig_rs, _orig_ps = type(runner).run_steps, type(runner.trainer).presort_slice

        def _rs(self_, *a_, **k_):
            timeline.append(("run_steps", time.perf_counter()))
            return _orig_rs(self_, *a_, **k_)

        def _ps(self_, *a_, **k_):
            timeline.append(("presort_slice", time.perf_counter()))
            r_ = _orig_ps(self_, *a_, **k_)
            timeline.append(("presort_ret", time.perf_counter()))
            return r_
        type(runner).run_steps, type(runner.trainer).presort_slice = _rs, _ps
    run(args.warmup)
    barrier()
    # Between the synchronise and the clock: the host side of the next run_steps() call walked once with ZERO steps (no
    # kernel is launched, nothing is skipped later) — the thread wakes from the wait with cold caches, and its first pass
    # through that Python code takes ~75 us instead of ~12, which a 20-step window would be charged as 9 % of its time
    # (FitRunner.touch_host_path; TRS_BENCH_TOUCH=0 switches it off).  The K timed steps below are complete steps.
    if os.environ.get("TRS_BENCH_TOUCH", "1") != "0" and hasattr(runner, "touch_host_path") and state["started"]:
        runner.touch_host_path()
    if timeline is not None:
        del timeline[:]
    t0 = time.perf_counter()
    run(args.steps)
    enqueued = time.perf_counter() - t0  # host time to enqueue the K steps (the GPU is still running them)
    barrier()
    elapsed = time.perf_counter() - t0
    if timeline is not None:
        _ops.FlagStepCall.__call__ = _orig_call
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- instrumented window (untimed): the same steps with HIP events around the kernels, on the launch stream ----
    events, gemm_events = {}, None
    inst_slice = None
    if not args.no_kernel_events:
        n_inst = 64 if is_mlp else 256
        if not is_mlp and "TRS_SLICE_BATCHES" not in os.environ:
            # The kernel-level measurements describe the step kernel as a fit() runs it: with the production presort
            # slices (512 batches).  The timed region above keeps slices no longer than the run (so that it carries its
            # share of the presort), and a 16-batch slice puts the presort's fixed latency beside a quarter of the
            # steps — a property of the short window, not of the kernel (profiles/EXPERIMENTS.md, slice length).
            from torchrecsys_amd.engine import SparseScorerTrainer
            inst_slice = SparseScorerTrainer.SLICE_BATCHES = 512
            run(16)  # enter the first long slice (its presort runs here) before anything is recorded
            torch.cuda.synchronize()
        if is_mlp:
            model.net.compute.gemm_events = []
            model.net.compute.gemm_steps_seen = model.net.compute.gemm_steps_timed = 0
        else:
            runner.trainer.kernel_events = {}
            runner.trainer.EVENT_EVERY = 7  # 36 sampled steps of 256 (a prime stride: never in phase with the C calls)
        run(n_inst)
        torch.cuda.synchronize()
        if is_mlp:
            gemm_events = model.net.compute.gemm_events
            model.net.compute.gemm_events = None
        else:
            events = runner.trainer.kernel_events or {}
            runner.trainer.kernel_events = None
    # One-launch steps: the launch stream holds NOTHING but the step kernel, so the interval between two events that are G
    # launches apart is G kernel durations (launch gaps included) and only 1/G of an event record — the per-step intervals
    # above contain a whole record each (~5 us on a ~32 us kernel; rocprofv3's durations in profiles/ say which is right).
    grouped_us = None
    if events and getattr(runner.trainer, "one_launch", False):
        groups, per_group = 16, 8
        run(per_group)  # (not synchronised: the queue is never empty when a group's first event is recorded)
        gev = []
        for _ in range(groups):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run(per_group)
            e1.record()
            gev.append((e0, e1))
        torch.cuda.synchronize()
        grouped_us = [1e3 * e0.elapsed_time(e1) / per_group for e0, e1 in gev]
    frac = slice_fractions(runner) if not is_mlp else None
    runner.end_epoch()  # also raises if any id was out of range

    triples = args.steps * B * world
    value = 2.0 * triples / elapsed
    M = len(cfg["meta"])
    R = 3 + 2 * M
    row = 4 * D + 4  # one embedding row + its 1-wide term
    step_bytes = 16 + 2 * R * row  # FM/Linear fused SGD step, rows read once + written once
    state_rows = {"sgd": 0, "sparse_adam": 2, "adagrad": 1, "adam": 2}[args.optimizer]  # state tables read + written per row
    step_bytes += 2 * state_rows * R * row
    dtype = "bf16" if cfg["amp"] else "f32"
    out = {
        "metric": "training interactions/sec (pos+neg)", "value": value, "unit": "interactions/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
        "config": {"workload": desc, "global_batch": B * world, "per_gpu_batch": B,
                   "parallelism": (f"dp{world}: interaction stream sharded, tables replicated, one flat all-reduce of the "
                                   f"dense gradients per step" if is_mlp else
                                   f"dp{world}: interaction stream sharded by user_id % {world}, tables replicated, no "
                                   f"per-step collective (FM/Linear have no dense parameters); item tables averaged per "
                                   f"epoch, user rows gathered from their owners after fit() — both OUTSIDE the timed "
                                   f"steps, measured in `replica_sync`"),
                   "rng": "device (Feistel epoch shuffle + Philox4x32-10 negative sampler)"},
    }
    out["host_enqueue_ms"] = 1e3 * enqueued  # of the timed region's `1e3 * elapsed` ms: the launches run ahead of the GPU
    if timeline is not None:
        out["host_timeline_us"] = [(k_, round(1e6 * (t_ - t0), 1)) for k_, t_ in timeline] + [("synced", round(1e6 * elapsed, 1))]
    if not is_mlp:
        out["step_algorithmic_GBps_per_gpu"] = step_bytes * B * args.steps / elapsed / 1e9
        out["step_frac_of_hbm_peak"] = out["step_algorithmic_GBps_per_gpu"] / HBM_PEAK_GBS
        out["step_algorithmic_bytes_per_triple"] = step_bytes
    # ---- MLP: the GEMMs are the dominant kernels, bound by the matrix cores ----
    if gemm_events:
        out["roofline"] = mlp_gemm_roofline(cfg, gemm_events, max(model.net.compute.gemm_steps_timed, 1),
                                            1e3 * elapsed / args.steps)
    # ---- roofline of the dominant kernel: algorithmic bytes per launch / mean launch duration (HIP events) ----
    if events:
        def _ms(rec):  # (TimingEvents, i, j) from the C step loop, or (torch start, torch end) from the generic path
            return rec[0].elapsed_ms(rec[1], rec[2]) if len(rec) == 3 else rec[0].elapsed_time(rec[1])
        n_samples = {k: len(v) for k, v in events.items()}
        raw_ms = {k: sum(_ms(r) for r in v) / len(v) for k, v in events.items()}
        med_ms = {k: sorted(_ms(r) for r in v)[len(v) // 2] for k, v in events.items()}
        # An interval between two hipEventRecords contains the second record's own cost (a barrier packet + timestamp
        # write).  On the presorted path the last two events of a step are recorded back to back, so an upper bound of
        # that cost is measured live.  `achieved` is computed from the RAW intervals (conservative: rocprofv3's kernel
        # durations in profiles/ are shorter); the intervals minus the measured record cost are reported beside them.
        # One-launch steps: from the grouped intervals measured above (G launches per event pair) instead.
        ev_ms = raw_ms.pop("event_overhead", 0.0)
        n_samples.pop("event_overhead", None)
        med_ms.pop("event_overhead", None)
        for d_ in (raw_ms, n_samples, med_ms):  # (one-launch flag mode: the second interval is an event record too)
            d_.pop("event_overhead_2", None)
        mean_ms = dict(raw_ms)
        # dominant = the kernel with the longest MEDIAN interval (a sample that overlaps a presort burst on the side
        # stream can be 20x a normal one and would decide a mean over 37 samples); `achieved` uses that kernel's MEAN
        dom = max(med_ms, key=med_ms.get)
        per_triple = kernel_algorithmic_bytes(R, row, state_rows, mean_ms, frac)
        dur_ms = mean_ms[dom]
        if grouped_us and dom == "fwd_stage_kernel" and frac and frac.get("one_launch"):
            dur_ms = 1e-3 * sum(grouped_us) / len(grouped_us)
        ach = per_triple[dom] * B / (dur_ms * 1e-3) / 1e9
        # measured HBM bytes per launch of that kernel: rocprofv3 PMC passes of THIS command committed under profiles/
        # (FETCH_SIZE / WRITE_SIZE in separate passes, corrected as MI355X_MICROARCH.md prescribes; tools/pmc_traffic.py).
        # A constant read from that file, not a live measurement: named in traffic_source; null when no profile of the
        # running config exists.
        traffic, traffic_source = None, None
        pmc_path = next((q for q in (os.path.join(ROOT, "profiles", f"{r_}_pmc_traffic_{args.config}.json") for r_ in ("r03", "r02")) if os.path.exists(q)), "")
        if world == 1 and args.optimizer == "sgd" and os.path.exists(pmc_path):
            pk = json.load(open(pmc_path))["kernels"]
            parts = {"item_update_kernel": ("item_owner_update_kernel", "item_update_kernel")}.get(dom, (dom,))
            if all(q in pk for q in parts):
                traffic = sum(pk[q]["traffic_bytes_per_launch"] for q in parts)
                traffic_source = f"profiles/{os.path.basename(pmc_path)} (rocprofv3 --pmc, not measured in this run)"
        out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                           "samples": n_samples[dom],
                           "launch_us": 1e3 * dur_ms,  # what `achieved` divides by
                           "launch_us_source": ("16 groups of 8 consecutive one-launch steps, one HIP event pair per group "
                                                "(launch gaps and 1/8 event record per launch included)"
                                                if dur_ms != mean_ms[dom] else "mean of the per-step event intervals"),
                           "grouped_launch_us": ({"mean": sum(grouped_us) / len(grouped_us), "min": min(grouped_us),
                                                  "max": max(grouped_us)} if grouped_us else None),
                           "mean_launch_us": {k: 1e3 * v for k, v in mean_ms.items()},
                           "median_launch_us": {k: 1e3 * v for k, v in med_ms.items()},
                           "mean_launch_us_minus_event_record": {k: 1e3 * max(v - ev_ms, 0.0) for k, v in mean_ms.items()},
                           "event_record_overhead_us": 1e3 * ev_ms,
                           "algorithmic_bytes_per_triple": per_triple[dom],
                           "algorithmic_bytes_per_triple_all": {k: per_triple[k] for k in mean_ms if k in per_triple},
                           "row_reference_fractions": frac,
                           "presort_slice_batches": {"timed_region": sl if not is_mlp and "TRS_SLICE_BATCHES" not in os.environ else None,
                                                     "instrumented_window": inst_slice},
                           "note": "HIP events on the launch stream, one step in 7 of a 256-step instrumented window "
                                   "that follows the timed region (the timed region itself records no events); the "
                                   "instrumented window runs with the production presort slices (512 batches)"}
    # ---- the north-star pass alone: fused pos+neg gather + pairwise score (trs_score_forward), read-only ----
    if not is_mlp and not args.no_pass and M == 0:
        out["roofline_pass"] = time_pass(model, runner, B, D, R, dev)
    # ---- data parallel: what fit() adds around the steps (not in the timed region): the per-epoch average of the tables
    # with several writers and the end-of-fit all-gather of the owners' user rows
    if world > 1 and not is_mlp:
        def timed(fn):
            barrier()
            t1 = time.perf_counter()
            fn()
            barrier()
            return time.perf_counter() - t1
        from torchrecsys_amd import dist as tdist
        model._sync_replicas()  # warm-up of the communicator
        t_avg = timed(model._sync_replicas)
        t_gather = timed(lambda: [tdist.gather_owned_rows_(t_.data) for t_ in model._user_tables()])
        shared = [p_ for p_ in model.net.table_params() if all(p_ is not u_ for u_ in model._user_tables())]
        out["replica_sync"] = {
            "per_epoch_average_ms": 1e3 * t_avg, "averaged_bytes": sum(4 * p_.numel() for p_ in shared),
            "steps_per_epoch": full, "amortised_us_per_step": 1e6 * t_avg / full,
            "end_of_fit_user_gather_ms": 1e3 * t_gather,
            "user_table_bytes": sum(4 * t_.numel() for t_ in model._user_tables())}
    # ---- CPU baseline: the op-sequence port of the reference's fit() loop on this box's host cores ----
    # (torch.optim.Adam rejects the sparse gradients of the CPU port's nn.Embedding(sparse=True): no CPU leg for it)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.optimizer != "adam":
        from oracle import cpu_fit
        cores = min(16, len(os.sched_getaffinity(0)))  # the one-GPU box share (oversubscribing collapses torch CPU ops)
        n_rows = min(n_inter, 4_000_000)
        res = cpu_fit.time_steps(net, n_users, n_items, D, B, n_rows=n_rows, steps=100, warmup=1, dynamic=dynamic,
                                 threads=cores, max_seconds=20.0, meta_cats=cfg["meta"], hidden=cfg["hidden"],
                                 optimizer=args.optimizer)
        out["cpu_baseline"] = {"value": res["interactions_per_s"], "unit": "interactions/s", "cores": res["threads"],
                               "kind": "port",
                               "sample": f"{res['steps']} steps of batch {B} ({res['seconds']:.1f} s) on full-size "
                                         f"tables, {n_rows} synthetic interactions, oracle/cpu_fit.py"}
    # ---- MFMA evidence in the same line (N = 1, default workload only): short windows of the two MLP configurations ----
    if rank == 0 and world == 1 and args.config == "c4" and args.optimizer == "sgd" and not args.no_mlp:
        del runner, opt, model
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        out["roofline_mlp"] = {}
        for name in ("c5", "c3"):
            out["roofline_mlp"][name] = mlp_leg(name, dev)
            gc.collect()
            torch.cuda.empty_cache()
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    from torchrecsys_amd.helper.cuda import host_threads
    with host_threads():  # torch's CPU pool capped at the container's CPU budget (N ranks share one host)
        main()
