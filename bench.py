# -*- coding: utf-8 -*-
"""bench.py — throughput of the north-star hot path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the training hot path over one batch of synthetic interactions already resident in HBM:
epoch-shuffle slice + dynamic negative sampling -> fused embedding gather + FM pairwise scoring + hinge + backward ->
sparse embedding-row SGD update.  The item references and user ids of every 512 batches are grouped by row once
(trs_epoch_presort / trs_epoch_user_dups, on a side stream); that work runs INSIDE the timed region at its true rate
(default 1024 timed steps = 2 slices).  Per-kernel HIP events are recorded on one step in 29 (a sampled step runs about
20 us longer); --no-kernel-events drops them and the roofline object.  Workload (BASELINE.json configs[1], SURVEY §8d "c2"): net_type='fm',
1M users x 100K items x 100M interactions (80M train triples after the 0.8 split), dim=64, dynamic_neg_sampling=True,
batch 65 536, torch.optim.SGD(lr=1e-2), fp32.  metric = training interactions/s (pos+neg) = 2 x triples/s.

Multi-GPU: one process per GPU; the interaction stream is sharded across ranks (each rank owns an
independent 100M-interaction shard, weak scaling: per-GPU batch fixed); embedding tables are replicated and the FM has
no dense parameters, so the step has no data-path collective.  Rank 0 prints ONE JSON line.
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


def synth_stream(n_users, n_items, n, device, seed):
    """Uniform synthetic interactions with guaranteed dense id coverage, generated in HBM."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    users = torch.cat([torch.arange(n_users, device=device, dtype=torch.int32),
                       torch.randint(0, n_users, (n - n_users,), device=device, dtype=torch.int32, generator=g)])
    reps = -(-n_users // n_items)
    items = torch.cat([torch.arange(n_items, device=device, dtype=torch.int32).repeat(reps)[:n_users],
                       torch.randint(0, n_items, (n - n_users,), device=device, dtype=torch.int32, generator=g)])
    perm = torch.randperm(n, device=device, generator=g)
    return users[perm].contiguous(), items[perm].contiguous()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1024)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--optimizer", default="sgd", choices=["sgd", "sparse_adam", "adagrad", "adam"],
                    help="sgd = BASELINE.json's primary optimiser; sparse_adam (lazy Adam) / adagrad = secondary; adam = "
                         "torch.optim.Adam as in the reference's README (lazy rows on the tables, dense Adam on the MLP)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true", help="skip per-kernel HIP event timing")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (there is no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI

    from torchrecsys_amd import _lib
    from torchrecsys_amd.model import TorchRecSys
    _lib.check(_lib.load().trs_check_device(), "trs_check_device")

    cfg = CONFIGS[args.config]
    net, n_users, n_items, n_inter, D, B, desc = (cfg[k] for k in ("net", "n_users", "n_items", "n", "D", "B", "desc"))
    dynamic = args.config != "c1"
    users, items = synth_stream(n_users, n_items, n_inter, dev, seed=1000 + rank)
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
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        torch.manual_seed(7)
        kw = dict(hidden_layers=cfg["hidden"]) if net == "mlp" else {}
        model = TorchRecSys.from_tensors(users, items, n_users=n_users, n_items=n_items, item_metadata=meta,
                                         n_factors=D, net_type=net, split_ratio=0.8, dynamic_neg_sampling=dynamic,
                                         use_amp=cfg["amp"], rng="device", seed=7 + rank, pre_sharded=True, **kw)
        # pre_sharded: every rank generated its own interaction shard
    del users, items
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

    run(args.warmup)
    barrier()
    is_mlp = net == "mlp"
    if not args.no_kernel_events:
        if is_mlp:
            model.net.compute.gemm_events = []
            model.net.compute.gemm_steps_seen = model.net.compute.gemm_steps_timed = 0
        else:
            runner.trainer.kernel_events = {}
    t0 = time.perf_counter()
    run(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    events = (getattr(runner.trainer, "kernel_events", None) or {}) if not is_mlp else {}
    gemm_events = model.net.compute.gemm_events if is_mlp else None
    if is_mlp:
        model.net.compute.gemm_events = None
    else:
        runner.trainer.kernel_events = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    runner.end_epoch()  # also raises if any id was out of range

    triples = args.steps * B * world
    value = 2.0 * triples / elapsed
    M = len(cfg["meta"])
    R = 3 + 2 * M
    step_bytes = 16 + 2 * R * (4 * D + 4)  # FM/Linear fused SGD step, rows read once + written once
    state_rows = {"sgd": 0, "sparse_adam": 2, "adagrad": 1, "adam": 2}[args.optimizer]  # state tables read + written per row
    step_bytes += 2 * state_rows * R * (4 * D + 4)
    dtype = "bf16" if cfg["amp"] else "f32"
    out = {
        "metric": "training interactions/sec (pos+neg)", "value": value, "unit": "interactions/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
        "config": {"workload": desc, "global_batch": B * world, "per_gpu_batch": B,
                   "parallelism": (f"dp{world}: interaction stream sharded, tables replicated, one flat all-reduce of the "
                                   f"dense gradients per step" if is_mlp else
                                   f"dp{world}: interaction stream sharded, tables replicated, no per-step collective "
                                   f"(FM/Linear have no dense parameters)"),
                   "rng": "device (Feistel epoch shuffle + Philox4x32-10 negative sampler)"},
    }
    if not is_mlp:
        out["step_algorithmic_GBps_per_gpu"] = step_bytes * B * args.steps / elapsed / 1e9
    # ---- MLP: the GEMMs are the dominant kernels, bound by the matrix cores ----
    if gemm_events:
        dims = [(2 + M) * D] + list(cfg["hidden"])
        P = sum(a_ * b_ for a_, b_ in zip(dims[:-1], dims[1:])) + dims[-1]  # MACs per sample
        timed_steps = max(model.net.compute.gemm_steps_timed, 1)  # the GEMMs of one step in 5 carry events
        ms = sum(e0.elapsed_time(e1) for e0, e1, _ in gemm_events)
        fl = sum(f for _, _, f in gemm_events)
        ach = fl / (ms * 1e-3) / 1e12
        peak = MFMA_PEAK_TFLOPS[dtype]
        out["roofline"] = {"bound": "mfma", "kernel": "gemm_bf16in_kernel" if cfg["amp"] else "gemm_f32_kernel",
                           "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": None,
                           "gemm_launches_per_step": len(gemm_events) / timed_steps,
                           "gemm_ms_per_step": ms / timed_steps,
                           "gemm_share_of_step": (ms / timed_steps) / (1e3 * elapsed / args.steps),
                           "timed_steps": timed_steps,
                           "algorithmic_flops_per_triple": 12 * P,
                           "whole_step_TFLOPs": 12.0 * P * B * args.steps / elapsed / 1e12,
                           "note": "event intervals around the GEMM launches (HIP events on the launch stream); they "
                                   "include one event record each"}
    # ---- roofline of the dominant kernel: algorithmic bytes per launch / mean launch duration (HIP events) ----
    if events:
        def _ms(rec):  # (TimingEvents, i, j) from the C step loop, or (torch start, torch end) from the generic path
            return rec[0].elapsed_ms(rec[1], rec[2]) if len(rec) == 3 else rec[0].elapsed_time(rec[1])
        raw_ms = {k: sum(_ms(r) for r in v) / len(v) for k, v in events.items()}
        # An interval between two hipEventRecords contains the second record's own cost (a barrier packet + timestamp
        # write).  On the presorted path the last two events of a step are recorded back to back, so an upper bound of
        # that cost is measured live (two barrier packets in a row are slower than one behind a kernel).  `achieved` is
        # computed from the RAW intervals (conservative: rocprofv3's kernel durations in profiles/ are a little shorter);
        # the intervals minus the measured record cost are reported beside them as the lower bound.
        ev_ms = raw_ms.pop("event_overhead", 0.0)
        mean_ms = dict(raw_ms)
        dom = max(mean_ms, key=mean_ms.get)
        row = 4 * D + 4  # one embedding row + its 1-wide term
        inline_user = "sorted_updates_fused_kernel" in mean_ms or "sorted_item_update_kernel" in mean_ms
        # algorithmic bytes per triple of each kernel (DESIGN.md "Kernels"): the forward+backward pass reads the ids and
        # the R rows once and writes the two loss-gradient scalars; the update passes write the R rows once
        per_triple = {"score_kernel<fwd_bwd>": 16 + R * row + 8, "score_sgd_update_kernel": 12 + R * row,
                      # fast path (csrc/fast_step.hip): K1 reads the ids + R rows; on the presorted path it also writes
                      # the user row (users referenced once per batch are updated in place by K1)
                      "fwd_stage_kernel": 16 + R * row + 8 + ((1 + 2 * state_rows) * row if inline_user else 0),
                      # item updates: write the 2 item rows (+ read and write their state rows), need the user row
                      "item_update_kernel": 12 + 3 * row, "sorted_item_update_kernel": 12 + 3 * row,
                      "sorted_updates_fused_kernel": 12 + 3 * row + 2 * 2 * state_rows * row,
                      "user_update_kernel": 4 + row, "sorted_user_dup_update_kernel": 4 + row}
        ach = per_triple[dom] * B / (mean_ms[dom] * 1e-3) / 1e9
        # measured HBM bytes per launch of that kernel from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE,
        # corrected as MI355X_MICROARCH.md prescribes; tools/pmc_traffic.py); valid for the c2 workload on one GPU only
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if args.config == "c2" and os.path.exists(pmc_path):
            pk = json.load(open(pmc_path))["kernels"]
            parts = {"item_update_kernel": ("item_owner_update_kernel", "item_update_kernel")}.get(dom, (dom,))
            if all(q in pk for q in parts):
                traffic = sum(pk[q]["traffic_bytes_per_launch"] for q in parts)
        out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                           "mean_launch_us": {k: 1e3 * v for k, v in mean_ms.items()},
                           "mean_launch_us_minus_event_record": {k: 1e3 * max(v - ev_ms, 0.0) for k, v in mean_ms.items()},
                           "event_record_overhead_us": 1e3 * ev_ms,
                           "algorithmic_bytes_per_triple": per_triple[dom]}
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
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    from torchrecsys_amd.helper.cuda import host_threads
    with host_threads():  # torch's CPU pool capped at the container's CPU budget (N ranks share one host)
        main()
