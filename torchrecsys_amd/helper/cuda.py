# -*- coding: utf-8 -*-
"""Device placement policy of the path (the reference's helper/cuda.py:3-16 `.cuda()` / `.cpu()` shims).

In this framework the compute device is always the MI355X PyTorch-ROCm exposes as `cuda`; the flag only decides
whether an object is moved there now.  `gpu(x, False)` returns `x` untouched (batches stay host tensors until a scorer
uploads them; scorers refuse to compute on host tensors' device)."""
import contextlib
import os

import torch


def _accelerator():
    if not torch.cuda.is_available():
        raise RuntimeError("no MI355X visible to PyTorch-ROCm: torchrecsys_amd has no CPU compute path")
    return torch.device("cuda", torch.cuda.current_device())


def gpu(tensor, gpu=False):
    """Move `tensor` (or a module) to the accelerator when `gpu` is true."""
    return tensor.to(_accelerator()) if gpu else tensor


def cpu(tensor):
    """Host copy of an accelerator tensor (identity for host tensors)."""
    return tensor.to("cpu") if getattr(tensor, "is_cuda", False) else tensor


def cpu_budget():
    """CPUs this process can really use: its affinity mask, capped by the cgroup CPU quota of a container."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # cgroup v2: "<quota> <period>" or "max <period>"
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:  # cgroup v1
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and period > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return max(1, n)


@contextlib.contextmanager
def host_threads():
    """Cap torch's intra-op CPU threads for the host side of fit() / evaluate() / predict().

    torch sizes its thread pool from the machine's core count; inside a container with a CPU quota the surplus threads
    spin the quota away and every host-side piece of an epoch (the reference-RNG permutation, the shuffle gathers, the
    uploads, even the kernel-launch loop) stalls for milliseconds — measured on a 256-core MI355X host with a 16-CPU
    quota: 17 ms per epoch of 78 steps against 2.5 ms.  The host work of the path is small index arithmetic, so the cap
    is min(budget, 8); TRS_HOST_THREADS overrides; the previous setting is restored on exit."""
    want = int(os.environ.get("TRS_HOST_THREADS", "0")) or min(cpu_budget(), 8)
    cur = torch.get_num_threads()
    if cur <= want:
        yield
        return
    torch.set_num_threads(want)
    try:
        yield
    finally:
        torch.set_num_threads(cur)
