# -*- coding: utf-8 -*-
"""Where the time of the one-launch flag-mode step (fwd_stage_kernel<..., INL 3>) goes, per workgroup: a DIAGNOSTIC build
(`make -C torchrecsys_amd/csrc EXTRA=-DTRS_K1_STAMPS`) stamps s_memrealtime at the kernel's start, the end of the main
loop, the end of the grid-wide wait and the end of the deferred atomics into the (otherwise unused) gz buffer; this script
runs c4-shaped steps and prints the distribution over the 512 workgroups of the last step.  Not for the product build."""
import contextlib
import io
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
model = bench.build_model("c4", 20_000_000, dev)
opt = torch.optim.SGD(model.parameters(), lr=1e-2)
from torchrecsys_amd.engine import SparseScorerTrainer  # noqa: E402
SparseScorerTrainer.SLICE_BATCHES = 64
r = model.make_runner(opt, 32768)
model.net.train()
r.begin_epoch()
for rep in range(6):
    r.run_steps(9)
    torch.cuda.synchronize()
    st = r.trainer.gz.view(torch.int64).reshape(-1)[:4 * 512].reshape(512, 4).cpu().numpy().astype(np.float64) * 0.01  # us
    t0 = st[:, 0].min()
    q = lambda a: "min %.1f  p50 %.1f  p90 %.1f  max %.1f" % (a.min(), np.median(a), np.percentile(a, 90), a.max())
    print(f"rep {rep}: start {q(st[:, 0] - t0)} | main loop end {q(st[:, 1] - t0)} | wait end {q(st[:, 2] - t0)} | "
          f"deferred end {q(st[:, 3] - t0)}")
