"""Scratch: end-to-end wall time of the reference-style flow at the c1 size (DataFrame in, fit() of 10 epochs)."""
import os, sys, time, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, pandas as pd, torch
from torchrecsys_amd.model import TorchRecSys
print("threads", torch.get_num_threads(), "cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), flush=True)
rs = np.random.RandomState(0)
n, nu, ni = 100_000, 3000, 1000
df = pd.DataFrame({"user_id": np.concatenate([np.arange(nu), rs.randint(0, nu, n - nu)]),
                   "item_id": np.concatenate([np.arange(ni), rs.randint(0, ni, n - ni)])})
for net, dyn, B in (("linear", False, 1024), ("fm", True, 1024), ("mlp", True, 1024), ("fm", True, 512)):
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        model = TorchRecSys(dataset=df, user_id_col="user_id", item_id_col="item_id", n_factors=32, net_type=net,
                            dynamic_neg_sampling=dyn)
    t1 = time.perf_counter()
    opt = torch.optim.SGD(model.parameters(), lr=1e-2) if net != "mlp" else torch.optim.Adam(model.parameters(), lr=1e-3)
    with contextlib.redirect_stdout(io.StringIO()):
        model.fit(optimizer=opt, epochs=1, batch_size=B)  # first epoch: allocations, first presort
    torch.cuda.synchronize(); t2 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        model.fit(optimizer=opt, epochs=10, batch_size=B)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        model.evaluate(batch_size=B)
    torch.cuda.synchronize(); t4 = time.perf_counter()
    print(f"{net:6s} dyn={dyn} B={B}: ctor {1e3*(t1-t0):7.1f} ms, first epoch {1e3*(t2-t1):7.1f} ms, "
          f"then {1e2*(t3-t2):6.2f} ms/epoch ({n*0.8/B:.0f} steps), evaluate {1e3*(t4-t3):6.1f} ms", flush=True)
