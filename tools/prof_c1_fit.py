"""Scratch: cProfile of fit() at the c1 size (where does the host time of an epoch go)."""
import os, sys, time, io, contextlib, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd, torch
from torchrecsys_amd.model import TorchRecSys
net, dyn, B = sys.argv[1], sys.argv[2] == "1", int(sys.argv[3])
rs = np.random.RandomState(0)
n, nu, ni = 100_000, 3000, 1000
df = pd.DataFrame({"user_id": np.concatenate([np.arange(nu), rs.randint(0, nu, n - nu)]),
                   "item_id": np.concatenate([np.arange(ni), rs.randint(0, ni, n - ni)])})
with contextlib.redirect_stdout(io.StringIO()):
    model = TorchRecSys(dataset=df, user_id_col="user_id", item_id_col="item_id", n_factors=32, net_type=net,
                        dynamic_neg_sampling=dyn)
    opt = torch.optim.SGD(model.parameters(), lr=1e-2) if net != "mlp" else torch.optim.Adam(model.parameters(), lr=1e-3)
    model.fit(optimizer=opt, epochs=1, batch_size=B)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    model.fit(optimizer=opt, epochs=10, batch_size=B)
    torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr, stream=sys.stdout)
st.sort_stats("cumulative").print_stats(45)
