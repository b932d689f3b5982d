import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import load_golden, sub
import test_gpu_model as T
from torchrecsys_amd.engine import SparseScorerTrainer
g = load_golden("g2_fm_M1_adagrad.npz")
net, b = T.build_net("fm", 1, g), T.golden_batch(g)
opt = torch.optim.Adagrad(list(net.parameters()), lr=0.05)
tr = SparseScorerTrainer(net, opt, 64)
ids = T.dev_ids(net, b)
losses = torch.zeros(3, device="cuda")
tr.step(ids, losses[0:1])
ref = sub(g, "step0"); init = sub(g, "init")
for k in ref:
    a = net.state_dict()[k].cpu().numpy(); d = np.abs(a - ref[k])
    print(k, d.max(), (d > 1e-5).sum(), d.size)
    for (i, j) in np.argwhere(d > 1e-5)[:6]:
        print("   ", i, j, "init", init[k][i, j], "ref", ref[k][i, j], "got", a[i, j], "ref upd", ref[k][i,j]-init[k][i,j], "got upd", a[i,j]-init[k][i,j])
print("users", b["user_id"].tolist()); print("pos", b["pos_item_id"].tolist()); print("neg", b["neg_item_id"].tolist())
