import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from conftest import load_golden, sub
import test_gpu_model as T
from torchrecsys_amd.mlp_engine import MLPTrainer
from oracle import nets as onets
name, M, oname = "mlp_nobn", 0, "adagrad"
g = load_golden(f"g2_{name}_M{M}_{oname}.npz")
net, b = T.build_net(name, M, g), T.golden_batch(g); net.train()
opt = torch.optim.Adagrad(list(net.parameters()), lr=0.05)
tr = MLPTrainer(net, opt, 64); ids = T.dev_ids(net, b)
losses = torch.zeros(3, device="cuda")
params = {k: v.cpu().numpy().copy() for k, v in net.state_dict().items()}
batch = {k: v.numpy() for k, v in b.items()}
sp, sn, loss, grads = onets.train_forward_backward("mlp", params, batch)
tr.step(ids, losses[0:1])
for k, v in sub(g, "step0").items():
    after = net.state_dict()[k].cpu().numpy(); d = np.abs(after - v)
    idx = np.argwhere(d > 1e-3)
    print(k, "n diff>1e-3:", len(idx), "of", d.size, "max", d.max())
    for ix in idx[:4]:
        ix = tuple(ix)
        print("     ", ix, "g_oracle", grads[k][ix], "init", params[k][ix], "ref", v[ix], "got", after[ix])
