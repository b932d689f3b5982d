"""Scratch micro-benchmark of the FM c2 step (not the contract bench)."""
import sys, time
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchrecsys_amd import ops

dev = "cuda:0"
NU, NI, D, B = 1_000_000, 100_000, int(sys.argv[1]) if len(sys.argv) > 1 else 64, 65536
N = 20_000_000
g = torch.Generator(device=dev); g.manual_seed(0)
user = torch.randn(NU, D, device=dev, generator=g) / D
item = torch.randn(NI, D, device=dev, generator=g) / D
ul = torch.randn(NU, 1, device=dev, generator=g)
il = torch.randn(NI, 1, device=dev, generator=g)
su = torch.randint(0, NU, (N,), device=dev, dtype=torch.int32, generator=g)
si = torch.randint(0, NI, (N,), device=dev, dtype=torch.int32, generator=g)
T, keep = ops.make_tables(user, item, ul, il)
R = 3
gr = torch.empty((R, B, D), device=dev); gl = torch.empty((R, B), device=dev)
out = {k: torch.empty(B, dtype=torch.int32, device=dev) for k in ("user", "pos", "neg")}
loss = torch.zeros(1, device=dev)
err = torch.zeros(1, dtype=torch.int32, device=dev)
Bt, kb = ops.make_batch(out["user"], out["pos"], out["neg"], None, None, err)

def step(i):
    ops.batch_prepare(su, si, None, 0x1234567, (i * B) % (N - B), B, NI, 7, i * B, None, out)
    ops.score_fwd_bwd("fm", T, Bt, B, D, 0, dev, loss, None, False, gr, gl)
    ops.score_sgd_update("fm", T, Bt, gr, gl, 0.01)

for i in range(20): step(i)
torch.cuda.synchronize()
K = 200
t0 = time.perf_counter()
for i in range(K): step(20 + i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print(f"D={D} step {dt*1e6:.1f} us  triples/s {B/dt/1e6:.1f} M  interactions/s {2*B/dt/1e6:.1f} M  algo GB/s {B*(16+2*3*(4*D+4))/dt/1e9:.0f}")
# per-kernel with events
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
acc = [0.0, 0.0, 0.0]
for i in range(50):
    ev[0].record(); ops.batch_prepare(su, si, None, 0x1234567, (i * B) % (N - B), B, NI, 7, i * B, None, out)
    ev[1].record(); ops.score_fwd_bwd("fm", T, Bt, B, D, 0, dev, loss, None, False, gr, gl)
    ev[2].record(); ops.score_sgd_update("fm", T, Bt, gr, gl, 0.01)
    ev[3].record(); torch.cuda.synchronize()
    for j in range(3): acc[j] += ev[j].elapsed_time(ev[j + 1])
print("prepare %.1f us  fwd_bwd %.1f us  sgd_update %.1f us" % tuple(a / 50 * 1e3 for a in acc))
# forward only
ps = torch.empty(B, device=dev); ns = torch.empty(B, device=dev)
import ctypes as C
from torchrecsys_amd import _lib
lib = _lib.load()
for _ in range(10): lib.trs_score_forward(1, C.byref(T), C.byref(Bt), ps.data_ptr(), ns.data_ptr(), torch.cuda.current_stream().cuda_stream)
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100): lib.trs_score_forward(1, C.byref(T), C.byref(Bt), ps.data_ptr(), ns.data_ptr(), torch.cuda.current_stream().cuda_stream)
e1.record(); torch.cuda.synchronize()
tf = e0.elapsed_time(e1) / 100 * 1e-3
print(f"forward-only {tf*1e6:.1f} us  algo GB/s {B*(16+3*(4*D+4)+8)/tf/1e9:.0f}  ({B*(16+3*(4*D+4)+8)/tf/8e12*100:.1f}% of 8 TB/s)  [same batch re-read: cache-warm]")
