"""Scratch: K1 duration under variants (scratch on/off, ids given / from stream)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchrecsys_amd import ops
dev = "cuda:0"
NU, NI, D, B, N = 1_000_000, 100_000, 64, 65536, 100_000_000
g = torch.Generator(device=dev); g.manual_seed(0)
user = torch.randn(NU, D, device=dev, generator=g) / D; item = torch.randn(NI, D, device=dev, generator=g) / D
ul = torch.randn(NU, 1, device=dev, generator=g); il = torch.randn(NI, 1, device=dev, generator=g)
su = torch.randint(0, NU, (N,), device=dev, dtype=torch.int32, generator=g)
si = torch.randint(0, NI, (N,), device=dev, dtype=torch.int32, generator=g)
T, keep = ops.make_tables(user, item, ul, il)
sui = ops.interleave_stream(su, si)
err = torch.zeros(1, dtype=torch.int32, device=dev)
gz = torch.empty((2, B), device=dev); du = torch.empty((B, D), device=dev); ls = torch.zeros(256, device=dev)
bufs = [torch.empty(B, dtype=torch.int32, device=dev) for _ in range(3)]
scratch = ops.train_scratch(NU, NI, B, D, dev)
outs = [ops.batch_prepare(su, si, None, 0x1234567, i * B, B, NI, 7, i * B) for i in range(64)]
stamp = [1]
for sc in (scratch, None):
    for r in range(2):   # from stream
        ops.train_steps_sgd("fm", T, sui, None, 0x1234567, 7, r * 64 * B, B, 64, 0.01, *bufs, gz, du, ls[:64], err, sc, stamp[0]); stamp[0] += 64
    for r in range(2):   # ids given
        for i in range(64):
            o = outs[i]
            ops.train_steps_sgd("fm", T, None, None, 0, 0, 0, B, 1, 0.01, o["user"], o["pos"], o["neg"], gz, du, ls[i:i+1], err, sc, stamp[0]); stamp[0] += 1
torch.cuda.synchronize()
