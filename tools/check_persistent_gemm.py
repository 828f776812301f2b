#!/usr/bin/env python3
"""256-tile prefill GEMM: persistent ping-pong (dh_set_tuning(1, 4)) against the per-tile launch (1, 1): same bits, and
time at the bench's launch shape (M = 2 x 32 x 512).  GPU box."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from dualhyp_amd import ops, _lib
lib = _lib.load()
D = "cuda:0"
M, d, I = 2 * 32 * 512, 2048, 5632
g = torch.Generator(device=D).manual_seed(0)
rn = lambda *s: (torch.randn(*s, device=D, generator=g) * 0.05).bfloat16()
x, act, res = rn(M, d), rn(M, I), rn(M, d)
L = 3
W1, W2, Wm = [rn(I, d) for _ in range(L)], [rn(I, d) for _ in range(L)], [rn(d, I) for _ in range(L)]
def bench(fn, n=10):
    for i in range(3): fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n): fn(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
outs = {}
for v in (1, 4):
    lib.dh_set_tuning(1, v)
    outs[v] = (ops.linear(x, W1[0], epilogue=ops.EPI_SWIGLU, w2=W2[0]).clone(), ops.linear(act, Wm[0], resid=res).clone(),
               ops.linear(x[:1000], W1[0], epilogue=ops.EPI_SWIGLU, w2=W2[0]).clone(), ops.linear(act[:777], Wm[0]).clone())
print("persistent == per-tile (swiglu, mlp proj + resid, ragged M swiglu, ragged M plain):", [bool(torch.equal(a, b)) for a, b in zip(outs[1], outs[4])])
for rep in range(2):
    for v in (1, 4):
        lib.dh_set_tuning(1, v)
        t1 = bench(lambda i: ops.linear(x, W1[i % L], epilogue=ops.EPI_SWIGLU, w2=W2[i % L]))
        t2 = bench(lambda i: ops.linear(act, Wm[i % L], resid=res))
        print(f"variant {v}: swiglu {t1*1e3:7.1f} us ({2*M*2*I*d/t1/1e9:6.0f} TF)   mlp proj {t2*1e3:7.1f} us ({2*M*d*I/t2/1e9:6.0f} TF)", flush=True)
lib.dh_set_tuning(1, 1)
