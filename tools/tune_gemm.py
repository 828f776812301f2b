#!/usr/bin/env python3
"""Prefill GEMM micro-benchmark at the bench shapes (M = 32 x 512 tokens), TFLOP/s per variant."""
import sys, torch
sys.path.insert(0, '.')
from dualhyp_amd import ops, _lib
D = "cuda:0"
lib = _lib.load()
M, d, I = 16384, 2048, 5632
x = (torch.randn(M, d, device=D) * 0.5).bfloat16()
act = (torch.randn(M, I, device=D) * 0.5).bfloat16()
L = 4
Wq = [(torch.randn(2560, d, device=D) * 0.02).bfloat16() for _ in range(L)]
Wp = [(torch.randn(d, d, device=D) * 0.02).bfloat16() for _ in range(L)]
W1 = [(torch.randn(I, d, device=D) * 0.02).bfloat16() for _ in range(L)]
W2 = [(torch.randn(I, d, device=D) * 0.02).bfloat16() for _ in range(L)]
Wm = [(torch.randn(d, I, device=D) * 0.02).bfloat16() for _ in range(L)]
xa = (torch.randn(M, 48, device=D) * 0.1).bfloat16(); Bq = (torch.randn(2560, 16, device=D) * 0.02).bfloat16()
res = (torch.randn(M, d, device=D)).bfloat16()
yq = torch.empty(M, 2560, device=D, dtype=torch.bfloat16); yd = torch.empty(M, d, device=D, dtype=torch.bfloat16); ya = torch.empty(M, I, device=D, dtype=torch.bfloat16)
def bench(fn, n=12):
    for i in range(3): fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n): fn(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3
outs = {}
for variant in (1, 2, 3):
    lib.dh_set_tuning(1, variant)
    outs[variant] = (ops.linear(x, Wq[0], epilogue=ops.EPI_LORA, xa=xa, lora_b=Bq, splits=(2048, 2304)).clone(),
                     ops.linear(x, W1[0], epilogue=ops.EPI_SWIGLU, w2=W2[0]).clone(), ops.linear(act, Wm[0], resid=res).clone())
for v in (2, 3):
    print(f"variant {v} == variant 1 bitwise:", [bool(torch.equal(a, b)) for a, b in zip(outs[1], outs[v])])
for variant in (1, 2, 3):
    lib.dh_set_tuning(1, variant)
    t = bench(lambda i: ops.linear(x, Wq[i % L], epilogue=ops.EPI_LORA, xa=xa, lora_b=Bq, splits=(2048, 2304), out=yq))
    print(f"variant {variant} qkv+lora   : {t*1e6:7.1f} us {2*M*2560*d/t/1e12:7.1f} TF")
    t = bench(lambda i: ops.linear(x, Wp[i % L], resid=res, out=yd))
    print(f"variant {variant} proj+resid : {t*1e6:7.1f} us {2*M*d*d/t/1e12:7.1f} TF")
    t = bench(lambda i: ops.linear(x, W1[i % L], epilogue=ops.EPI_SWIGLU, w2=W2[i % L], out=ya))
    print(f"variant {variant} swiglu     : {t*1e6:7.1f} us {2*M*2*I*d/t/1e12:7.1f} TF")
    t = bench(lambda i: ops.linear(act, Wm[i % L], resid=res, out=yd))
    print(f"variant {variant} mlp+resid  : {t*1e6:7.1f} us {2*M*d*I/t/1e12:7.1f} TF")
