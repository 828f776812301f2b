#!/usr/bin/env python3
"""Micro-benchmark of the decode kernels at TinyLlama shapes (run on the GPU box).
Weights rotate over 22 distinct buffers so they come from HBM, as in the real layer loop."""
import sys, time, torch
sys.path.insert(0, '.')
from dualhyp_amd import ops, _lib
D = "cuda:0"
L = 22
def bench(fn, reps=20):
    """Per-call microseconds of a hipGraph of L back-to-back calls (host launch cost excluded)."""
    for i in range(L): fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            for i in range(L): fn(i)
    torch.cuda.synchronize()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (reps * L) * 1e3
d, I, M = 2048, 5632, 32
x = torch.randn(M, d, device=D).bfloat16(); xa = torch.randn(M, I, device=D).bfloat16()
Wq = [torch.randn(2560, d, device=D).bfloat16() * 0.02 for _ in range(L)]
A48 = [torch.randn(48, d, device=D).bfloat16() * 0.02 for _ in range(L)]
Wp = [torch.randn(d, d, device=D).bfloat16() * 0.02 for _ in range(L)]
A16 = [torch.randn(16, d, device=D).bfloat16() * 0.02 for _ in range(L)]
W1 = [torch.randn(I, d, device=D).bfloat16() * 0.02 for _ in range(L)]
W2 = [torch.randn(I, d, device=D).bfloat16() * 0.02 for _ in range(L)]
Wm = [torch.randn(d, I, device=D).bfloat16() * 0.02 for _ in range(L)]
lib = _lib.load()
for variant in (0, 1):
    lib.dh_set_tuning(0, variant)
    for ks in ((1, 2, 4) if variant == 0 else (8,)):
        t = bench(lambda i: ops.linear_partial(x, Wq[i % L], A48[i % L], ksplit=ks))
        print(f"variant {variant} qkv' ks={ks}: {t:6.1f} us  {2608*d*2/t/1e6:5.2f} TB/s")
        t = bench(lambda i: ops.linear_partial(x, Wp[i % L], A16[i % L], ksplit=ks))
        print(f"variant {variant} proj' ks={ks}: {t:6.1f} us  {2064*d*2/t/1e6:5.2f} TB/s")
    for ks in ((1, 2, 4) if variant == 0 else (11,)):
        t = bench(lambda i: ops.linear_partial(xa, Wm[i % L], None, ksplit=ks))
        print(f"variant {variant} mlp ks={ks}: {t:6.1f} us  {d*I*2/t/1e6:5.2f} TB/s")
Wl = [torch.randn(32000, d, device=D).bfloat16() * 0.02 for _ in range(3)]
sc = torch.ones(32000, device=D).bfloat16(); bi = torch.zeros(32000, device=D).bfloat16()
for v in (0, 1):
    lib.dh_set_tuning(2, v)
    t = bench(lambda i: ops.linear(x, W1[i % L], epilogue=ops.EPI_SWIGLU, w2=W2[i % L]))
    print(f"swiglu (32-row variant={v}): {t:6.1f} us  {2*I*d*2/t/1e6:5.2f} TB/s")
t = bench(lambda i: ops.linear(x, Wl[i % 3], epilogue=ops.EPI_ADAPTER, scale=sc, bias=bi))
print(f"lm_head: {t:6.1f} us  {32000*d*2/t/1e6:5.2f} TB/s")

t = bench(lambda i: ops.linear(x, Wq[i % L]))
print(f"plain skinny qkv: {t:6.1f} us  {2560*d*2/t/1e6:5.2f} TB/s")
y32 = ops.linear_partial(x, Wp[0], A16[0], ksplit=2)
xr = torch.randn(M, d, device=D).bfloat16(); wn = torch.ones(d, device=D).bfloat16(); Bp = torch.randn(d, 16, device=D).bfloat16()
t = bench(lambda i: ops.finish_norm(y32, d, xr, wn, 1e-5, lora_b=Bp, lora_scale=1.0))
print(f"finish_norm(2 parts): {t:6.1f} us")
y8 = ops.linear_partial(x, Wp[0], A16[0], ksplit=8)
t = bench(lambda i: ops.finish_norm(y8, d, xr, wn, 1e-5, lora_b=Bp, lora_scale=1.0))
print(f"finish_norm(8 parts): {t:6.1f} us")
t = bench(lambda i: ops.rmsnorm(x, wn, 1e-5))
print(f"rmsnorm: {t:6.1f} us   (includes torch.empty + ctypes overhead)")
# fused attention at S ~ 544 (mid-decode), 32 sequences, TinyLlama heads
H, G, hs, S, B = 32, 4, 64, 640, 32
kc = [torch.randn(B, G, S, hs, device=D).bfloat16() for _ in range(L)]
vt = [torch.randn(B, G, hs, S, device=D).bfloat16() for _ in range(L)]
cos = torch.randn(S, hs, device=D).bfloat16(); sin = torch.randn(S, hs, device=D).bfloat16()
Bq = torch.randn(2560, 16, device=D).bfloat16() * 0.02
i32 = torch.int32
slot = torch.arange(B, dtype=i32, device=D); kvl = torch.full((B,), 545, dtype=i32, device=D)
lib.dh_set_tuning(0, 1)
for ks in (2, 8):
    q32 = torch.randn(ks, B, 2608, device=D) * 0.1
    t = bench(lambda i: ops.attn_decode_fused(q32, 2560, Bq, 1.0, (2048, 2304), cos, sin, slot, kvl, kc[i % L], vt[i % L], H))
    print(f"attn_decode_fused (S=545, {ks} partials): {t:6.1f} us  kv {B*G*545*hs*2*2/t/1e6:5.2f} TB/s")
