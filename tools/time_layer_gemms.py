#!/usr/bin/env python3
"""Time of each prefill GEMM launch of a TinyLlama layer at the bench's shape (M = 2 x 32 x 512), HIP events over back-to-back
launches, for the library DUALHYP_HIP_LIB selects: the same-box A/B of GEMM kernel changes.  GPU box."""
import sys, os
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from dualhyp_amd import ops, _lib
D = "cuda:0"
M, d, I = int(os.environ.get("M", 2 * 32 * 512)), 2048, 5632
g = torch.Generator(device=D).manual_seed(0)
rn = lambda *s: (torch.randn(*s, device=D, generator=g) * 0.05).bfloat16()
x, act = rn(M, d), rn(M, I)
W1, W2, Wm, Wq, Wp = rn(I, d), rn(I, d), rn(d, I), rn(2560, d), rn(d, d)
A48, A16, Bq, Bp = rn(48, d), rn(16, d), rn(2560, 16), rn(d, 16)
H, G, hs, S = 32, 4, 64, 512
cos, sin = rn(S, hs), rn(S, hs)
nseq = M // S
kc = torch.zeros(nseq, G, S, hs, device=D, dtype=torch.bfloat16); vt = torch.zeros(nseq, G, hs, S, device=D, dtype=torch.bfloat16)
slot = torch.arange(nseq, device=D, dtype=torch.int32).repeat_interleave(S)
pos = torch.arange(S, device=D, dtype=torch.int32).repeat(nseq)
res = rn(M, d)
cases = [("qkv + LoRA + rope + cache", lambda: ops.linear_qkv_lora_rope_cache(x, Wq, A48, Bq, cos, sin, slot, pos, kc, vt, H, G), 2 * M * 2560 * d + 2 * M * 48 * d),
         ("attn proj + LoRA + resid", lambda: ops.linear_lora(x, Wp, A16, Bp, lora_scale=1.0, resid=res), 2 * M * d * d + 2 * M * 16 * d),
         ("fc_1 / fc_2 SwiGLU", lambda: ops.linear(x, W1, epilogue=ops.EPI_SWIGLU, w2=W2), 2 * M * 2 * I * d),
         ("mlp proj + resid", lambda: ops.linear(act, Wm, resid=res), 2 * M * d * I)]
reps = int(os.environ.get("REPS", 20))
tot = 0.0
for nm, fn, fl in cases:
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    tot += us
    print(f"{nm:28s} {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s  ({fl / us / 1e6 / 2500:.3f} of 2.5 PFLOP/s)")
print(f"{'layer total':28s} {tot:8.1f} us")
