#!/usr/bin/env python3
"""The four prefill GEMM launches of one TinyLlama layer at the bench shape (32 x 512 tokens), a few
times each and nothing else: the workload for the rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE),
which take minutes on the full bench.  Weights rotate so every launch streams them from HBM."""
import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from dualhyp_amd import ops
D = "cuda:0"
M, d, I, r = 2 * 32 * 512, 2048, 5632, 16     # two batches per prefill launch, as bench.py runs
g = torch.Generator(device=D).manual_seed(0)
rn = lambda *s: (torch.randn(*s, device=D, generator=g) * 0.05).bfloat16()
x, act = rn(M, d), rn(M, I)
R = 3
Wq, Wp, W1, W2, Wm = [rn(2560, d) for _ in range(R)], [rn(d, d) for _ in range(R)], [rn(I, d) for _ in range(R)], [rn(I, d) for _ in range(R)], [rn(d, I) for _ in range(R)]
A48, A16, Bq, Bp = rn(48, d), rn(16, d), rn(2560, 16), rn(d, 16)
H, G, hs, S = 32, 4, 64, 512
cos, sin = rn(S, hs), rn(S, hs)
nseq = M // S
kc = torch.zeros(nseq, G, S, hs, device=D, dtype=torch.bfloat16); vt = torch.zeros(nseq, G, hs, S, device=D, dtype=torch.bfloat16)
slot = torch.arange(nseq, device=D, dtype=torch.int32).repeat_interleave(S)
pos = torch.arange(S, device=D, dtype=torch.int32).repeat(nseq)
torch.cuda.synchronize()
for i in range(R):      # the bench's own launches (round 4): fused QKV with the in-GEMM LoRA down-projection, proj + LoRA + residual, SwiGLU, mlp proj + residual
    ops.linear_qkv_lora_rope_cache(x, Wq[i], A48, Bq, cos, sin, slot, pos, kc, vt, H, G)
    ops.linear_lora(x, Wp[i], A16, Bp, lora_scale=1.0, resid=x)
    ops.linear(x, W1[i], epilogue=ops.EPI_SWIGLU, w2=W2[i])
    ops.linear(act, Wm[i], resid=x)
torch.cuda.synchronize()
print("done")
