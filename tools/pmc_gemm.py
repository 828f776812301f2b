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
xa48, xa16, Bq, Bp = rn(M, 48), rn(M, 16), rn(2560, 16), rn(d, 16)
torch.cuda.synchronize()
for i in range(R):
    ops.linear(x, Wq[i], epilogue=ops.EPI_LORA, xa=xa48, lora_b=Bq, lora_scale=1.0, splits=(2048, 2304))
    ops.linear(x, Wp[i], epilogue=ops.EPI_LORA, xa=xa16, lora_b=Bp, lora_scale=1.0, splits=(d, d), resid=x)
    ops.linear(x, W1[i], epilogue=ops.EPI_SWIGLU, w2=W2[i])
    ops.linear(act, Wm[i], resid=x)
torch.cuda.synchronize()
print("done")
