#!/usr/bin/env python3
"""The decode-step GEMMs of one TinyLlama layer at 32 (or argv[1]) rows, a few launches each on rotating weights:
workload for `rocprofv3 --pmc FETCH_SIZE --kernel-trace` (fabric bytes per launch vs the weights' size)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from dualhyp_amd import ops, _lib
lib = _lib.load()
D, L = "cuda:0", 6
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32
d, I = 2048, 5632
W1 = [torch.randn(I, d, device=D).bfloat16() * 0.02 for _ in range(L)]
W2 = [torch.randn(I, d, device=D).bfloat16() * 0.02 for _ in range(L)]
Wq = [torch.randn(2560, d, device=D).bfloat16() * 0.02 for _ in range(L)]
Wp = [torch.randn(d, I, device=D).bfloat16() * 0.02 for _ in range(L)]
x = torch.randn(M, d, device=D).bfloat16()
xi = torch.randn(M, I, device=D).bfloat16()
lib.dh_set_tuning(3, 1); lib.dh_set_tuning(4, 2)
for i in range(2 * L):
    ops.linear(x, W1[i % L], epilogue=ops.EPI_SWIGLU, w2=W2[i % L])
    ops.linear_partial(x, Wq[i % L], None, ksplit=4)
    ops.linear_partial(xi, Wp[i % L], None, ksplit=4)
torch.cuda.synchronize()
print("done")
