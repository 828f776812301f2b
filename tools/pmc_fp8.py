#!/usr/bin/env python3
"""The tiled fp8 GEMM at the Llama-3-8B prefill shapes, a few launches each: workload for rocprofv3 --pmc passes."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from dualhyp_amd import ops, _lib
_lib.load()
D = "cuda:0"
M = 32 * 1536
mk = lambda *s: (torch.randint(0, 255, s, dtype=torch.uint8, device=D) & 0x77)
xs = torch.ones(M, device=D)
for (N, K, sw) in ((14336, 4096, True), (4096, 14336, False), (6144, 4096, False)):
    xq, wq, w2 = mk(M, K), mk(N, K), mk(N, K)
    ws = torch.ones(N, device=D)
    kw = dict(epilogue=ops.EPI_SWIGLU, w2q=w2, w2_scale=ws) if sw else {}
    for _ in range(3): ops.linear_fp8(xq, xs, wq, ws, **kw)
    torch.cuda.synchronize()
print("done")
