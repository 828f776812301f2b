#!/usr/bin/env python3
"""Is the tiled fp8 GEMM held back by the clock the chip keeps under load?  Same launches on random and on all-zero operands
(MI355X_MICROARCH.md, DVFS give-back: zero data draws less power and holds a higher clock).  GPU box."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from dualhyp_amd import ops, _lib
_lib.load()
D = "cuda:0"
def bench(fn, reps=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
M = 32 * 1536
for (nm, N, K, sw) in (("swiglu", 14336, 4096, True), ("mlp_proj", 4096, 14336, False), ("qkv", 6144, 4096, False)):
    for kind in ("random", "zeros"):
        mk = (lambda *s: (torch.randint(0, 255, s, dtype=torch.uint8, device=D) & 0x77)) if kind == "random" else (lambda *s: torch.zeros(s, dtype=torch.uint8, device=D))
        xq, wq, w2 = mk(M, K), mk(N, K), mk(N, K)
        xs, ws = torch.ones(M, device=D), torch.ones(N, device=D)
        kw = dict(epilogue=ops.EPI_SWIGLU, w2q=w2, w2_scale=ws) if sw else {}
        t = bench(lambda: ops.linear_fp8(xq, xs, wq, ws, **kw))
        fl = 2.0 * M * N * K * (2 if sw else 1)
        print(f"{nm:9s} {kind:7s}: {t:7.3f} ms  {fl / t / 1e9:7.0f} TFLOP/s", flush=True)
