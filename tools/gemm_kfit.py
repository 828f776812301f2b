#!/usr/bin/env python3
"""Per-tile fixed cost of the 256-tile GEMM: plain y = x W^T at M = 16384, N = 2048 (512 tiles = 2 per CU) for a range of K;
a least-squares line T(K) = a + b K / 64 gives the time per 64-deep iteration (b) and the cost per launch that does not scale with K
(a: launch, cold first stages, epilogue, store drain; two tiles per CU).  GPU box."""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from dualhyp_amd import ops, _lib
lib = _lib.load()
D = "cuda:0"
g = torch.Generator(device=D).manual_seed(0)
rn = lambda *s: (torch.randn(*s, device=D, generator=g) * 0.05).bfloat16()
M, N = 16384, 2048
for variant, persist in ((4, 0), (5, 0), (5, 1)):
    lib.dh_set_tuning(1, variant)
    lib.dh_set_tuning(22, persist)
    ks, ts = [], []
    for K in (512, 1024, 2048, 3072, 4096, 6144, 8192):
        x, w = rn(M, K), rn(N, K)
        y = torch.empty(M, N, device=D, dtype=torch.bfloat16)
        for _ in range(3):
            ops.linear(x, w, out=y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            ops.linear(x, w, out=y)
        e1.record()
        torch.cuda.synchronize()
        ks.append(K / 64); ts.append(e0.elapsed_time(e1) / 40 * 1e3)
    b, a = np.polyfit(ks, ts, 1)
    print(f"variant {variant} persist {persist}: " + "  ".join(f"K={int(k * 64)}: {t:.1f}us" for k, t in zip(ks, ts)))
    print(f"    fit: {a:.1f} us fixed per launch (2 tiles per CU) + {b / 2:.3f} us per 64-deep iteration of a tile  ->  steady-state {8.39e6 * 256 / (b / 2) * 1e-6:.0f} TFLOP/s", flush=True)
