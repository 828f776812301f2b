#!/usr/bin/env python3
"""Decode partial-sum GEMMs at the driver's 640 rows (and neighbours): K-sliced streaming kernel (slices) against the tiled
split-K kernel (pair sums, 128 x 128 and 128 x 256 tiles), plus the consumers on either input (GPU box)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dualhyp_amd import ops, _lib
from tune_decode_common import bench
lib = _lib.load()
dev = "cuda"
d, I, L = 2048, 5632, 8
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.05).bfloat16()
Wq, Wp, Wm = [rn(2560, d) for _ in range(L)], [rn(d, d) for _ in range(L)], [rn(d, I) for _ in range(L)]
A48, A16 = [rn(48, d) for _ in range(L)], [rn(16, d) for _ in range(L)]
for M in [int(a) for a in sys.argv[1:]] or [160, 256, 640, 1024, 2048]:
    x, xa = rn(M, d), rn(M, I)
    xr, wn = rn(M, d), rn(d)
    row = [f"M={M:5d}"]
    for name, xin, W, A, ks in (("qkv'", x, Wq, A48, 8), ("proj'", x, Wp, A16, 8), ("mlp'", xa, Wm, None, 11)):
        t0 = bench(lambda i: ops.linear_partial(xin, W[i % L], A[i % L] if A else None, ksplit=ks))
        ts = []
        for wnv in (2, 4):
            lib.dh_set_tuning(17, wnv)
            ts.append(bench(lambda i: ops.linear_partial_pairs(xin, W[i % L], A[i % L] if A else None, ksplit=ks)))
        lib.dh_set_tuning(17, 0)
        row.append(f"{name} slices {t0:5.1f} | pairs 128x128 {ts[0]:5.1f} 128x256 {ts[1]:5.1f}")
    y8 = ops.linear_partial(x, Wp[0], A16[0], ksplit=8)
    y4 = ops.linear_partial_pairs(x, Wp[0], A16[0], ksplit=8)
    Bp = rn(d, 16)
    f8 = bench(lambda i: ops.finish_norm(y8, d, xr, wn, 1e-5, lora_b=Bp, lora_scale=1.0, pairs=True))
    f4 = bench(lambda i: ops.finish_norm(y4, d, xr, wn, 1e-5, lora_b=Bp, lora_scale=1.0, pairs=False))
    row.append(f"finish 8 slices {f8:5.1f} | 4 pairs {f4:5.1f}")
    print("  ".join(row), flush=True)
