#!/usr/bin/env python3
"""finish_norm (sum of K-slice partials + LoRA + residual + RMSNorm) at 32 and 640 rows (GPU box)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from dualhyp_amd import ops, _lib
from tools.tune_decode_common import bench, D
_lib.load()
d = 2048
wn = torch.ones(d, device=D).bfloat16(); Bp = torch.randn(d, 16, device=D).bfloat16()
for B in (32, 640):
    xr = torch.randn(B, d, device=D).bfloat16()
    for ks, lora in ((8, True), (11, False)):
        ys = [torch.randn(ks, B, d + (16 if lora else 0), device=D) for _ in range(4)]
        t = bench(lambda i: ops.finish_norm(ys[i % 4], d, xr, wn, 1e-5, lora_b=Bp if lora else None, lora_scale=1.0))
        print(f"rows {B:4d} partials {ks:2d} lora {lora}: {t:5.1f} us", flush=True)
