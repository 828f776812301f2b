#!/usr/bin/env python3
"""sample_kernel (temperature -> arg-max / top-k draw -> append) at 32 and 640 rows of 32000 logits (GPU box)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from dualhyp_amd import ops, _lib
from tools.tune_decode_common import bench, D
_lib.load()
V = 32000
for B in (32, 640):
    lg = [torch.randn(B, V, device=D).bfloat16() for _ in range(4)]
    tokens = torch.zeros((B, 4096), dtype=torch.int64, device=D)
    length = torch.zeros(B, dtype=torch.int32, device=D)
    done = torch.zeros(B, dtype=torch.int32, device=D)
    for top_k in (1, 50):
        def f(i):
            ops.sample(lg[i % 4], tokens, length, done, temperature=0.2, top_k=top_k, eos_id=None, seed=1, step=i)
        t = bench(f)
        print(f"rows {B:4d} top_k {top_k:2d}: {t:5.1f} us", flush=True)
