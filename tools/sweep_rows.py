#!/usr/bin/env python3
"""K-sliced partial-sum kernels (gemm_skinny_rows_kernel) of a TinyLlama decode layer vs row count (GPU box).
Optional env DH_TUNE="k=v,k=v" -> dh_set_tuning."""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from dualhyp_amd import ops, _lib
from tools.tune_decode_common import bench, L, D
lib = _lib.load()
for kv in filter(None, os.environ.get("DH_TUNE", "").split(",")):
    k, v = kv.split("="); assert lib.dh_set_tuning(int(k), int(v)) == 0
d, I = 2048, 5632
Wq = [torch.randn(2560, d, device=D).bfloat16() * 0.02 for _ in range(L)]
A48 = [torch.randn(48, d, device=D).bfloat16() * 0.02 for _ in range(L)]
Wp = [torch.randn(d, d, device=D).bfloat16() * 0.02 for _ in range(L)]
A16 = [torch.randn(16, d, device=D).bfloat16() * 0.02 for _ in range(L)]
Wm = [torch.randn(d, I, device=D).bfloat16() * 0.02 for _ in range(L)]
rows = [int(r) for r in os.environ.get("DH_ROWS", "32,64,128,256,288,320,512,640,768,1024").split(",")]
for M in rows:
    x = torch.randn(M, d, device=D).bfloat16(); xa = torch.randn(M, I, device=D).bfloat16()
    tq = bench(lambda i: ops.linear_partial(x, Wq[i % L], A48[i % L], ksplit=8))
    tp = bench(lambda i: ops.linear_partial(x, Wp[i % L], A16[i % L], ksplit=8))
    tm = bench(lambda i: ops.linear_partial(xa, Wm[i % L], None, ksplit=11))
    fl = lambda n, k: 2.0 * M * n * k
    print(f"M={M:5d}  qkv' {tq:6.1f} us ({fl(2608, d)/tq/1e6:5.0f} TF)   proj' {tp:6.1f} us ({fl(2064, d)/tp/1e6:5.0f} TF)   "
          f"mlp' {tm:6.1f} us ({fl(d, I)/tm/1e6:5.0f} TF)", flush=True)
