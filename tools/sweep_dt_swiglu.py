#!/usr/bin/env python3
"""Decode-phase SwiGLU (gemm_dt.hip MODE 1) vs rows: 4-wave 128 x 64-pair tiles against the 8-wave 128 x 128-pair tile
(dh_set_tuning(15, -1 | 1)), and that both give the same bits.  GPU box."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from dualhyp_amd import ops, _lib
from tools.tune_decode_common import bench, L, D
lib = _lib.load()
d, I = 2048, 5632
W1 = [torch.randn(I, d, device=D).bfloat16() * 0.02 for _ in range(L)]
W2 = [torch.randn(I, d, device=D).bfloat16() * 0.02 for _ in range(L)]
lib.dh_set_tuning(4, 2)
for M in (200, 256, 384, 512, 640, 1024, 2048):
    x = torch.randn(M, d, device=D).bfloat16()
    out = {}
    for wide in (-1, 1):
        lib.dh_set_tuning(15, wide)
        out[wide] = ops.linear(x, W1[0], epilogue=ops.EPI_SWIGLU, w2=W2[0])
        t = bench(lambda i: ops.linear(x, W1[i % L], epilogue=ops.EPI_SWIGLU, w2=W2[i % L]))
        print(f"M={M:5d} wide={wide:2d}: {t:6.1f} us  {2*M*d*2*I/t/1e6:6.0f} TFLOP/s", flush=True)
    print("        same bits:", bool(torch.equal(out[-1], out[1])))
Wl = [torch.randn(32000, d, device=D).bfloat16() * 0.02 for _ in range(3)]
sc = (1 + 0.1 * torch.randn(32000, device=D)).bfloat16(); bi = (0.1 * torch.randn(32000, device=D)).bfloat16()
for M in (256, 640, 2048):
    x = torch.randn(M, d, device=D).bfloat16()
    out = {}
    for wide in (-1, 1):
        lib.dh_set_tuning(15, wide)
        out[wide] = ops.linear(x, Wl[0], epilogue=ops.EPI_ADAPTER, scale=sc, bias=bi)
        t = bench(lambda i: ops.linear(x, Wl[i % 3], epilogue=ops.EPI_ADAPTER, scale=sc, bias=bi))
        print(f"lm_head M={M:5d} wide={wide:2d}: {t:6.1f} us  {2*M*d*32000/t/1e6:6.0f} TFLOP/s", flush=True)
    print("        same bits:", bool(torch.equal(out[-1], out[1])))
lib.dh_set_tuning(15, 0)
