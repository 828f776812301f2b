#!/usr/bin/env python3
"""The tiled decode GEMMs (gemm_dt.hip) at a decode step's row count, per launch inside a hipGraph of 22 launches over distinct
weights (HBM-cold like the decode loop): SwiGLU, lm_head, the three pair-sum GEMMs.  A/B by DUALHYP_HIP_LIB.  GPU box."""
import sys, os
sys.path.insert(0, '.')
import torch
from dualhyp_amd import ops, _lib
from tools.tune_decode_common import bench, L, D
lib = _lib.load()
d, I, V = 2048, 5632, 32000
M = int(os.environ.get("ROWS", "640"))
g = torch.Generator(device=D).manual_seed(1)
rn = lambda *s: (torch.randn(*s, device=D, generator=g) * 0.02).bfloat16()
W1, W2 = [rn(I, d) for _ in range(L)], [rn(I, d) for _ in range(L)]
Wq, A48, Wp, A16, Wm = [rn(2560, d) for _ in range(L)], [rn(48, d) for _ in range(L)], [rn(d, d) for _ in range(L)], [rn(16, d) for _ in range(L)], [rn(d, I) for _ in range(L)]
Wh = rn(V, d); sc, bi = torch.ones(V, device=D, dtype=torch.bfloat16), torch.zeros(V, device=D, dtype=torch.bfloat16)
x, xa = rn(M, d) * 50, rn(M, I) * 50
lib.dh_set_tuning(4, 2)          # decode phase
t = bench(lambda i: ops.linear(x, W1[i % L], epilogue=ops.EPI_SWIGLU, w2=W2[i % L])); print(f"rows {M} SwiGLU        {t:7.1f} us  {2*M*d*2*I/t/1e6:6.0f} TFLOP/s")
t = bench(lambda i: ops.linear(x, Wh, epilogue=ops.EPI_ADAPTER, scale=sc, bias=bi)); print(f"rows {M} lm_head       {t:7.1f} us  {2*M*d*V/t/1e6:6.0f} TFLOP/s")
t = bench(lambda i: ops.linear_partial_pairs(x, Wq[i % L], A48[i % L], ksplit=8)); print(f"rows {M} qkv' pairs    {t:7.1f} us")
t = bench(lambda i: ops.linear_partial_pairs(x, Wp[i % L], A16[i % L], ksplit=8)); print(f"rows {M} proj' pairs   {t:7.1f} us")
t = bench(lambda i: ops.linear_partial_pairs(xa, Wm[i % L], None, ksplit=11)); print(f"rows {M} mlp' pairs    {t:7.1f} us")
