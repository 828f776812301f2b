#!/usr/bin/env python3
"""gemm_mid.hip vs gemm_skinny.hip at decode shapes (run on the GPU box)."""
import sys, torch
sys.path.insert(0, '.')
from dualhyp_amd import ops, _lib
from tools.tune_decode_common import bench, L, D
lib = _lib.load()
d, I = 2048, 5632
W1 = [torch.randn(I, d, device=D).bfloat16() * 0.02 for _ in range(L)]
W2 = [torch.randn(I, d, device=D).bfloat16() * 0.02 for _ in range(L)]
Wl = [torch.randn(32000, d, device=D).bfloat16() * 0.02 for _ in range(3)]
sc = (1 + 0.1 * torch.randn(32000, device=D)).bfloat16(); bi = (0.1 * torch.randn(32000, device=D)).bfloat16()
for M in (32, 64, 128, 256):
    x = torch.randn(M, d, device=D).bfloat16()
    ref_s = ref_h = None
    if M <= 32:
        lib.dh_set_tuning(3, 0); lib.dh_set_tuning(4, 0)
        ref_s = ops.linear(x, W1[0], epilogue=ops.EPI_SWIGLU, w2=W2[0]).float()
        ref_h = ops.linear(x, Wl[0], epilogue=ops.EPI_ADAPTER, scale=sc, bias=bi).float()
        t = bench(lambda i: ops.linear(x, W1[i % L], epilogue=ops.EPI_SWIGLU, w2=W2[i % L]))
        print(f"M={M} swiglu skinny: {t:6.1f} us")
        t = bench(lambda i: ops.linear(x, Wl[i % 3], epilogue=ops.EPI_ADAPTER, scale=sc, bias=bi))
        print(f"M={M} lm_head skinny: {t:6.1f} us")
    lib.dh_set_tuning(3, 1); lib.dh_set_tuning(4, 2)   # decode-phase kernels whatever M
    ys = ops.linear(x, W1[0], epilogue=ops.EPI_SWIGLU, w2=W2[0]).float()
    yh = ops.linear(x, Wl[0], epilogue=ops.EPI_ADAPTER, scale=sc, bias=bi).float()
    # fp32 torch reference with the same rounding points
    xf = x.float()
    g = (xf @ W1[0].float().T).bfloat16().float(); u = (xf @ W2[0].float().T).bfloat16().float()
    rs = (torch.nn.functional.silu(g).bfloat16().float() * u).bfloat16().float()
    rh = (sc.float() * ((xf @ Wl[0].float().T).bfloat16().float() + bi.float()).bfloat16().float()).bfloat16().float()
    print(f"M={M} mid swiglu max|d| vs torch {float((ys - rs).abs().max()):.3g} (rms {float(rs.pow(2).mean().sqrt()):.3g}), "
          f"head {float((yh - rh).abs().max()):.3g} (rms {float(rh.pow(2).mean().sqrt()):.3g})")
    if ref_s is not None:
        print(f"   vs skinny: swiglu differ {float((ys != ref_s).float().mean()):.2e}, head differ {float((yh != ref_h).float().mean()):.2e}")
    # rows of a larger call equal the same rows computed alone (batch invariance)
    if M > 32:
        y1 = ops.linear(x[32:64].contiguous(), W1[0], epilogue=ops.EPI_SWIGLU, w2=W2[0]).float()
        print("   rows 32..63 alone == in batch:", bool((y1 == ys[32:64]).all()))
    t = bench(lambda i: ops.linear(x, W1[i % L], epilogue=ops.EPI_SWIGLU, w2=W2[i % L]))
    print(f"M={M} swiglu mid: {t:6.1f} us")
    t = bench(lambda i: ops.linear(x, Wl[i % 3], epilogue=ops.EPI_ADAPTER, scale=sc, bias=bi))
    print(f"M={M} lm_head mid: {t:6.1f} us")
