#!/usr/bin/env python3
"""Fused decode attention (decode_fused.hip) at the Llama-3-8B head shape (32 q / 8 kv heads x 128): time vs cached keys
and rows (GPU box).  The intercept is the prologue (finish q/k/v, rope, cache append), the slope the KV stream."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from dualhyp_amd import ops, _lib
from tools.tune_decode_common import bench, D
_lib.load()
L = 8
H, G, hs, S = 32, 8, 128, 640
i32 = torch.int32
cos = torch.randn(S, hs, device=D).bfloat16(); sin = torch.randn(S, hs, device=D).bfloat16()
qkv_dim = (H + 2 * G) * hs
for B in (32, 128):
    kc = [torch.randn(B, G, S, hs, device=D).bfloat16() for _ in range(L)]
    vt = [torch.randn(B, G, hs, S, device=D).bfloat16() for _ in range(L)]
    slot = torch.arange(B, dtype=i32, device=D)
    for kvlen in (1, 33, 129, 257, 545):
        kvl = torch.full((B,), kvlen, dtype=i32, device=D)
        q32 = torch.randn(1, B, qkv_dim, device=D) * 0.1
        t = bench(lambda i: ops.attn_decode_fused(q32, qkv_dim, None, 1.0, (H * hs, (H + G) * hs), cos, sin, slot, kvl, kc[i % L], vt[i % L], H))
        mb = B * G * kvlen * hs * 2 * 2 / 1e6
        print(f"rows {B:4d} kv_len {kvlen:4d}: {t:6.1f} us   ({mb:6.1f} MB of K/V, {mb / t / 1e6 * 1e6:.2f} TB/s)", flush=True)
    del kc, vt

# the same call between 218-MB weight streams over a 7-GB footprint (what a Llama-3-8B decode step does between two
# attention launches): does the attention kernel slow down when its pages / lines are cold in every cache and TLB?
B, NL = 32, 32
kc = [torch.randn(B, G, S, hs, device=D).bfloat16() for _ in range(NL)]
vt = [torch.randn(B, G, hs, S, device=D).bfloat16() for _ in range(NL)]
wbig = [torch.empty(218 * 1024 * 1024 // 2, device=D, dtype=torch.bfloat16).normal_() for _ in range(NL)]
slot = torch.arange(B, dtype=i32, device=D)
kvl = torch.full((B,), 545, dtype=i32, device=D)
q32 = torch.randn(1, B, qkv_dim, device=D) * 0.1
import tools.tune_decode_common as tdc
tdc.L = NL
t_w = tdc.bench(lambda i: wbig[i % NL].sum())
t_wa = tdc.bench(lambda i: (wbig[i % NL].sum(), ops.attn_decode_fused(q32, qkv_dim, None, 1.0, (H * hs, (H + G) * hs), cos, sin, slot, kvl, kc[i % NL], vt[i % NL], H)))
t_a = tdc.bench(lambda i: ops.attn_decode_fused(q32, qkv_dim, None, 1.0, (H * hs, (H + G) * hs), cos, sin, slot, kvl, kc[i % NL], vt[i % NL], H))
print(f"weights alone {t_w:6.1f} us, weights + attention {t_wa:6.1f} us -> attention {t_wa - t_w:6.1f} us; attention alone over {NL} caches {t_a:6.1f} us")
