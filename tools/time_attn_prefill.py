#!/usr/bin/env python3
"""us per launch of the prefill attention at the bench's shape (64 sequences x 512 tokens, 32 heads / 4 groups, hs 64) for the library
DUALHYP_HIP_LIB selects: same-box A/B of attention.hip changes.  GPU box."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from dualhyp_amd import ops
D = "cuda:0"
B, T, H, G, hs = 64, 512, 32, 4, 64
g = torch.Generator(device=D).manual_seed(0)
rn = lambda *s: (torch.randn(*s, device=D, generator=g) * 0.5).bfloat16()
qkv = rn(B * T, (H + 2 * G) * hs)
cos, sin = rn(T, hs), rn(T, hs)
i32 = torch.int32
slot = torch.arange(B, dtype=i32, device=D).repeat_interleave(T)
pos = torch.arange(T, dtype=i32, device=D).repeat(B)
kc = torch.zeros(B, G, T, hs, device=D, dtype=torch.bfloat16); vt = torch.zeros(B, G, hs, T, device=D, dtype=torch.bfloat16)
q = ops.qkv_rope_cache(qkv, cos, sin, slot, pos, kc, vt, H, G)
seq = torch.arange(B, dtype=i32, device=D); qs = seq * T; ql = torch.full((B,), T, dtype=i32, device=D); z = torch.zeros(B, dtype=i32, device=D)
for _ in range(3): y = ops.attn_prefill(q, kc, vt, seq, qs, ql, z, T)
for rnd in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): y = ops.attn_prefill(q, kc, vt, seq, qs, ql, z, T)
    e1.record(); torch.cuda.synchronize()
    print(f"round {rnd}: {e0.elapsed_time(e1) / 30 * 1e3:7.1f} us   checksum {y.float().abs().sum().item():.6e}", flush=True)
