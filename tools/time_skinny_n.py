#!/usr/bin/env python3
"""us per call of dh_linear_bf16 at the fine-tune's LoRA down-projection shapes: gemm_skinny_n_kernel against the 128-tile kernel
(dh_set_tuning(31, 1 | 0)).  GPU box."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from dualhyp_amd import ops, _lib
lib = _lib.load()
D = "cuda:0"
g = torch.Generator(device=D).manual_seed(0)
rn = lambda *s: (torch.randn(*s, device=D, generator=g) * 0.05).bfloat16()
for M, N, K in ((17920, 48, 2048), (17920, 16, 2048), (17920, 64, 2560), (32768, 48, 2048)):
    x, w = rn(M, K), rn(N, K)
    line = [f"M {M} N {N} K {K}:"]
    for knob in (0, 1, 0, 1):
        lib.dh_set_tuning(31, knob)
        for _ in range(3): ops.linear(x, w)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): ops.linear(x, w)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 30 * 1e3
        line.append(f"{'new' if knob else 'old'} {us:6.1f} us ({M * K * 2 / us / 1e6:4.2f} TB/s)")
    print("  ".join(line), flush=True)
lib.dh_set_tuning(31, 1)
# K = 64 up-projections (gemm_k64_kernel), with and without the multiplier
for M, N in ((17920, 2048),):
    x, w, mk = rn(M, 64), rn(N, 64), (torch.rand(M, N, device=D, generator=g) > 0.05).to(torch.bfloat16)
    for name, fn in (("plain", lambda: ops.linear(x, w)), ("mul", lambda: ops.linear_mul(x, w, mk))):
        line = [f"K 64 M {M} N {N} {name}:"]
        for knob in (0, 1, 0, 1):
            lib.dh_set_tuning(31, knob)
            for _ in range(3): fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30): fn()
            e1.record(); torch.cuda.synchronize()
            line.append(f"{'new' if knob else 'old'} {e0.elapsed_time(e1) / 30 * 1e3:6.1f} us")
        print("  ".join(line), flush=True)
lib.dh_set_tuning(31, 1)
