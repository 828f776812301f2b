#!/usr/bin/env python3
"""Same-box A/B of the persistent attn-proj LoRA GEMM (dh_set_tuning(30, 1 | 0)): bit equality and us per launch, alternating.  GPU box."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from dualhyp_amd import ops, _lib
lib = _lib.load()
D = "cuda:0"
M, d = 2 * 32 * 512, 2048
g = torch.Generator(device=D).manual_seed(0)
rn = lambda *s: (torch.randn(*s, device=D, generator=g) * 0.05).bfloat16()
x, Wp, A16, Bp, res = rn(M, d), rn(d, d), rn(16, d), rn(d, 16), rn(M, d)
I = 5632
act, Wm = rn(M, I), rn(d, I)
for name, fn in (("attn proj + LoRA + resid", lambda: ops.linear_lora(x, Wp, A16, Bp, lora_scale=1.0, resid=res)),
                 ("mlp proj + resid", lambda: ops.linear(act, Wm, resid=res))):
    outs = {}
    for rnd in range(3):
        for knob in (0, 1):
            lib.dh_set_tuning(30, knob)
            for _ in range(3): y = fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30): y = fn()
            e1.record(); torch.cuda.synchronize()
            outs[knob] = y
            print(f"{name}: round {rnd} persist {knob}: {e0.elapsed_time(e1) / 30 * 1e3:7.1f} us", flush=True)
    print("bit-equal:", torch.equal(outs[0], outs[1]))
for M3 in (8115, 700):      # ragged rows, plain + residual
    a3, r3 = rn(M3, I), rn(M3, d)
    ys = []
    for knob in (0, 1):
        lib.dh_set_tuning(30, knob)
        ys.append(ops.linear(a3, Wm, resid=r3))
    print(f"plain + resid M {M3}: bit-equal", torch.equal(ys[0], ys[1]))
# ragged M and a second scale
for M2, sc in ((8115, 2.0), (700, 1.0)):
    x2, r2 = rn(M2, d), rn(M2, d)
    ys = []
    for knob in (0, 1):
        lib.dh_set_tuning(30, knob)
        ys.append(ops.linear_lora(x2, Wp, A16, Bp, lora_scale=sc, resid=r2))
    print(f"M {M2} scale {sc}: bit-equal", torch.equal(ys[0], ys[1]))
lib.dh_set_tuning(30, 0)
