#!/usr/bin/env python3
"""Same-box A/B of the persistent fused-QKV GEMM (dh_set_tuning(25, 1 | 0)): bit equality of q and both caches, us per launch.  GPU box."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from dualhyp_amd import ops, _lib
lib = _lib.load()
D = "cuda:0"
g = torch.Generator(device=D).manual_seed(0)
rn = lambda *s: (torch.randn(*s, device=D, generator=g) * 0.05).bfloat16()
H, G, hs, S, d = 32, 4, 64, 512, 2048
Wq, A48, Bq = rn(2560, d), rn(48, d), rn(2560, 16)
cos, sin = rn(S, hs), rn(S, hs)
def setup(nseq, T):
    M = nseq * T
    x = rn(M, d)
    slot = torch.arange(nseq, device=D, dtype=torch.int32).repeat_interleave(T)
    pos = torch.arange(T, device=D, dtype=torch.int32).repeat(nseq)
    return x, slot, pos
for nseq, T in ((64, 512), (13, 397)):
    x, slot, pos = setup(nseq, T)
    outs = {}
    for rnd in range(3):
        for knob in (0, 1):
            lib.dh_set_tuning(25, knob)
            kc = torch.zeros(nseq, G, S, hs, device=D, dtype=torch.bfloat16); vt = torch.zeros(nseq, G, hs, S, device=D, dtype=torch.bfloat16)
            fn = lambda: ops.linear_qkv_lora_rope_cache(x, Wq, A48, Bq, cos, sin, slot, pos, kc, vt, H, G)
            for _ in range(3): q = fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30): q = fn()
            e1.record(); torch.cuda.synchronize()
            outs[knob] = (q.clone(), kc.clone(), vt.clone())
            print(f"{nseq} x {T}: round {rnd} persist {knob}: {e0.elapsed_time(e1) / 30 * 1e3:7.1f} us", flush=True)
    print("bit-equal q / K cache / V^T cache:", [torch.equal(a, b) for a, b in zip(outs[0], outs[1])])
lib.dh_set_tuning(25, 0)
