#!/usr/bin/env python3
"""LoRA-gradient token contraction (dh_tn_accum_f32) at the packed fine-tune's shapes: the MFMA kernel against the VALU kernel of
rounds 2-3 (dh_set_tuning(26, 1 | 0)), us per call and the error against torch fp32.  GPU box."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from dualhyp_amd import ops, _lib
lib = _lib.load()
D = "cuda:0"
T = int(sys.argv[1]) if len(sys.argv) > 1 else 32 * 560
g = torch.Generator(device=D).manual_seed(0)
rn = lambda *s: (torch.randn(*s, device=D, generator=g) * 0.5).bfloat16()
x2048, dqkv, xa48, t64, t16 = rn(T, 2048), rn(T, 2560), rn(T, 48), rn(T, 64), rn(T, 16)
cases = [("gB2  [2048,16] = dx1^T xa2", x2048, t16), ("gA2  [16,2048] = t^T yd", t16, x2048),
         ("gA1  [48,2048] = t3[:, :48]^T n1d", t64[:, :48], x2048),
         ("gB1q [2048,16] = dqkv[:, :2048]^T xa[:, :16]", dqkv[:, :2048], xa48[:, :16]),
         ("gB1k [256,16]  = dqkv[:, 2048:2304]^T xa[:, 16:32]", dqkv[:, 2048:2304], xa48[:, 16:32])]
for name, a, b in cases:
    want = a.float().T @ b.float()
    line = [f"{name:52s}"]
    for knob in (0, 1):
        lib.dh_set_tuning(26, knob)
        out = torch.empty(a.size(1), b.size(1), dtype=torch.float32, device=D)
        for _ in range(3): ops.tn_accum(a, b, out, scale=1.0, accumulate=False)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.tn_accum(a, b, out, scale=1.0, accumulate=False)
        e1.record(); torch.cuda.synchronize()
        err = ((out - want).abs().max() / want.abs().max()).item()
        line.append(f"{'mfma' if knob else 'valu'} {e0.elapsed_time(e1) * 50:7.1f} us  rel err {err:.1e}")
    print("   ".join(line), flush=True)
