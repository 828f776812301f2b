#!/usr/bin/env python3
"""Workload for PMC passes over the prefill attention kernel at the bench's launch shape (64 sequences x 512 tokens, 32 heads, 4 KV
groups, head size 64, causal): five launches; also prints the HIP-event time.  `--summary DIR` prints per-kernel counter means.  GPU box."""
import sys, glob, csv, collections
from pathlib import Path
if len(sys.argv) > 2 and sys.argv[1] == "--summary":
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(sys.argv[2] + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "attn_prefill" in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(k)
        for c, v in sorted(d.items()):
            print(f"    {c:42s} {sum(v) / len(v):16.1f}   (n={len(v)})")
    sys.exit(0)
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from dualhyp_amd import ops
D = "cuda:0"
H, G, hs, S, B = 32, 4, 64, 512, 64
g = torch.Generator(device=D).manual_seed(0)
q = (torch.randn(B * S, H, hs, device=D, generator=g)).bfloat16()
kc = torch.randn(B, G, S, hs, device=D, generator=g).bfloat16()
vt = torch.randn(B, G, hs, S, device=D, generator=g).bfloat16()
i32 = torch.int32
slot = torch.arange(B, dtype=i32, device=D)
qs = torch.arange(B, dtype=i32, device=D) * S
ql = torch.full((B,), S, dtype=i32, device=D)
p0 = torch.zeros(B, dtype=i32, device=D)
for _ in range(2):
    ops.attn_prefill(q, kc, vt, slot, qs, ql, p0, S)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    ops.attn_prefill(q, kc, vt, slot, qs, ql, p0, S)
e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 5 * 1e3
fl = 4.0 * B * H * hs * S * (S + 1) / 2
print(f"attn_prefill {B} x {S} tokens: {t:.1f} us per launch, {fl / t * 1e-6:.0f} TFLOP/s (causal FLOP)")
