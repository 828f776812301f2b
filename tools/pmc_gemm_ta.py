#!/usr/bin/env python3
"""Workload for the texture-addresser / L1 PMC passes: the mlp-proj shape (16384 x 2048 x 5632) three times each through
torch.matmul (the vendor library's 256 x 256 x 64 kernel), this repo's 8-wave kernel and its 4-wave kernel.
`--summary DIR` prints the per-kernel means of every counter found in DIR's *counter_collection.csv files.  GPU box."""
import sys, glob, csv, collections
from pathlib import Path
if len(sys.argv) > 2 and sys.argv[1] == "--summary":
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(sys.argv[2] + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "gemm_nt256" in k or "Cijk" in k:
                acc[k[:75]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(k)
        for c, v in sorted(d.items()):
            print(f"    {c:42s} {sum(v) / len(v):16.1f}   (n={len(v)})")
    sys.exit(0)
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from dualhyp_amd import ops, _lib
lib = _lib.load()
D = "cuda:0"
g = torch.Generator(device=D).manual_seed(0)
rn = lambda *s: (torch.randn(*s, device=D, generator=g) * 0.05).bfloat16()
M, N, K = 16384, 2048, 5632
x, w = rn(M, K), rn(N, K)
y = torch.empty(M, N, device=D, dtype=torch.bfloat16)
for _ in range(3):
    torch.matmul(x, w.t(), out=y)
for v in (2, 5):   # 8-wave ping-pong, 4-wave full-line
    lib.dh_set_tuning(1, v)
    for _ in range(3):
        ops.linear(x, w, out=y)
if "--epilogues" in sys.argv:   # the four prefill launches of a layer with their real epilogues (4-wave kernel)
    M, d, I = 32768, 2048, 5632
    x, act = rn(M, d), rn(M, I)
    Wq, Wp, W1, W2, Wm = rn(2560, d), rn(d, d), rn(I, d), rn(I, d), rn(d, I)
    xa48, xa16, Bq, Bp = rn(M, 48), rn(M, 16), rn(2560, 16), rn(d, 16)
    H, G, hs, S = 32, 4, 64, 512
    cos, sin = rn(S, hs), rn(S, hs)
    nseq = M // S
    kc = torch.zeros(nseq, G, S, hs, device=D, dtype=torch.bfloat16); vt = torch.zeros(nseq, G, hs, S, device=D, dtype=torch.bfloat16)
    tok_slot = torch.arange(nseq, device=D, dtype=torch.int32).repeat_interleave(S)
    tok_pos = torch.arange(S, device=D, dtype=torch.int32).repeat(nseq)
    for _ in range(3):
        ops.linear(x, W1, epilogue=ops.EPI_SWIGLU, w2=W2)
        ops.linear(act, Wm, resid=x)
        ops.linear_qkv_rope_cache(x, Wq, cos, sin, tok_slot, tok_pos, kc, vt, H, G, xa=xa48, lora_b=Bq)
        ops.linear(x, Wp, epilogue=ops.EPI_LORA, xa=xa16, lora_b=Bp, lora_scale=1.0, splits=(d, d), resid=x)
torch.cuda.synchronize()
print("done")
