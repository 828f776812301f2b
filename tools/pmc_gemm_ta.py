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
                acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
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
torch.cuda.synchronize()
print("done")
