#!/usr/bin/env python3
"""Does a weight-streaming decode kernel run faster when its weights were touched shortly before (Infinity-Cache
resident) than when they come from HBM?  Basis for a next-layer weight prefetch on a side stream of the decode graph.
Per kernel: [flush 1 GiB] -> [optional: read the weights with a plain torch reduction] -> event -> kernel -> event."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualhyp_amd import ops
dev = "cuda"
d, I = 2048, 5632
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.05).bfloat16()
x, xa = rn(32, d), rn(32, I)
Wq, A48, Wp, A16, W1, W2, Wm = rn(2560, d), rn(48, d), rn(d, d), rn(16, d), rn(I, d), rn(I, d), rn(d, I)
flush = torch.empty(1 << 28, dtype=torch.float32, device=dev)     # 1 GiB
cases = {
    "qkv' (10.7 MB)": (lambda: ops.linear_partial(x, Wq, A48, ksplit=8), [Wq, A48]),
    "proj' (8.4 MB)": (lambda: ops.linear_partial(x, Wp, A16, ksplit=8), [Wp, A16]),
    "swiglu (46 MB)": (lambda: ops.linear(x, W1, epilogue=ops.EPI_SWIGLU, w2=W2), [W1, W2]),
    "mlp' (23 MB)": (lambda: ops.linear_partial(xa, Wm, None, ksplit=11), [Wm]),
}
for name, (fn, ws) in cases.items():
    res = {}
    for mode in ("cold", "warm", "hot"):
        ts = []
        for it in range(30):
            if mode != "hot":
                flush.fill_(1.0)
            if mode == "warm":
                for w in ws:
                    w.view(torch.int16).sum()          # plain loads of every weight byte
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            a.record(); fn(); b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        ts.sort()
        res[mode] = ts[len(ts) // 2]
    print(f"{name:16s} cold {res['cold']:6.1f} us   prefetched {res['warm']:6.1f} us   back-to-back {res['hot']:6.1f} us", flush=True)
