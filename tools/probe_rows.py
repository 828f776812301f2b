#!/usr/bin/env python3
"""Per-block timeline of the K-sliced partial-sum kernel (gemm_skinny_rows_kernel) from a -DDH_ROWS_STAMPS build
(DUALHYP_HIP_LIB=tools/bin/libR.so): W fetched + transposed, x round staged, MFMA + partial stores.  GPU box."""
import ctypes, os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from dualhyp_amd import ops, _lib
lib = _lib.load()
raw = ctypes.CDLL(str(_lib.LIB_PATH))
D = "cuda:0"
d, I = 2048, 5632
M = int(os.environ.get("DH_M", "640"))
for (nm, N, ext, K, ks) in (("qkv'", 2560, 48, d, 8), ("mlp'", d, 0, I, 11)):
    W = [torch.randn(N, K, device=D).bfloat16() * 0.02 for _ in range(4)]
    A = [torch.randn(ext, K, device=D).bfloat16() * 0.02 for _ in range(4)] if ext else [None] * 4
    x = torch.randn(M, K, device=D).bfloat16()
    for trial in range(2):
        for i in range(3): ops.linear_partial(x, W[(trial + i) % 4], A[(trial + i) % 4], ksplit=ks)
        torch.cuda.synchronize()
        buf = np.zeros(1024 * 8, dtype=np.uint64)
        assert raw.dh_debug_rows_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
        st = buf.reshape(1024, 8).astype(np.int64)
        st = st[st[:, 0] > 0]
        t0 = st[:, 0].min()
        rel = (st - t0) * 0.01
        names = ["start", "W transposed", "x round 0 landed (barrier)", "round 0 MFMA done, round 1 landed", "-", "last round MFMA done", "end (stores issued)"]
        print(f"{nm} M={M} trial {trial}: {len(st)} blocks, span {rel[:, 6].max():.2f} us")
        for j, n in enumerate(names):
            v = rel[:, j][st[:, j] > 0]
            if len(v): print(f"   {n:28s} min {v.min():6.2f} median {np.median(v):6.2f} max {v.max():6.2f}")
