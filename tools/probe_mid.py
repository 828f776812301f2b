#!/usr/bin/env python3
"""Where the time of the 32-row SwiGLU streaming kernel (gemm_mid.hip) goes: HBM-cold vs cache-warm weights, and
per-block 100 MHz timestamps from a -DDH_MID_STAMPS build (DUALHYP_HIP_LIB=tools/bin/libS.so).  GPU box only."""
import ctypes, os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from dualhyp_amd import ops, _lib
from tools.tune_decode_common import bench, L, D
lib = _lib.load()
d, I = 2048, 5632
W1 = [torch.randn(I, d, device=D).bfloat16() * 0.02 for _ in range(L)]
W2 = [torch.randn(I, d, device=D).bfloat16() * 0.02 for _ in range(L)]
lib.dh_set_tuning(3, 1); lib.dh_set_tuning(4, 2)
if os.environ.get('DH_WLDS'): lib.dh_set_tuning(13, int(os.environ['DH_WLDS']))
for M in (1, 16, 32, 64):
    x = torch.randn(M, d, device=D).bfloat16()
    tc = bench(lambda i: ops.linear(x, W1[i % L], epilogue=ops.EPI_SWIGLU, w2=W2[i % L]))
    tw = bench(lambda i: ops.linear(x, W1[0], epilogue=ops.EPI_SWIGLU, w2=W2[0]))
    print(f"M={M:3d} swiglu mid: cold {tc:6.1f} us ({2*I*d*2/tc/1e6:.2f} TB/s)   same weights every call {tw:6.1f} us")
for (N, K, nm) in ((2048, 5632, "mlp_proj"), (2560, 2048, "qkv"), (2048, 2048, "proj")):
    Wp = [torch.randn(N, K, device=D).bfloat16() * 0.02 for _ in range(L)]
    x = torch.randn(32, K, device=D).bfloat16()
    tc = bench(lambda i: ops.linear(x, Wp[i % L]))
    tw = bench(lambda i: ops.linear(x, Wp[0]))
    print(f"M= 32 {nm} plain mid: cold {tc:6.1f} us ({N*K*2/tc/1e6:.2f} TB/s)   warm {tw:6.1f} us")
    del Wp
raw = ctypes.CDLL(str(_lib.LIB_PATH))
if hasattr(raw, "dh_debug_mid_stamps"):
    x = torch.randn(32, d, device=D).bfloat16()
    for trial in range(3):
        for i in range(4): ops.linear(x, W1[(5 * trial + i) % L], epilogue=ops.EPI_SWIGLU, w2=W2[(5 * trial + i) % L])
        torch.cuda.synchronize()
        buf = np.zeros(1024 * 8, dtype=np.uint64)
        assert raw.dh_debug_mid_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
        st = buf.reshape(1024, 8)[:I // 32].astype(np.int64)
        t0 = st[:, 0].min()
        rel = (st - t0) * 0.01   # us
        names = ["c:start", "c:W ring issued", "c:x slice 0 landed", "c:main loop done", "l:start", "l:x0 landed", "c:end"]
        print(f"trial {trial}: kernel span {rel[:, 6].max():.2f} us over {len(st)} blocks")
        for j, nme in enumerate(names):
            v = rel[:, j]
            print(f"   {nme:22s} min {v.min():6.2f}  median {np.median(v):6.2f}  max {v.max():6.2f}")
        loop = rel[:, 3] - rel[:, 2]
        print(f"   main loop per block: min {loop.min():.2f} median {np.median(loop):.2f} max {loop.max():.2f} us")
