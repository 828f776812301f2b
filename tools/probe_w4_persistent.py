#!/usr/bin/env python3
"""Per-TILE timeline of the 4-wave GEMM's persistent blocks from the -DDH_G256_STAMPS build (DUALHYP_HIP_LIB=tools/bin/lib_stamps.so):
K loop, epilogue, and the gap from a tile's last store issued to the next tile's K loop start (the wait for the next stages, the
barrier), for the attn-proj LoRA GEMM, the mlp-proj GEMM and SwiGLU at the bench's shape.  GPU box."""
import ctypes, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from dualhyp_amd import ops, _lib
lib = _lib.load()
raw = ctypes.CDLL(str(_lib.LIB_PATH))
D = "cuda:0"
M, d, I = 2 * 32 * 512, 2048, 5632
g = torch.Generator(device=D).manual_seed(0)
rn = lambda *s: (torch.randn(*s, device=D, generator=g) * 0.05).bfloat16()
x, act, res = rn(M, d), rn(M, I), rn(M, d)
W1, W2, Wm, Wp = rn(I, d), rn(I, d), rn(d, I), rn(d, d)
A16, Bp = rn(16, d), rn(d, 16)
def stamps(n):
    buf = np.zeros(8192 * 16, dtype=np.uint64)
    assert raw.dh_debug_g256_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    return buf.reshape(8192, 16).astype(np.int64)[:n]
cases = (("attn proj + LoRA + residual (persistent, late hook)", lambda: ops.linear_lora(x, Wp, A16, Bp, lora_scale=1.0, resid=res), 128 * 8),
         ("mlp proj + residual (persistent, late hook)", lambda: ops.linear(act, Wm, resid=res), 128 * 8),
         ("SwiGLU (persistent, early requests)", lambda: ops.linear(x, W1, epilogue=ops.EPI_SWIGLU, w2=W2), 128 * 44))
for nm, fn, ntile in cases:
    for _ in range(3): fn()
    torch.cuda.synchronize()
    st = stamps(ntile)
    loop, epi = (st[:, 2] - st[:, 1]) * 0.01, (st[:, 3] - st[:, 2]) * 0.01
    gap = (st[256:, 1] - st[:-256, 3]) * 0.01               # tile vb + 256 follows tile vb on the same block
    span = (st[:, 3].max() - st[:256, 1].min()) * 0.01
    print(f"{nm}: {ntile} tiles, span from the first K loop {span:.1f} us")
    for name, v in (("K loop", loop), ("epilogue (to last store issued)", epi), ("gap to the next tile's K loop", gap)):
        print(f"   {name:34s} median {np.median(v):6.2f}  p10 {np.percentile(v, 10):6.2f}  p90 {np.percentile(v, 90):6.2f} us")
    per = np.median(loop) + np.median(epi) + np.median(gap)
    print(f"   median tile {per:.2f} us x {ntile // 256} tiles per block = {per * (ntile // 256):.1f} us")
