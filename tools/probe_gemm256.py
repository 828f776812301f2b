#!/usr/bin/env python3
"""Per-tile timeline of the 256-tile prefill GEMM from a -DDH_G256_STAMPS build (DUALHYP_HIP_LIB=tools/bin/libG.so):
launch -> first stage in LDS -> main loop done -> epilogue stores issued.  Wave 0 stamps; the stores drain after the
last stamp, so the gap to the NEXT tile's start on the same CU is part of the epilogue's real cost.  GPU box."""
import ctypes, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import os
import numpy as np, torch
from dualhyp_amd import ops, _lib
lib = _lib.load()
raw = ctypes.CDLL(str(_lib.LIB_PATH))
D = "cuda:0"
M, d, I = 2 * 32 * 512, 2048, 5632
g = torch.Generator(device=D).manual_seed(0)
rn = lambda *s: (torch.randn(*s, device=D, generator=g) * 0.05).bfloat16()
x, act = rn(M, d), rn(M, I)
W1, W2, Wm = rn(I, d), rn(I, d), rn(d, I)
def stamps(n):
    slots = int(os.environ.get("G256_SLOTS", "16"))          # 4: a library built from round 3's gemm256.hip
    buf = np.zeros(8192 * 16, dtype=np.uint64)
    assert raw.dh_debug_g256_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    out = np.zeros((8192, 16), dtype=np.int64)
    out[:, :slots] = buf[:8192 * slots].reshape(8192, slots).astype(np.int64)
    return out[:n]
lib.dh_set_tuning(1, int(os.environ.get("G256_VARIANT", "5")))      # 1: 8-wave ping-pong, 5: 4-wave full-line
lib.dh_set_tuning(22, 0)     # per-tile launches: a block = a tile
lib.dh_set_tuning(25, 0)     # ... for the fused-QKV kernel
lib.dh_set_tuning(30, 0)     # ... and the LoRA / residual kernels (tools/probe_w4_persistent.py has the persistent blocks' per-tile timeline)
H, G, hs, S = 32, 4, 64, 512
Wq, Wp = rn(2560, d), rn(d, d)
xa48, xa16, Bq, Bp = rn(M, 48), rn(M, 16), rn(2560, 16), rn(d, 16)
A48, A16 = rn(48, d), rn(16, d)
lib.dh_set_tuning(24, int(os.environ.get("G256_FAST_EPI", "3")))   # bit 0 / 1: the v_dot2 QKV / LoRA-residual epilogues (0: round-3 forms)
cos, sin = rn(S, hs), rn(S, hs)
nseq = M // S
kc = torch.zeros(nseq, G, S, hs, device=D, dtype=torch.bfloat16); vt = torch.zeros(nseq, G, hs, S, device=D, dtype=torch.bfloat16)
tok_slot = torch.arange(nseq, device=D, dtype=torch.int32).repeat_interleave(S)
tok_pos = torch.arange(S, device=D, dtype=torch.int32).repeat(nseq)
for (nm, fn, nblk) in (("SwiGLU (K 2048)", lambda: ops.linear(x, W1, epilogue=ops.EPI_SWIGLU, w2=W2), 128 * 44),
                       ("mlp proj + residual (K 5632)", lambda: ops.linear(act, Wm, resid=x), 128 * 8),
                       ("QKV + LoRA + rope + cache append", lambda: ops.linear_qkv_lora_rope_cache(x, Wq, A48, Bq, cos, sin, tok_slot, tok_pos, kc, vt, H, G), 128 * 10),
                       ("attn proj + LoRA + residual", lambda: ops.linear_lora(x, Wp, A16, Bp, lora_scale=1.0, resid=x), 128 * 8)):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    st = stamps(nblk)
    t0 = st[:, 0].min()
    seg = np.stack([st[:, 1] - st[:, 0], st[:, 2] - st[:, 1], st[:, 3] - st[:, 2], st[:, 3] - st[:, 0]], 1) * 0.01
    print(f"{nm}: {nblk} tiles, kernel span {(st[:, 3].max() - t0) * 0.01:.1f} us")
    for j, n in enumerate(["launch -> first stage landed", "main loop", "epilogue (to last store issued)", "whole tile"]):
        v = seg[:, j]
        print(f"   {n:32s} median {np.median(v):6.2f}  p10 {np.percentile(v, 10):6.2f}  p90 {np.percentile(v, 90):6.2f} us")
    # gap between consecutive tiles on the same CU cannot be read from block ids; estimate from occupancy:
    if nm.startswith("QKV") and st[:, 4].max() > 0:      # the fused-QKV epilogue's own stamp: its set-up (positions, rope rows, x.A^T rows of strip 0) issued
        inner = np.stack([st[:, 4] - st[:, 2], st[:, 3] - st[:, 4]], 1) * 0.01
        print("   fused-QKV epilogue, median us: loop end -> set-up issued, the 8 strips:", np.round(np.median(inner, 0), 2).tolist())
    busy = seg[:, 3].sum() / 256
    print(f"   sum of tile times / 256 CUs = {busy:.1f} us of the {(st[:, 3].max() - t0) * 0.01:.1f} us span")
