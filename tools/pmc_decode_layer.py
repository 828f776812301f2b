#!/usr/bin/env python3
"""One TinyLlama decode layer at 640 rows (or argv[1]) through the C ABI, kernel by kernel as the engine runs it (pair-sum
GEMMs, fused attention over ~544 cached keys, finish_norm, SwiGLU), three passes on rotating weights and caches: workload
for `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace` (fabric bytes per launch against the algorithmic bytes)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from dualhyp_amd import ops, _lib
from dualhyp_amd.gpt import build_rope_cache
lib = _lib.load()
D, L = "cuda:0", 3
M = int(sys.argv[1]) if len(sys.argv) > 1 else 640
d, I, H, G, hs, S, s_max = 2048, 5632, 32, 4, 64, 544, 576
g = torch.Generator(device=D).manual_seed(0)
rn = lambda *s: (torch.randn(*s, device=D, generator=g) * 0.05).bfloat16()
Wq, A48, Bq = [rn(2560, d) for _ in range(L)], [rn(48, d) for _ in range(L)], [rn(2560, 16) for _ in range(L)]
Wp, A16, Bp = [rn(d, d) for _ in range(L)], [rn(16, d) for _ in range(L)], [rn(d, 16) for _ in range(L)]
W1, W2, Wm = [rn(I, d) for _ in range(L)], [rn(I, d) for _ in range(L)], [rn(d, I) for _ in range(L)]
kc = [rn(M, G, s_max, hs) for _ in range(L)]
vt = [rn(M, G, hs, s_max) for _ in range(L)]
cos, sin = build_rope_cache(s_max, hs, device=D)
x, xr, wn = rn(M, d), rn(M, d), rn(d)
slot = torch.arange(M, dtype=torch.int32, device=D)
kvl = torch.full((M,), S, dtype=torch.int32, device=D)
for it in range(3):
    i = it % L
    q32 = ops.linear_partial_pairs(x, Wq[i], A48[i], ksplit=8)
    att = ops.attn_decode_fused(q32, 2560, Bq[i], 1.0, (2048, 2304), cos, sin, slot, kvl, kc[i], vt[i], H, pairs=False)
    p32 = ops.linear_partial_pairs(att, Wp[i], A16[i], ksplit=8)
    x1, n2 = ops.finish_norm(p32, d, xr, wn, 1e-5, lora_b=Bp[i], lora_scale=1.0, pairs=False)
    lib.dh_set_tuning(4, 2)                      # decode-phase kernels for the fused-epilogue GEMM
    act = ops.linear(n2, W1[i], epilogue=ops.EPI_SWIGLU, w2=W2[i])
    lib.dh_set_tuning(4, 0)
    m32 = ops.linear_partial_pairs(act, Wm[i], None, ksplit=11)
    ops.finish_norm(m32, d, x1, wn, 1e-5, pairs=False)
torch.cuda.synchronize()
print("done")
