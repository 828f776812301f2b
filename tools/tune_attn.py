import sys, torch
sys.path.insert(0, '.')
from dualhyp_amd import ops, _lib
exec(open('tools/tune_decode.py').read().split("d, I, M = 2048")[0])   # bench(), imports
H, G, hs, S, B = 32, 4, 64, 640, 32
kc = [torch.randn(B, G, S, hs, device=D).bfloat16() for _ in range(L)]
vt = [torch.randn(B, G, hs, S, device=D).bfloat16() for _ in range(L)]
cos = torch.randn(S, hs, device=D).bfloat16(); sin = torch.randn(S, hs, device=D).bfloat16()
Bq = torch.randn(2560, 16, device=D).bfloat16() * 0.02
i32 = torch.int32
slot = torch.arange(B, dtype=i32, device=D)
for kvlen in (1, 33, 257, 545):
    kvl = torch.full((B,), kvlen, dtype=i32, device=D)
    for ks, lora in ((1, False), (2, True), (8, True)):
        q32 = torch.randn(ks, B, 2608, device=D) * 0.1
        t = bench(lambda i: ops.attn_decode_fused(q32, 2560, Bq if lora else None, 1.0, (2048, 2304), cos, sin, slot, kvl, kc[i % L], vt[i % L], H))
        print(f"kv_len {kvlen:4d} partials {ks} lora {lora}: {t:6.1f} us")
d = 2048
xr = torch.randn(B, d, device=D).bfloat16(); wn = torch.ones(d, device=D).bfloat16(); Bp = torch.randn(d, 16, device=D).bfloat16()
for ks in (1, 2, 4, 8, 11):
    y = torch.randn(ks, B, d + 16, device=D)
    t = bench(lambda i: ops.finish_norm(y, d, xr, wn, 1e-5, lora_b=Bp, lora_scale=1.0))
    print(f"finish_norm {ks} partials: {t:5.1f} us")
