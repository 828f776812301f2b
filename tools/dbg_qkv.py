import math, sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from dualhyp_amd import ops, _lib
from dualhyp_amd.synth import uniform, stream_id
from oracle import ger_oracle as O
dev = torch.device("cuda:0")
U = lambda shape, b, name: uniform(shape, b, stream_id(11, name))
M, d, kv = 8192, 2048, 256
N = d + 2 * kv
x = U((M, d), 1.0, "fx").to(dev); w = U((N, d), 0.05, "fw").to(dev)
A48, B16 = U((48, d), 1 / math.sqrt(d), "fa").to(dev), U((N, 16), 0.05, "fb").to(dev)
hs, n_head, n_groups, s_max = 64, 32, 4, 512
cos, sin = O.build_rope_cache(s_max, hs); cos, sin = cos.to(dev), sin.to(dev)
i32 = torch.int32
nseq = M // s_max
slot = torch.cat([torch.full((s_max,), i, dtype=i32) for i in range(nseq)]).to(dev)
pos = torch.cat([torch.arange(s_max, dtype=i32) for _ in range(nseq)]).to(dev)
mk = lambda: (torch.zeros((nseq, n_groups, s_max, hs), dtype=torch.bfloat16, device=dev), torch.zeros((nseq, n_groups, hs, s_max), dtype=torch.bfloat16, device=dev))
lib = _lib.load()
for scale in (1.0, 2.0):
    out = {}
    for mode in (0, 3):
        lib.dh_set_tuning(24, mode)
        kc, vt = mk()
        q = ops.linear_qkv_lora_rope_cache(x, w, A48, B16, cos, sin, slot, pos, kc, vt, n_head, n_groups, lora_scale=scale)
        out[mode] = (q.float().cpu(), kc.float().cpu(), vt.float().cpu())
    for name, a, b in zip(("q", "k", "vT"), out[0], out[3]):
        ne = (a != b)
        print(f"scale {scale} {name}: mismatches {int(ne.sum())} of {ne.numel()}")
        if ne.any() and name == "q":
            qq = ne.view(M, n_head, hs)
            print("  by column mod 8:", [int(qq[..., c::8].sum()) for c in range(8)])
            print("  by column // 8:", [int(qq[..., 8*c:8*c+8].sum()) for c in range(8)])
            print("  by row mod 16:", [int(qq[r::16].sum()) for r in range(16)])
            print("  by head:", [int(qq[:, h].sum()) for h in range(n_head)])
            print("  old:", a.view(M, n_head, hs)[5, 0, :16].tolist()); print("  new:", b.view(M, n_head, hs)[5, 0, :16].tolist())
