import sys, torch
sys.path.insert(0, '.')
from dualhyp_amd import ops, _lib
from tools.tune_decode_common import bench, L, D
lib = _lib.load()
d, I = 2048, 5632
Wq = [torch.randn(2560, d, device=D).bfloat16() * 0.02 for _ in range(L)]
A48 = [torch.randn(48, d, device=D).bfloat16() * 0.02 for _ in range(L)]
Wm = [torch.randn(d, I, device=D).bfloat16() * 0.02 for _ in range(L)]
W1 = [torch.randn(I, d, device=D).bfloat16() * 0.02 for _ in range(L)]
W2 = [torch.randn(I, d, device=D).bfloat16() * 0.02 for _ in range(L)]
lib.dh_set_tuning(4, 2)
for M in (128, 256):
    x = torch.randn(M, d, device=D).bfloat16(); xa = torch.randn(M, I, device=D).bfloat16()
    t = bench(lambda i: ops.linear_partial(x, Wq[i % L], A48[i % L], ksplit=8)); print(f"M={M} qkv' rows partial: {t:6.1f} us")
    t = bench(lambda i: ops.linear_chain(x, Wq[i % L], A48[i % L], ksplit=8)); print(f"M={M} qkv' tiled chain : {t:6.1f} us")
    t = bench(lambda i: ops.linear_partial(xa, Wm[i % L], None, ksplit=11)); print(f"M={M} mlp' rows partial: {t:6.1f} us")
    t = bench(lambda i: ops.linear_chain(xa, Wm[i % L], None, ksplit=11)); print(f"M={M} mlp' tiled chain : {t:6.1f} us")
    for mn in (1000, 65):
        lib.dh_set_tuning(6, mn)
        t = bench(lambda i: ops.linear(x, W1[i % L], epilogue=ops.EPI_SWIGLU, w2=W2[i % L])); print(f"M={M} swiglu {'mid' if mn > M else 'tiled'}: {t:6.1f} us")
