import sys, torch
sys.path.insert(0, '.')
from dualhyp_amd import ops, _lib
from tools.tune_decode_common import bench, L, D
lib = _lib.load()
d, I = 2048, 5632
W1 = [torch.randn(I, d, device=D).bfloat16() * 0.02 for _ in range(L)]
W2 = [torch.randn(I, d, device=D).bfloat16() * 0.02 for _ in range(L)]
Wq = [torch.randn(2560, d, device=D).bfloat16() * 0.02 for _ in range(L)]
A48 = [torch.randn(48, d, device=D).bfloat16() * 0.02 for _ in range(L)]
Wm = [torch.randn(d, I, device=D).bfloat16() * 0.02 for _ in range(L)]
lib.dh_set_tuning(4, 2)
for M in (256, 512, 1024, 2048):
    x = torch.randn(M, d, device=D).bfloat16(); xa = torch.randn(M, I, device=D).bfloat16()
    for st in (4, 2):
        lib.dh_set_tuning(8, st)
        t = bench(lambda i: ops.linear(x, W1[i % L], epilogue=ops.EPI_SWIGLU, w2=W2[i % L])); print(f"M={M} swiglu tiled stages={st}: {t:6.1f} us  {2*M*d*2*I/t/1e6:6.0f} TF")
        lib.dh_set_tuning(7, 65)
        t = bench(lambda i: ops.linear_chain(x, Wq[i % L], A48[i % L], ksplit=8)); print(f"M={M} qkv' chain stages={st}: {t:6.1f} us")
        t = bench(lambda i: ops.linear_chain(xa, Wm[i % L], None, ksplit=11)); print(f"M={M} mlp' chain stages={st}: {t:6.1f} us")
        lib.dh_set_tuning(7, 1 << 30)
    t = bench(lambda i: ops.linear_partial(x, Wq[i % L], A48[i % L], ksplit=8)); print(f"M={M} qkv' rows partial: {t:6.1f} us")
    t = bench(lambda i: ops.linear_partial(xa, Wm[i % L], None, ksplit=11)); print(f"M={M} mlp' rows partial: {t:6.1f} us")
