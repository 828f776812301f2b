import sys, os, torch
sys.path.insert(0, '.')
from dualhyp_amd import ops, _lib
from tools.tune_decode_common import bench, L, D
d, I = 2048, 5632
Wq = [torch.randn(2560, d, device=D).bfloat16() * 0.02 for _ in range(L)]
A48 = [torch.randn(48, d, device=D).bfloat16() * 0.02 for _ in range(L)]
Wm = [torch.randn(d, I, device=D).bfloat16() * 0.02 for _ in range(L)]
for M in (32, 256):
    x = torch.randn(M, d, device=D).bfloat16(); xa = torch.randn(M, I, device=D).bfloat16()
    t = bench(lambda i: ops.linear_partial(x, Wq[i % L], A48[i % L], ksplit=8))
    print(f"DBG={os.environ.get('DH_ROWS_DBG','0')} M={M} qkv' ks=8: {t:6.1f} us")
    t = bench(lambda i: ops.linear_partial(xa, Wm[i % L], None, ksplit=11))
    print(f"DBG={os.environ.get('DH_ROWS_DBG','0')} M={M} mlp' ks=11: {t:6.1f} us")
