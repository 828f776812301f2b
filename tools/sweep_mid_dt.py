import sys, torch
sys.path.insert(0, '/root/repo')
from dualhyp_amd import ops, _lib
from tools.tune_decode_common import bench, L, D
lib = _lib.load()
d, I = 2048, 5632
W1 = [torch.randn(I, d, device=D).bfloat16() * 0.02 for _ in range(L)]
W2 = [torch.randn(I, d, device=D).bfloat16() * 0.02 for _ in range(L)]
Wl = [torch.randn(32000, d, device=D).bfloat16() * 0.02 for _ in range(3)]
sc = (1 + 0.1 * torch.randn(32000, device=D)).bfloat16(); bi = (0.1 * torch.randn(32000, device=D)).bfloat16()
lib.dh_set_tuning(4, 2)
for M in (96, 128, 130, 160, 192):
    x = torch.randn(M, d, device=D).bfloat16()
    for thr in (1 << 20, 65):
        lib.dh_set_tuning(6, thr)
        t = bench(lambda i: ops.linear(x, W1[i % L], epilogue=ops.EPI_SWIGLU, w2=W2[i % L]))
        th = bench(lambda i: ops.linear(x, Wl[i % 3], epilogue=ops.EPI_ADAPTER, scale=sc, bias=bi))
        print(f"M={M:4d} {'mid' if thr > 65 else 'dt '}: swiglu {t:6.1f} us   lm_head {th:6.1f} us", flush=True)
