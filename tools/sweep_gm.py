import sys, torch
sys.path.insert(0, '.')
from dualhyp_amd import ops, _lib
D = "cuda:0"; lib = _lib.load()
M, d, I = 32768, 2048, 5632
x = (torch.randn(M, d, device=D) * 0.5).bfloat16(); act = (torch.randn(M, I, device=D) * 0.5).bfloat16()
L = 4
Wq = [(torch.randn(2560, d, device=D) * 0.02).bfloat16() for _ in range(L)]
W1 = [(torch.randn(I, d, device=D) * 0.02).bfloat16() for _ in range(L)]; W2 = [(torch.randn(I, d, device=D) * 0.02).bfloat16() for _ in range(L)]
Wm = [(torch.randn(d, I, device=D) * 0.02).bfloat16() for _ in range(L)]
res = torch.randn(M, d, device=D).bfloat16()
ya = torch.empty(M, I, device=D, dtype=torch.bfloat16); yd = torch.empty(M, d, device=D, dtype=torch.bfloat16); yq = torch.empty(M, 2560, device=D, dtype=torch.bfloat16)
def bench(fn, n=10):
    for i in range(3): fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n): fn(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3
for gm in (2, 4, 8, 16):
    lib.dh_set_tuning(5, gm)
    t1 = bench(lambda i: ops.linear(x, W1[i % L], epilogue=ops.EPI_SWIGLU, w2=W2[i % L], out=ya))
    t2 = bench(lambda i: ops.linear(act, Wm[i % L], resid=res, out=yd))
    t3 = bench(lambda i: ops.linear(x, Wq[i % L], out=yq))
    print(f"gm={gm}: swiglu {t1*1e6:7.1f} us {2*M*2*I*d/t1/1e12:6.0f} TF | mlp {t2*1e6:7.1f} us {2*M*d*I/t2/1e12:6.0f} TF | qkv {t3*1e6:7.1f} us {2*M*2560*d/t3/1e12:6.0f} TF")
