import sys, torch
sys.path.insert(0, '.')
from dualhyp_amd import ops, _lib
D = "cuda:0"
M, N = 16384, 2048
def bench(fn, n=12):
    for i in range(3): fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n): fn(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for K in (512, 1024, 2048, 4096, 8192):
    x = (torch.randn(M, K, device=D) * 0.5).bfloat16()
    W = [(torch.randn(N, K, device=D) * 0.02).bfloat16() for _ in range(4)]
    y = torch.empty(M, N, device=D, dtype=torch.bfloat16)
    t = bench(lambda i: ops.linear(x, W[i % 4], out=y))
    print(f"K={K}: {t:7.1f} us per launch (2 tile rounds) -> {t/2:6.1f} us per tile, {2*M*N*K/t/1e6:6.0f} TF")
