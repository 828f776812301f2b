#!/usr/bin/env python3
"""The tiled fp8 GEMM at the Llama-3-8B prefill shapes (M = 32 x 1536): 128 x 128 tiles (4 waves, two blocks per CU)
against 256 x 256 tiles (16 waves, one block per CU); prints us and PFLOP/s and checks the outputs are bit-identical."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dualhyp_amd import ops, _lib
lib = _lib.load()
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 49152
d, I = 4096, 14336
def q8(*s):
    return (torch.randn(*s, device=dev, generator=g) * 0.5).to(torch.float8_e4m3fn).view(torch.uint8)
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for name, N, K, epi in (("qkv", 6144, d, ops.EPI_PLAIN), ("proj+resid", d, d, ops.EPI_PLAIN), ("swiglu", I, d, ops.EPI_SWIGLU), ("mlp+resid", d, I, ops.EPI_PLAIN)):
    x, w = q8(M, K), q8(N, K)
    w2 = q8(N, K) if epi == ops.EPI_SWIGLU else None
    xs, ws = torch.rand(M, device=dev) + 0.5, torch.rand(N, device=dev) * 0.01 + 0.005
    resid = (torch.randn(M, N, device=dev, generator=g)).bfloat16() if "resid" in name else None
    flop = 2.0 * M * N * K * (2 if epi == ops.EPI_SWIGLU else 1)
    out = {}
    for tile in (128, 256):
        lib.dh_set_tuning(19, tile)
        for gm in (1, 2, 4, 8):                  # m-tiles per band of the tile walk
            lib.dh_set_tuning(20, gm)
            fn = lambda: ops.linear_fp8(x, xs, w, ws, epilogue=epi, w2q=w2, w2_scale=ws if w2 is not None else None, resid=resid, kernel=1)
            out[tile] = fn()
            t = timeit(fn)
            print(f"{name:11s} M={M} N={N} K={K}  tile {tile} band {gm}: {t:8.1f} us  {flop / t / 1e9:6.3f} PFLOP/s", flush=True)
    lib.dh_set_tuning(19, 0); lib.dh_set_tuning(20, 0)
    print(f"{name:11s} bit-identical across tile sizes: {torch.equal(out[128], out[256])}", flush=True)
