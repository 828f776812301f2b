#!/usr/bin/env python3
"""Yardstick only (never the product): torch.matmul (hipBLASLt / rocBLAS) against this repo's plain GEMM on the
prefill's and the 640-row decode step's shapes, random bf16 data, HIP events over back-to-back launches.  The question it
answers: how far is the hand-written main loop from the vendor library's on the same chip, clocks and data?  Run it
under `rocprofv3 --kernel-trace --stats` to read the library's kernel names (tile shapes).  GPU box."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from dualhyp_amd import ops, _lib
lib = _lib.load()

D = "cuda:0"
g = torch.Generator(device=D).manual_seed(0)
rn = lambda *s: (torch.randn(*s, device=D, generator=g) * 0.05).bfloat16()


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us


shapes = [("QKV", 16384, 2560, 2048), ("proj", 16384, 2048, 2048), ("fc (one half of SwiGLU)", 16384, 5632, 2048),
          ("fc both halves", 16384, 11264, 2048), ("mlp proj", 16384, 2048, 5632), ("lm_head", 640, 32000, 2048),
          ("decode QKV", 640, 2560, 2048), ("decode proj", 640, 2048, 2048), ("decode fc both", 640, 11264, 2048),
          ("decode mlp proj", 640, 2048, 5632), ("square 8192", 8192, 8192, 8192)]
print(f"{'shape':28s} {'M':>6s} {'N':>6s} {'K':>6s} | {'torch us':>9s} {'TFLOP/s':>8s} | {'8-wave us':>9s} {'TFLOP/s':>8s} | {'4-wave us':>9s} {'TFLOP/s':>8s} | torch/ours torch/4-wave")
for nm, M, N, K in shapes:
    x, w = rn(M, K), rn(N, K)
    y = torch.empty(M, N, device=D, dtype=torch.bfloat16)
    reps = 30 if M >= 8192 else 200
    t_lib = timed(lambda: torch.matmul(x, w.t(), out=y), reps)
    lib.dh_set_tuning(1, 4)                    # the 8-wave kernel (the default until round 3)
    t_own = timed(lambda: ops.linear(x, w, out=y), reps)
    y_own = ops.linear(x, w)
    lib.dh_set_tuning(1, 5)                    # the 4-wave 128 x 128-per-wave full-line kernel (default)
    t_w4 = timed(lambda: ops.linear(x, w, out=y), reps)
    y_w4 = ops.linear(x, w)
    fl = 2.0 * M * N * K
    print(f"{nm:28s} {M:6d} {N:6d} {K:6d} | {t_lib:9.1f} {fl / t_lib * 1e-6:8.0f} | {t_own:9.1f} {fl / t_own * 1e-6:8.0f} | {t_w4:9.1f} {fl / t_w4 * 1e-6:8.0f} | "
          f"{t_lib / t_own:5.2f}x {t_lib / t_w4:5.2f}x  bit-equal {bool(torch.equal(y_own, y_w4))}", flush=True)
    ref = torch.matmul(x, w.t()).float()
    for yy in (y_own, y_w4):
        err = (yy.float() - ref).abs().max().item()
        assert err <= 2e-2 * ref.abs().max().item() + 1e-3, (nm, err)
