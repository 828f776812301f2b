#!/usr/bin/env python3
"""Race screen for the kernels whose LDS hand-over rests on counted s_waitcnt + raw s_barrier (cdna_hip_programming.md
§5 'place reads by the vmcnt/barrier count, never by clean runs'): many launches on fresh random data, several
shapes, compared bit for bit with a structurally different kernel that computes the same chains.
  * 256-tile GEMM: ping-pong loop (variant 1), 4-stage loop (2), persistent ping-pong blocks (4) and the four-wave full-line kernel (5,
    per tile and with persistent blocks; its in-GEMM LoRA down-projection vs two launches) vs the __syncthreads double buffer (3)
  * 128-tile GEMM: 4-stage counted-wait loop vs the 2-stage __syncthreads loop
  * decode GEMMs: gemm_mid (loader waves) vs gemm_dt (tiled) rows; tiled chain vs ordered sum of K-sliced partials"""
import sys, torch
sys.path.insert(0, '.')
from dualhyp_amd import ops, _lib
lib = _lib.load()
D = "cuda:0"
g = torch.Generator(device=D).manual_seed(123)
rn = lambda *s: (torch.randn(*s, device=D, generator=g) * 0.5).bfloat16()
bad = 0
def check(name, a, b):
    global bad
    if not torch.equal(a, b):
        bad += 1
        print(f"MISMATCH {name}: {(a != b).sum().item()} elements differ")
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for (M, N, K) in [(16384, 2560, 2048), (4096, 2048, 5632), (1024, 768, 512), (777, 1024, 2048), (256, 256, 64), (2304, 5632, 2048)]:
    for it in range(REPS):
        x, w, r = rn(M, K), rn(N, K) * 0.1, rn(M, N)
        outs = []
        for v in (3, 1, 2, 4, 5, 5):
            lib.dh_set_tuning(1, v)
            lib.dh_set_tuning(22, 2 if len(outs) == 5 else 0)     # the second pass of variant 5: persistent blocks for every epilogue
            outs.append(ops.linear(x, w, resid=r))
        lib.dh_set_tuning(22, 1)
        check(f"gemm256 pingpong M={M} N={N} K={K} it={it}", outs[1], outs[0])
        check(f"gemm256 pipe M={M} N={N} K={K} it={it}", outs[2], outs[0])
        check(f"gemm256 persistent M={M} N={N} K={K} it={it}", outs[3], outs[0])
        check(f"gemm256 four waves M={M} N={N} K={K} it={it}", outs[4], outs[0])
        check(f"gemm256 four waves persistent M={M} N={N} K={K} it={it}", outs[5], outs[0])
        if N % 64 == 0 and it % 4 == 0:
            w2 = rn(N, K) * 0.1
            so = []
            for v in (3, 1, 4, 5):
                lib.dh_set_tuning(1, v)
                so.append(ops.linear(x, w, epilogue=ops.EPI_SWIGLU, w2=w2))
            check(f"gemm256 swiglu M={M} N={N} K={K} it={it}", so[1], so[0])
            check(f"gemm256 swiglu persistent M={M} N={N} K={K} it={it}", so[2], so[0])
            check(f"gemm256 swiglu four waves (persistent) M={M} N={N} K={K} it={it}", so[3], so[0])
        if N % 256 == 0 and M >= 1024 and it % 2 == 0:
            # the LoRA down-projection in the 4-wave kernel's K loop (17th DMA piece, LDS image hand-over) vs the two-launch form
            A16, B16r = rn(16, K) * 0.1, rn(N, 16) * 0.1
            two = ops.linear(x, w, epilogue=ops.EPI_LORA, xa=ops.linear(x, A16), lora_b=B16r, lora_scale=1.0, resid=r)
            check(f"gemm256 in-GEMM x.A^T M={M} N={N} K={K} it={it}", ops.linear_lora(x, w, A16, B16r, lora_scale=1.0, resid=r), two)
    print(f"gemm256 {M}x{N}x{K}: {REPS} runs done", flush=True)
lib.dh_set_tuning(1, 5)
for (M, N, K) in [(560, 2048, 2048), (560, 2560, 2048), (200, 512, 5632), (100, 128, 64)]:
    for it in range(REPS):
        x, w = rn(M, K), rn(N, K) * 0.1
        lib.dh_set_tuning(1, 0)      # always the 128-tile kernel
        lib.dh_set_tuning(9, 2); a = ops.linear(x, w)
        lib.dh_set_tuning(9, 4); b = ops.linear(x, w)
        check(f"gemm128 4-stage M={M} N={N} K={K} it={it}", b, a)
    print(f"gemm128 {M}x{N}x{K}: {REPS} runs done", flush=True)
lib.dh_set_tuning(1, 5); lib.dh_set_tuning(9, 0)
lib.dh_set_tuning(4, 2)
for (M, d, I) in [(256, 2048, 5632), (1024, 2048, 5632), (100, 512, 768)]:
    for it in range(REPS):
        x, w1, w2 = rn(M, d), rn(I, d) * 0.1, rn(I, d) * 0.1
        lib.dh_set_tuning(6, 1 << 20); a = ops.linear(x, w1, epilogue=ops.EPI_SWIGLU, w2=w2)     # gemm_mid
        for st in (2, 4):
            for wide in (-1, 1):
                lib.dh_set_tuning(6, 65); lib.dh_set_tuning(8, st); lib.dh_set_tuning(15, wide)
                b = ops.linear(x, w1, epilogue=ops.EPI_SWIGLU, w2=w2)   # gemm_dt, 4-wave / 8-wave SwiGLU tile
                check(f"mid vs dt({st}, wide {wide}) swiglu M={M} d={d} it={it}", b, a)
        lib.dh_set_tuning(15, 0)
        # <= 64 rows: W through the per-wave LDS ring (LDS-DMA, counted vmcnt) vs W straight to VGPRs
        xs = x[:48].contiguous()
        lib.dh_set_tuning(6, 1 << 20); lib.dh_set_tuning(13, 0); a48 = ops.linear(xs, w1, epilogue=ops.EPI_SWIGLU, w2=w2)
        lib.dh_set_tuning(13, 1); check(f"mid LDS ring vs VGPR M=48 d={d} it={it}", ops.linear(xs, w1, epilogue=ops.EPI_SWIGLU, w2=w2), a48)
        A = rn(48, d) * 0.1
        wq = rn(d + 512, d) * 0.1
        ks = (d // 32 + 7) // 8
        parts = ops.linear_partial(x, wq, A, ksplit=ks)
        seq = ops.combine_partials(parts, pairs=True)     # the decode family's order: adjacent pairs, then pair sums in turn
        for wn in (2, 4):
            lib.dh_set_tuning(17, wn)
            check(f"split-K pairs(wn {wn}) vs partials M={M} d={d} it={it}", ops.combine_partials(ops.linear_partial_pairs(x, wq, A, ksplit=ks), pairs=False), seq)
        lib.dh_set_tuning(17, 0)
        lib.dh_set_tuning(7, 65)
        for st in (2, 4):
            lib.dh_set_tuning(8, st)
            check(f"chain({st}) vs partials M={M} d={d} it={it}", ops.linear_chain(x, wq, A, ksplit=ks), seq)
        # the K-sliced kernel (LDS-DMA double-buffered x rounds, full-line stores): 8 vs 10 row groups per block, and
        # rows of a wide call vs the same rows alone (different round structure)
        lib.dh_set_tuning(14, 8); p8 = ops.linear_partial(x, wq, A, ksplit=ks)
        lib.dh_set_tuning(14, 10); check(f"partials 10 vs 8 groups M={M} d={d} it={it}", ops.linear_partial(x, wq, A, ksplit=ks), p8)
        lib.dh_set_tuning(14, 0)
        check(f"partials rows alone M={M} d={d} it={it}", ops.linear_partial(x[32:64].contiguous(), wq, A, ksplit=ks), parts[:, 32:64])
    print(f"decode GEMMs M={M} d={d}: {REPS} runs done", flush=True)
lib.dh_set_tuning(6, 129); lib.dh_set_tuning(7, 1 << 30); lib.dh_set_tuning(8, 0); lib.dh_set_tuning(4, 0)
# ---- round 3: the fp8 256-tile (sixteen waves, two-stage __syncthreads loop) against the 128-tile, ragged shapes
def q8(*s):
    return (torch.randn(*s, device=D, generator=g) * 0.5).to(torch.float8_e4m3fn).view(torch.uint8)
for (M, N, K) in [(1536, 6144, 4096), (777, 1024, 512), (300, 4096, 1792), (4096, 14336, 4096)]:
    for it in range(max(REPS // 4, 2)):
        x8, w8, w28 = q8(M, K), q8(N, K), q8(N, K)
        xs, ws = torch.rand(M, device=D) + 0.5, torch.rand(N, device=D) * 0.01 + 0.005
        r = rn(M, N)
        outs = {}
        for tile in (128, 256):
            lib.dh_set_tuning(19, tile)
            outs[tile] = (ops.linear_fp8(x8, xs, w8, ws, resid=r, kernel=1), ops.linear_fp8(x8, xs, w8, ws, epilogue=ops.EPI_SWIGLU, w2q=w28, w2_scale=ws, kernel=1))
        lib.dh_set_tuning(19, 0)
        check(f"fp8 256-tile vs 128-tile plain+resid M={M} N={N} K={K} it={it}", outs[256][0], outs[128][0])
        check(f"fp8 256-tile vs 128-tile swiglu M={M} N={N} K={K} it={it}", outs[256][1], outs[128][1])
    print(f"fp8 tiles {M}x{N}x{K} done", flush=True)
print("race screen:", "CLEAN" if bad == 0 else f"{bad} MISMATCHES")
sys.exit(1 if bad else 0)
