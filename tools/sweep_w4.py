#!/usr/bin/env python3
"""Schedule sweep of the 4-wave 256-tile GEMM: every tools/bin/lib_<name>.so given on the command line (built by
tools/build_variants.py) runs the prefill's plain shapes through variant 5 in its own process; TFLOP/s per shape.  GPU box."""
import os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, str(ROOT))
    import torch
    from dualhyp_amd import ops, _lib
    lib = _lib.load()
    lib.dh_set_tuning(1, int(os.environ.get("W4_VARIANT", "5")))
    lib.dh_set_tuning(22, int(os.environ.get("W4_PERSIST", "1")))
    D = "cuda:0"
    g = torch.Generator(device=D).manual_seed(0)
    rn = lambda *s: (torch.randn(*s, device=D, generator=g) * 0.05).bfloat16()
    res = []
    for M, N, K in ((16384, 2560, 2048), (16384, 2048, 2048), (16384, 11264, 2048), (16384, 2048, 5632), (8192, 8192, 8192)):
        x, w = rn(M, K), rn(N, K)
        y = torch.empty(M, N, device=D, dtype=torch.bfloat16)
        for _ in range(3):
            ops.linear(x, w, out=y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            ops.linear(x, w, out=y)
        e1.record()
        torch.cuda.synchronize()
        res.append(2.0 * M * N * K / (e0.elapsed_time(e1) / 30 * 1e-3) * 1e-12)
    print(" ".join(f"{r:7.0f}" for r in res), flush=True)
    sys.exit(0)
print(f"{'variant':10s}  QKV     proj    fc-both mlp     8192^3   (TFLOP/s)")
for name in sys.argv[1:]:
    env = dict(os.environ)
    if name.startswith("old"):
        env["W4_VARIANT"] = "4"
    elif name == "default":
        pass
    elif name == "nopersist":
        env["W4_PERSIST"] = "0"
    else:
        env["DUALHYP_HIP_LIB"] = str(ROOT / "tools" / "bin" / f"lib_{name}.so")
    r = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True, timeout=300)
    print(f"{name:10s} {r.stdout.strip() if r.returncode == 0 else 'FAILED ' + r.stderr[-300:]}", flush=True)
