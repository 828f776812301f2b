#!/usr/bin/env python3
"""HBM-side traffic per prefill GEMM launch from two rocprofv3 --pmc passes over tools/pmc_gemm.py:
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmcF -o f -- python tools/pmc_gemm.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmcW -o w -- python tools/pmc_gemm.py
    python tools/pmc_summary.py gpurun_out/pmcF/f_results.db gpurun_out/pmcW/w_results.db profiles/r02_pmc_gemm.json
FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (128-byte
requests are tallied at 64 B).  Algorithmic bytes = operands read once + output written once (+ residual / LoRA reads)."""
import json
import re
import sqlite3
import sys

M, d, I = 2 * 32 * 512, 2048, 5632
ALG = {   # kernel template args -> (what, algorithmic bytes per launch)
    # keyed by (epilogue, residual): the third template argument is the loop variant (8-wave kernel: 2 per-tile, 3 persistent blocks;
    # 4-wave kernel: persistent blocks or not)
    "<0, true": ("mlp proj + residual", 2 * (M * I + d * I + 2 * M * d)),
    "<1, false": ("qkv + LoRA", 2 * (M * d + 2560 * d + M * 2560 + M * 48 + 2560 * 16)),
    "<1, true": ("attn proj + LoRA + residual", 2 * (M * d + d * d + 2 * M * d + M * 16 + d * 16)),
    "<2, false": ("fc_1/fc_2 SwiGLU", 2 * (M * d + 2 * I * d + M * I)),
    # round 4's bench launch: the fused QKV GEMM (EPI 4) reads x, W, A, B, writes q / k / v (2560 columns per token) once
    "<4, false": ("qkv + LoRA + rope + cache append", 2 * (M * d + 2560 * d + 48 * d + 2560 * 16 + M * 2560)),
}


def per_kernel(db_path, counter):
    db = sqlite3.connect(db_path)
    rows = db.execute("select kernel_name, dispatch_id, sum(value) from counters_collection where counter_name = ? "
                      "group by kernel_name, dispatch_id", (counter,)).fetchall()
    out = {}
    for name, _, v in rows:
        m = re.search(r"gemm_nt256(?:w4)?_kernel(<\d+, (?:true|false))", name)   # the 8-wave kernel or the 4-wave one (round 3)
        if m:
            out.setdefault(m.group(1), []).append(float(v))
    return {k: sum(v) / len(v) for k, v in out.items()}


def main():
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    kernels = []
    for k, (what, alg) in ALG.items():
        if k not in fetch:
            continue
        f, w = fetch[k] * 1024 * 2, write.get(k, 0.0) * 1024
        kernels.append({"kernel": f"gemm_nt256[w4]_kernel{k}, *>", "what": what, "fetch_bytes_corrected": f, "write_bytes": w,
                        "traffic_bytes": f + w, "algorithmic_bytes": alg, "ratio": (f + w) / alg})
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- python tools/pmc_gemm.py; "
                     "FETCH_SIZE (KiB) doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); counts L2 misses "
                     "served by the fabric (Infinity Cache hits included)",
           "shape": "M=32768 (2 x 32 x 512 tokens: one prefill launch of bench.py), TinyLlama layer, one launch each, mean of 3",
           "kernels": kernels,
           "traffic_bytes_per_launch_mean": sum(x["traffic_bytes"] for x in kernels) / max(len(kernels), 1),
           "algorithmic_bytes_per_launch_mean": sum(x["algorithmic_bytes"] for x in kernels) / max(len(kernels), 1)}
    with open(sys.argv[3], "w") as fh:
        json.dump(res, fh, indent=1)
    for x in kernels:
        print(f"{x['kernel']:36s} {x['what']:30s} traffic {x['traffic_bytes'] / 1e9:6.3f} GB  algorithmic {x['algorithmic_bytes'] / 1e9:6.3f} GB  x{x['ratio']:.2f}")


if __name__ == "__main__":
    main()
