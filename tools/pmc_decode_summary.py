#!/usr/bin/env python3
"""Fabric bytes per launch of the decode-layer kernels from two rocprofv3 --pmc passes over tools/pmc_decode_layer.py:
    python tools/pmc_decode_summary.py fetch.db write.db out.json [rows]
FETCH_SIZE (KiB) doubled for gfx950 (MI355X_MICROARCH.md) + WRITE_SIZE (KiB); algorithmic bytes per launch alongside."""
import json
import re
import sqlite3
import sys

M = int(sys.argv[4]) if len(sys.argv) > 4 else 640
d, I, S = 2048, 5632, 544
ALG = [   # (regex on the kernel name, what, algorithmic bytes: operands read once + outputs written once)
    (r"gemm_dt_kernel<4", "pair-sum GEMMs (QKV', proj', mlp': mean)", None),
    (r"attn_decode_fused_kernel", "fused decode attention (KV prefix + pair sums + new K/V)", M * 4 * 2 * S * 64 * 2 + 4 * M * 2608 * 4 + M * 2048 * 2),
    (r"finish_norm_kernel", "finish_norm (pair sums + residual -> x, xn; mean of proj' / mlp')", (4 * M * 2064 * 4 + 6 * M * 2048 * 4) // 2 + 3 * M * d * 2),
    (r"gemm_dt_kernel<1", "SwiGLU (fc_1 / fc_2)", 2 * (M * d + 2 * I * d + M * I)),
]
PAIRS = (2 * (M * d + 2608 * d) + 4 * M * 2608 * 4 + 2 * (M * d + 2064 * d) + 4 * M * 2064 * 4 + 2 * (M * I + d * I) + 6 * M * d * 4) // 3


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, dispatch_id, sum(value) from counters_collection where counter_name = ? "
                      "group by kernel_name, dispatch_id", (counter,)).fetchall()
    out = {}
    for name, _, v in rows:
        for rx, what, alg in ALG:
            if re.search(rx, name):
                out.setdefault(rx, []).append(float(v))
    return {k: sum(v) / len(v) for k, v in out.items()}


def main():
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    ks = []
    for rx, what, alg in ALG:
        if rx not in fetch:
            continue
        alg = PAIRS if alg is None else alg
        f, w = fetch[rx] * 1024 * 2, write.get(rx, 0.0) * 1024
        ks.append({"kernel": rx, "what": what, "fetch_bytes_corrected": f, "write_bytes": w, "traffic_bytes": f + w, "algorithmic_bytes": alg,
                   "ratio": (f + w) / alg})
        print(f"{what:70s} traffic {(f + w) / 1e6:8.1f} MB  algorithmic {alg / 1e6:8.1f} MB  x{(f + w) / alg:.2f}")
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- python tools/pmc_decode_layer.py; FETCH_SIZE doubled "
                         "(gfx950); L2-miss traffic on the fabric, Infinity-Cache hits included", "rows": M, "kernels": ks}, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
