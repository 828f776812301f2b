#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes over tools/pmc_fp8.py (the tiled fp8 GEMM at the Llama-3-8B prefill shapes):
    python tools/pmc_fp8_summary.py out.json pass1.db pass2.db ...
Per kernel instantiation: mean of every collected counter per launch, plus derived figures —
  mfma_busy_per_simd_cycle = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)
  wave-cycle shares (waiting at s_waitcnt / barriers, waiting to issue, issuing) = SQ_WAIT_ANY, SQ_WAIT_INST_ANY,
  SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES; fabric bytes = FETCH_SIZE x 2 (gfx950, MI355X_MICROARCH.md) + WRITE_SIZE (KiB)."""
import json
import re
import sqlite3
import sys


def main():
    out_path, dbs = sys.argv[1], sys.argv[2:]
    acc = {}
    for path in dbs:
        db = sqlite3.connect(path)
        rows = db.execute("select kernel_name, dispatch_id, counter_name, sum(value) from counters_collection "
                          "group by kernel_name, dispatch_id, counter_name").fetchall()
        for name, _, counter, v in rows:
            m = re.search(r"(gemm_fp8_kernel<[^>]*>)", name)
            if m:
                acc.setdefault(m.group(1), {}).setdefault(counter, []).append(float(v))
    kernels = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}
    derived = {}
    for k, c in kernels.items():
        d = {}
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
            d["mfma_busy_per_simd_cycle"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024)
        if "SQ_WAVE_CYCLES" in c:
            for src, dst in (("SQ_WAIT_ANY", "wave_cycles_waiting"), ("SQ_WAIT_INST_ANY", "wave_cycles_waiting_to_issue"),
                             ("SQ_ACTIVE_INST_ANY", "wave_cycles_issuing")):
                if src in c:
                    d[dst] = c[src] / c["SQ_WAVE_CYCLES"]
        if "FETCH_SIZE" in c:
            d["fabric_read_bytes"] = c["FETCH_SIZE"] * 1024 * 2
        if "WRITE_SIZE" in c:
            d["fabric_write_bytes"] = c["WRITE_SIZE"] * 1024
        derived[k] = d
    res = {"source": "rocprofv3 --pmc <counters of one block per pass> --kernel-trace -- python tools/pmc_fp8.py (M = 49152: SwiGLU N 14336 K 4096, "
                     "mlp proj N 4096 K 14336, qkv N 6144 K 4096; three launches each), mean per launch; tools/pmc_fp8_summary.py",
           "kernels": kernels, "derived": derived}
    with open(out_path, "w") as fh:
        json.dump(res, fh, indent=1)
    for k, d in derived.items():
        print(k, {a: (round(b, 4) if b < 10 else f"{b / 1e9:.3f} GB") for a, b in d.items()})


if __name__ == "__main__":
    main()
