#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 rocpd database (.db), grouped by (kernel, next kernel):
    python tools/rocpd_gaps.py x_results.db [min_calls]
Shows where the kernel boundaries of the decode loop cost the most (gap = next.start - this.end, same queue order)."""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)[:60]


def main() -> None:
    db = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = db.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
    pairs = {}
    for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
        gap = s1 - e0
        if gap < 0 or gap > 50000:          # overlapping streams / host-side pauses
            continue
        k = (short(n0), short(n1))
        p = pairs.setdefault(k, [0, 0, 0])
        p[0] += 1; p[1] += gap; p[2] += e0 - s0
    min_calls = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    tot = sum(p[1] for p in pairs.values())
    print(f"total gap time {tot / 1e6:.1f} ms over {sum(p[0] for p in pairs.values())} boundaries")
    for k, p in sorted(pairs.items(), key=lambda kv: -kv[1][1])[:30]:
        if p[0] >= min_calls:
            print(f"{k[0]:60s} -> {k[1]:60s} n={p[0]:6d} gap avg {p[1] / p[0] / 1e3:6.2f} us  (kernel avg {p[2] / p[0] / 1e3:7.2f} us)")


if __name__ == "__main__":
    main()
