#!/usr/bin/env python3
"""Kernel statistics (the table `rocprofv3 --stats` prints) from a rocprofv3 rocpd database (.db):
    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [out.csv]
Columns: Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs."""
import csv
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.sub(r"\(.*$", "", name) if len(name) > 120 else name


def main() -> None:
    db = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = db.execute(f"select {name_col}, count(*), sum(end - start), min(end - start), max(end - start) from kernels group by {name_col}").fetchall()
    total = sum(r[2] for r in rows) or 1
    rows.sort(key=lambda r: -r[2])
    out = [("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")]
    for n, c, t, lo, hi in rows:
        out.append((short(n), c, t, round(t / c, 1), round(100.0 * t / total, 3), lo, hi))
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w", newline="") as f:
            csv.writer(f).writerows(out)
    for r in out[:40]:
        print("%-110s %8s %14s %12s %8s" % (str(r[0])[:110], r[1], r[2], r[3], r[4]))


if __name__ == "__main__":
    main()
