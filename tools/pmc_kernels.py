#!/usr/bin/env python3
"""Mean of one PMC counter per kernel from a rocprofv3 rocpd database: pmc_kernels.py results.db COUNTER [scale]."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
rows = db.execute("select kernel_name, dispatch_id, sum(value) from counters_collection where counter_name = ? "
                  "group by kernel_name, dispatch_id", (sys.argv[2],)).fetchall()
acc = {}
for name, _, v in rows:
    acc.setdefault(name.split("(")[0][:70], []).append(float(v) * scale)
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:72s} n={len(v):4d} mean={sum(v)/len(v):14.1f} min={min(v):14.1f} max={max(v):14.1f}")
