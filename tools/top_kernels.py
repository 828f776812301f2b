#!/usr/bin/env python3
"""Top kernels of a rocprofv3 `--kernel-trace --stats --output-format csv` run: tools/top_kernels.py <kernel_stats.csv> [n] -> share, calls, average us, name."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 20]:
    print("%6.2f%%  %6d x %9.1f us  %s" % (float(r["Percentage"]), int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:100]))
