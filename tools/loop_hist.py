#!/usr/bin/env python3
"""Instruction histogram of the basic block with the most MFMAs of one kernel in a --save-temps .s file:
tools/loop_hist.py file.s <substring of the kernel symbol>.  Catches register-allocator shuffles (v_accvgpr_*, v_mov) in a main loop."""
import re, sys, collections
txt = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(txt) if l.startswith("_Z") and sys.argv[2] in l.split(":")[0])
end = next(i for i in range(start, len(txt)) if "s_endpgm" in txt[i])
lines = txt[start:end]
labels = [i for i, l in enumerate(lines) if re.match(r"^\.LBB\d+_\d+:", l)] + [len(lines)]
want = sys.argv[3] if len(sys.argv) > 3 else "v_mfma"        # the block must also hold this (e.g. global_load_lds: the steady-state loop)
a, b = max(zip(labels, labels[1:]), key=lambda ab: sum("v_mfma" in l for l in lines[ab[0]:ab[1]]) * any(want in l for l in lines[ab[0]:ab[1]]))
c = collections.Counter(l.split()[0] for l in lines[a:b] if l.strip() and l[0] in "\t " and not l.strip().startswith(";") and not l.strip().startswith("."))
print(lines[0].split(":")[0][:80], "block lines", b - a, dict(c.most_common(12)))
