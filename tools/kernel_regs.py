#!/usr/bin/env python3
"""Registers / spills / LDS per kernel from a hipcc --save-temps .s file (amdhsa metadata)."""
import re, sys
txt = open(sys.argv[1]).read()
md = txt[txt.index("amdhsa.kernels:"):]
for blk in md.split("  - .agpr_count:")[1:]:
    blk = ".agpr_count:" + blk
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = g("name")
    if len(sys.argv) > 2 and sys.argv[2] not in name:
        continue
    print(f"{name[:70]:70s} agpr {g('agpr_count'):>4s} vgpr {g('vgpr_count'):>4s} sgpr {g('sgpr_count'):>4s} spill {g('vgpr_spill_count'):>4s} scratch {g('private_segment_fixed_size'):>5s} lds {g('group_segment_fixed_size')}")
