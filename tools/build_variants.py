#!/usr/bin/env python3
"""A/B builds of one csrc file: tools/build_variants.py gemm256.hip name1:-DX=1,-DY=2 name2:-DX=3 ... compiles the file once per
variant with the extra flags and links it with the other objects of the regular build (build/obj) into
tools/bin/lib_<name>.so; run a tool with DUALHYP_HIP_LIB=tools/bin/lib_<name>.so to use it.  Runs here (no GPU)."""
import subprocess, sys, concurrent.futures as cf
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as G
G.build()
src = G.CSRC / sys.argv[1]
out = ROOT / "tools" / "bin"
out.mkdir(exist_ok=True)
(ROOT / "build" / "variants").mkdir(parents=True, exist_ok=True)
others = [str(o) for o in sorted((ROOT / "build" / "obj").glob("*.o")) if o.stem != src.stem]
def one(spec):
    name, _, flags = spec.partition(":")
    obj = ROOT / "build" / "variants" / f"{src.stem}_{name}.o"
    r = subprocess.run([G.HIPCC, *G.FLAGS, *[f for f in flags.split(",") if f], "-c", str(src), "-o", str(obj)], capture_output=True, text=True)
    if r.returncode != 0:
        return f"{name}: FAILED\n{r.stderr[-2000:]}"
    r = subprocess.run([G.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(out / f"lib_{name}.so"), str(obj), *others], capture_output=True, text=True)
    return f"{name}: {'ok' if r.returncode == 0 else 'LINK FAILED ' + r.stderr[-500:]}"
with cf.ThreadPoolExecutor(max_workers=6) as ex:
    for line in ex.map(one, sys.argv[2:]):
        print(line)
