#!/bin/bash
# Re-link the same-box A/B libraries against the CURRENT objects of the regular build (build/obj): round 3's gemm256.hip
# (tools/bin/lib_r03gemm.so) and the previous gemm_dt.hip (tools/bin/lib_olddt.so) compiled into build/variants/ by hand
# (see MEASUREMENTS.md, round 4).  Run after `python __graft_entry__.py` whenever the C ABI grew.
set -e
cd "$(dirname "$0")/.."
[ -f build/variants/gemm256_r03.o ] && hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/lib_r03gemm.so build/variants/gemm256_r03.o build/variants/stub_fast_epi.o $(ls build/obj/*.o | grep -v gemm256.o)
[ -f build/variants/gemm_dt_old.o ] && hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/lib_olddt.so build/variants/gemm_dt_old.o $(ls build/obj/*.o | grep -v gemm_dt.o)
echo relinked
