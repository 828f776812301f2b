#!/bin/bash
# A/B of two library builds inside one gpurun call: tools/ab_call.sh <tag> (base = tools/bin/lib_base.so, new = the in-tree build)
set -e
tag=$1
export DUALHYP_HIP_LIB=tools/bin/lib_base.so
python tools/tune_attn.py > gpurun_out/ab_${tag}_attn_base.txt 2>&1
python tools/sweep_finish.py > gpurun_out/ab_${tag}_finish_base.txt 2>&1
unset DUALHYP_HIP_LIB
python tools/tune_attn.py > gpurun_out/ab_${tag}_attn_new.txt 2>&1
python tools/sweep_finish.py > gpurun_out/ab_${tag}_finish_new.txt 2>&1
python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "fused_decode or decode_attention or rows_invariant" > gpurun_out/ab_${tag}_tests.txt 2>&1
python -m pytest tests/test_hip_model.py -x -q -m gpu -k "joint or generate_ids or full_tinyllama" >> gpurun_out/ab_${tag}_tests.txt 2>&1
DUALHYP_HIP_LIB=tools/bin/lib_base.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab_${tag}_bench_base.json 2> gpurun_out/ab_${tag}_bench_base.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab_${tag}_bench_new.json 2> gpurun_out/ab_${tag}_bench_new.err
DUALHYP_HIP_LIB=tools/bin/lib_base.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab_${tag}_bench_base2.json 2> gpurun_out/ab_${tag}_bench_base2.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab_${tag}_bench_new2.json 2> gpurun_out/ab_${tag}_bench_new2.err
tail -3 gpurun_out/ab_${tag}_tests.txt
