#!/bin/bash
set -e
DUALHYP_HIP_LIB=tools/bin/lib_base.so python tools/bench_sample.py > gpurun_out/ab_sample_base.txt 2>&1
python tools/bench_sample.py > gpurun_out/ab_sample_new.txt 2>&1
python -m pytest tests/test_hip_edges.py tests/test_hip_ops.py -x -q -m gpu -k "edges or sampling" > gpurun_out/ab_edges_tests.txt 2>&1 || { tail -40 gpurun_out/ab_edges_tests.txt; exit 1; }
tail -3 gpurun_out/ab_edges_tests.txt
