#!/bin/bash
set -e
python -m pytest tests/test_hip_edges.py -x -q -m gpu > gpurun_out/ab_prune_tests.txt 2>&1 || { tail -40 gpurun_out/ab_prune_tests.txt; exit 1; }
tail -2 gpurun_out/ab_prune_tests.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-overlap-probe --tune 23=0 > gpurun_out/ab_prune_bench_off.json 2> gpurun_out/ab_prune_bench_off.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-overlap-probe > gpurun_out/ab_prune_bench_on.json 2> gpurun_out/ab_prune_bench_on.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-overlap-probe --tune 23=0 > gpurun_out/ab_prune_bench_off2.json 2> gpurun_out/ab_prune_bench_off2.err
python bench.py --steps 20 --warmup 5 > gpurun_out/ab_prune_bench_on2.json 2> gpurun_out/ab_prune_bench_on2.err
