#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/ft_prof
rocprofv3 --kernel-trace --stats -d gpurun_out/ft_prof -o ft --output-format csv -- python3 bench.py --config finetune-tinyllama --steps 4 --warmup 1 > gpurun_out/ft_prof_bench.json 2> gpurun_out/ft_prof_bench.err
find gpurun_out/ft_prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/ft_kernel_stats.csv \;
find gpurun_out/ft_prof -type f ! -name "*kernel_stats.csv" -delete
