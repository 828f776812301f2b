#!/bin/bash
# End-of-session records (GPU box): the driver-form bench line, the same command under rocprofv3 --kernel-trace --stats, smoke().
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
rm -rf gpurun_out/final_prof
rocprofv3 --kernel-trace --stats -d gpurun_out/final_prof -o bench640 --output-format csv -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-overlap-probe > gpurun_out/final_bench_rocprof.json 2> gpurun_out/final_bench_rocprof.err
find gpurun_out/final_prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/final_kernel_stats.csv \;
find gpurun_out/final_prof -type f ! -name "*kernel_stats.csv" -delete
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final_smoke.txt 2>&1
tail -1 gpurun_out/final_smoke.txt
