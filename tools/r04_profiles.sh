#!/bin/bash
# Round-4 records in ONE gpurun call: driver-form bench + the same command under rocprofv3 --kernel-trace --stats + smoke (final_profile.sh),
# the two --pmc passes over the layer's four prefill GEMM launches (separate runs: FETCH_SIZE and WRITE_SIZE do not fit one pass), the side lines.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/final_profile.sh
rm -rf gpurun_out/pmcF gpurun_out/pmcW
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmcF -o f -- python3 tools/pmc_gemm.py > gpurun_out/pmcF.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmcW -o w -- python3 tools/pmc_gemm.py > gpurun_out/pmcW.log 2>&1
python tools/pmc_summary.py $(find gpurun_out/pmcF -name "*.db") $(find gpurun_out/pmcW -name "*.db") gpurun_out/r04_pmc_gemm.json > gpurun_out/r04_pmc_gemm.txt 2>&1 || true
rm -rf gpurun_out/pmcF gpurun_out/pmcW
echo "pmc done"; cat gpurun_out/r04_pmc_gemm.txt
python bench.py --ragged --steps 20 --warmup 5 --no-overlap-probe > gpurun_out/r04_bench_ragged.json 2> gpurun_out/r04_side.err
python bench.py --no-overlap-probe > gpurun_out/r04_bench_default64.json 2>> gpurun_out/r04_side.err
python bench.py --config llama3-8b-fp8 --steps 8 --warmup 2 > gpurun_out/r04_bench_llama3_fp8.json 2>> gpurun_out/r04_side.err
python bench.py --config llama3-8b-bf16 --steps 8 --warmup 2 > gpurun_out/r04_bench_llama3_bf16.json 2>> gpurun_out/r04_side.err
python bench.py --config finetune-tinyllama --steps 16 --warmup 2 > gpurun_out/r04_bench_finetune.json 2>> gpurun_out/r04_side.err
echo "side lines done"
