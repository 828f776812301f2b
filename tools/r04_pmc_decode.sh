#!/bin/bash
# fabric traffic of the 640-row decode layer's kernels: two --pmc passes (FETCH_SIZE, WRITE_SIZE) over tools/pmc_decode_layer.py
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmcDF gpurun_out/pmcDW
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmcDF -o f -- python3 tools/pmc_decode_layer.py > gpurun_out/pmcDF.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmcDW -o w -- python3 tools/pmc_decode_layer.py > gpurun_out/pmcDW.log 2>&1
python tools/pmc_decode_summary.py $(find gpurun_out/pmcDF -name "*.db") $(find gpurun_out/pmcDW -name "*.db") gpurun_out/r04_pmc_decode_layer640.json > gpurun_out/r04_pmc_decode_layer640.txt 2>&1 || true
rm -rf gpurun_out/pmcDF gpurun_out/pmcDW
cat gpurun_out/r04_pmc_decode_layer640.txt
