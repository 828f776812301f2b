#!/bin/bash
# bench.py at several --prefill-batches (packed prefill launches of 32 x pb prompts), driver form; one line per run
for pb in "$@"; do
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-overlap-probe --prefill-batches $pb > gpurun_out/bench_pbx.json 2> gpurun_out/bench_pbx.err || exit 1
    python -c "
import json
r=json.loads(open('gpurun_out/bench_pbx.json').read().strip().splitlines()[-1])
print($pb, round(r['value'],1), round(r['roofline']['frac'],4), round(r['phases']['prefill_ms_per_step'],2), round(r['phases']['decode_ms_per_step'],2))"
done
