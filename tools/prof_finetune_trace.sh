#!/bin/bash
# per-dispatch kernel trace of the fine-tune bench, summarised by (kernel, grid): which launches of a shared kernel cost what
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/ft_trace
rocprofv3 --kernel-trace -d gpurun_out/ft_trace -o ft --output-format csv -- python3 bench.py --config finetune-tinyllama --steps 2 --warmup 1 > gpurun_out/ft_trace_bench.json 2> gpurun_out/ft_trace_bench.err
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/ft_trace/**/*kernel_trace.csv', recursive=True)[0]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    name = r['Kernel_Name'][:70]
    key = (name, r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?')))
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    agg[key][0] += 1; agg[key][1] += d
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for v in agg.values())
with open('gpurun_out/ft_trace_summary.txt', 'w') as o:
    for (name, g, w), (n, t) in rows[:60]:
        o.write(f"{name:70s} grid {g:>9s} wg {w:>5s} calls {n:6d} avg {t / n:9.1f} us  {100 * t / tot:5.2f} %\n")
PY
find gpurun_out/ft_trace -type f -delete
