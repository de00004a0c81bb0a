#!/bin/bash
# usage (on the GPU box): bash tools/prof_step.sh <tag>   -> gpurun_out/<tag>_kernel_stats.csv + bench line
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$1 -o $1 -- python3 $R/bench.py --no-cpu-baseline --no-fwd-only --no-query --no-f32-backward --steps 5 --warmup 2 > $R/gpurun_out/$1_bench.json 2> $R/gpurun_out/$1_prof.err
find $R/gpurun_out/prof_$1 -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/$1_kernel_stats.csv \;
cat $R/gpurun_out/$1_bench.json | grep -o '"ms_per_step": *[0-9.]*' | head -1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/$1_kernel_stats.csv")))
for r in rows[:22]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} total_ms/step={float(r['TotalDurationNs'])/1e6/7:8.3f}")
PY
