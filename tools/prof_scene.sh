#!/bin/bash
# usage (GPU box): bash tools/prof_scene.sh <tag>  -> gpurun_out/<tag>_kernel_stats.csv (config-5 step under rocprofv3 --stats)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$1 -o $1 -- python3 $R/tools/bench_scene.py --steps 5 --warmup 2 > $R/gpurun_out/$1_bench.json 2> $R/gpurun_out/$1_prof.err || { echo "rocprofv3 failed"; tail -5 $R/gpurun_out/$1_prof.err; exit 1; }
find $R/gpurun_out/prof_$1 -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/$1_kernel_stats.csv \;
cat $R/gpurun_out/$1_bench.json
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/$1_kernel_stats.csv")))
for r in rows[:28]:
    print(f"{r['Name'][:72]:72s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} total_ms/step={float(r['TotalDurationNs'])/1e6/7:8.3f}")
PY
