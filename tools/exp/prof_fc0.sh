#!/bin/bash
# usage (GPU box): bash tools/exp/prof_fc0.sh <tag> [dbg]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export SVR_FC0_DBG=${2:-0}
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$1 -o $1 -- python3 $R/tools/exp/bench_fc0.py > $R/gpurun_out/$1.log 2>&1
find $R/gpurun_out/prof_$1 -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/$1_kernel_stats.csv \;
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/$1_kernel_stats.csv")))
for r in rows[:8]:
    print(f"{r['Name'][:80]:80s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f}")
PY
