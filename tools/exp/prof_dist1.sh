#!/bin/bash
# usage (GPU box): bash tools/exp/prof_dist1.sh <tag>  -- the bench step as rank 0 of a 1-rank RCCL group (no launcher: env only)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$1 -o $1 -- python3 $R/bench.py --gpus 1 --no-cpu-baseline --no-fwd-only --no-query --steps 5 --warmup 2 > $R/gpurun_out/$1_bench.json 2> $R/gpurun_out/$1_prof.err
grep -o '"ms_per_step": *[0-9.]*' $R/gpurun_out/$1_bench.json | head -1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/prof_$1/$1_kernel_stats.csv")))
for r in rows:
    if 'nccl' in r['Name'].lower() or 'rccl' in r['Name'].lower() or 'AllReduce' in r['Name']:
        print(r['Name'][:100], r['Calls'], float(r['AverageNs'])/1e3)
PY
