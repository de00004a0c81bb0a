#!/bin/bash
# usage: bash tools/exp/pmc.sh <exe> <tag> "<counters pass 1>" "<counters pass 2>" ...
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
EXE=$R/$1; TAG=$2; shift 2
i=0
for C in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_${TAG}_$i -o p -- $EXE > $R/gpurun_out/pmc_${TAG}_$i.log 2>&1 || { tail -5 $R/gpurun_out/pmc_${TAG}_$i.log; }
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/pmc_${TAG}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
