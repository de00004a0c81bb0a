#!/bin/bash
# usage (GPU box): bash tools/exp/fc0_dvfs.sh -- does the clock the chip holds under the kernel's matrix instructions set its time?
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for a in "" "zerow" "zerov" "zerow zerov"; do
  echo "[$a] $(timeout -k 10 200 python tools/exp/bench_fc0.py uniform $a 2>/dev/null | tail -1)"
done
