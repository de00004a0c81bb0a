#!/bin/bash
# usage (GPU box): bash tools/exp/fc0_variants.sh  -- builds gather_fc0.hip tile variants in place and times them
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in "128 4" "64 1" "64 2"; do
  set -- $v
  touch single-view-3d-reconstruction_amd/csrc/gather_fc0.hip
  SVR_FC_TM=$1 SVR_FC_DEPTH=$2 python -c 'import __graft_entry__ as g; g.build()' > /dev/null 2>&1
  echo "TM=$1 DEPTH=$2: $(timeout -k 10 200 python tools/exp/bench_fc0.py 2>/dev/null | tail -1)"
done
touch single-view-3d-reconstruction_amd/csrc/gather_fc0.hip
python -c 'import __graft_entry__ as g; g.build()' > /dev/null 2>&1
