#!/bin/bash
# usage: bash tools/exp/dbg_variants.sh "<defs 1>" "<defs 2>" ... : production build (no MEASURE) of each variant + dbg_fc0.py summary
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for defs in "$@"; do
  SVR_FC_DEFS="$defs" python -c 'import __graft_entry__ as g; g.build()' > /dev/null 2>&1 || { echo "[$defs] build failed"; continue; }
  echo "[$defs] $(SVR_FC0_STAGE=0 timeout -k 10 120 python tools/exp/dbg_fc0.py 2>&1 | grep -c differ) level-reps differ (of 12); $(SVR_FC0_STAGE=0 timeout -k 10 120 python tools/exp/dbg_fc0.py 2>&1 | grep -c differ) again"
done
python -c 'import __graft_entry__ as g; g.build()' > /dev/null 2>&1
