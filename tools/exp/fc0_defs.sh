#!/bin/bash
# usage (GPU box): bash tools/exp/fc0_defs.sh "<defs 1>" "<defs 2>" ...   e.g. "" "FC_PRIO=1" "FC_PK=0 FC_FMA=1"
# builds gather_fc0.hip with each set of -D switches (SVR_FC_DEFS) and times the fused kernel stand-alone (uniform points and
# one repeated point) -- same box, one after the other; the default build is restored at the end
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for defs in "$@"; do
  SVR_FC_DEFS="$defs" python -c 'import __graft_entry__ as g; g.build()' > /dev/null 2>&1 || { echo "[$defs] build failed"; continue; }
  for d in uniform same; do
    echo "[$defs] $(timeout -k 10 200 python tools/exp/bench_fc0.py $d 2>/dev/null | tail -1 | sed 's/gather [0-9.]* ms, gather+fc_0 separate [0-9.]* ms, //')"
  done
done
python -c 'import __graft_entry__ as g; g.build()' > /dev/null 2>&1
