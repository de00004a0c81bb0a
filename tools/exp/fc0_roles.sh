#!/bin/bash
# usage (GPU box): bash tools/exp/fc0_roles.sh -- measurement build (-DSVR_FC0_MEASURE: WRONG results by construction) of the fused
# gather -> fc_0 kernel with its roles switched off: SVR_FC0_DBG 0 = complete, 1 = producers idle, 2 = consumers idle, 3 = both (the
# barrier skeleton), for uniform points and for one repeated point (no memory system behind the loads)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
SVR_FC0_MEASURE=1 python -c 'import __graft_entry__ as g; g.build()' > /dev/null 2>&1
for d in uniform same; do for g in 0 1 2 3; do
  echo "dbg=$g $(SVR_FC0_DBG=$g timeout -k 10 200 python tools/exp/bench_fc0.py $d 2>/dev/null | tail -1 | sed 's/gather [0-9.]* ms, gather+fc_0 separate [0-9.]* ms, //')"
done; done
python -c 'import __graft_entry__ as g; g.build()' > /dev/null 2>&1
