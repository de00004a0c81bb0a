#!/bin/bash
# usage (GPU box): bash tools/exp/fc0_timeline.sh ["<SVR_FC_DEFS>"] -- measurement build + in-kernel timeline (uniform points, then one repeated point)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
SVR_FC_DEFS="$1" SVR_FC0_MEASURE=1 python -c 'import __graft_entry__ as g; g.build()' > /dev/null 2>&1
timeout -k 10 200 python tools/exp/fc0_timeline.py 2>&1 | grep -v amdgpu.ids
echo "=== same point"
timeout -k 10 200 python tools/exp/fc0_timeline.py same 2>&1 | grep -v amdgpu.ids | grep -v "    slab"
python -c 'import __graft_entry__ as g; g.build()' > /dev/null 2>&1
