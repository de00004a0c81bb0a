#!/bin/bash
# usage: bash tools/exp/dbg_geo.sh "<SVR_FC_DEFS>" <script>
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
SVR_FC_DEFS="$1" SVR_FC0_MEASURE=1 python -c 'import __graft_entry__ as g; g.build()' > gpurun_out/dbg_build.log 2>&1 || tail -5 gpurun_out/dbg_build.log
SVR_FC0_STAGE=0 timeout -k 10 120 python tools/exp/${2:-dbg_geo.py} 2>&1 | grep -v amdgpu.ids
python -c 'import __graft_entry__ as g; g.build()' > /dev/null 2>&1
