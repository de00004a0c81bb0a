#!/bin/bash
# usage (GPU box): bash tools/exp/compare_backward.sh [modeA modeB] -- per-kernel times (single-stream brackets) of two backward arithmetics
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
A=${1:-bf16x3}; B=${2:-f16x3s}
for m in $A $B; do SVR_BACKWARD=$m python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-query --no-fwd-only --no-f32-backward --detail gpurun_out/detail_$m.json > /dev/null 2>&1; done
python - <<PY
import json
a=json.load(open("gpurun_out/detail_$A.json")); b=json.load(open("gpurun_out/detail_$B.json"))
print("step", round(a["ms_per_step"],3), round(b["ms_per_step"],3))
ka={k["kernel"]:k for k in a["roofline_kernels"]}; kb={k["kernel"]:k for k in b["roofline_kernels"]}
tot=0
for name in ka:
    if name in kb:
        x=ka[name].get("ms_per_step", ka[name].get("ms_per_launch",0)); y=kb[name].get("ms_per_step", kb[name].get("ms_per_launch",0))
        if abs(y-x)>0.008: print(f"{name[:84]:84s} {x:7.3f} -> {y:7.3f}  {y-x:+.3f}"); tot+=y-x
print("sum of differences", round(tot,3))
PY
