#!/bin/bash
# usage (GPU box): bash tools/exp/unet_target.sh 512 1024 2048 : kernel-time totals of the UNet micro-benchmark per value of SVR_IG_TARGET
# (workgroups the reduction split aims for; any other measurement variable through VAR=...)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for t in "$@"; do
  export ${VAR:-SVR_IG_TARGET}=$t
  rm -rf $R/gpurun_out/ut_$t
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ut_$t -o u -- python3 $R/tools/exp/bench_unet.py --reps 10 > $R/gpurun_out/ut_$t.log 2>&1 || { echo "target $t failed"; continue; }
  python3 - $R/gpurun_out/ut_$t/u_kernel_stats.csv $t <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6
sel = {k: sum(float(r["TotalDurationNs"]) for r in rows if k in r["Name"]) / 1e6 for k in ("conv2d_igemm_kernel<64", "conv2d_igemm_kernel<128", "ig_reduce", "linear_tn_x3_tr", "conv_wgrad_reduce")}
print("target", sys.argv[2], "total kernel ms (13 fwd + 13 steps)", round(tot, 2), {k: round(v, 2) for k, v in sel.items()})
PY
done
