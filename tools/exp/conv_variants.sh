#!/bin/bash
# conv3d_bf16.hip measurement builds on the GPU box: encoder conv kernel times with parts of the inner loop changed
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for e in ${CONV_EXPS:-0 1 2 3 4}; do
  SVR_CONV_EXP=$e python3 -c "
import importlib,sys,os
sys.path.insert(0,'.')
b=importlib.import_module('single-view-3d-reconstruction_amd.build')
os.utime('single-view-3d-reconstruction_amd/csrc/conv3d_bf16.hip')
b.build()" || exit 1
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cve$e -o cve$e -- python3 $R/bench.py --no-cpu-baseline --no-fwd-only --no-query --no-diag --no-f32-backward --steps 3 --warmup 1 > /dev/null 2> $R/gpurun_out/cve${e}.err)
  f=$(find $R/gpurun_out/prof_cve$e -name "*kernel_stats.csv" | head -1)
  echo "== SVR_CONV_EXP=$e" | tee $R/gpurun_out/conv_exp_$e.txt
  grep "conv3d_brick_x3_kernel" $f | awk -F'","' '{print substr($1,1,90), $2, $4}' | tee -a $R/gpurun_out/conv_exp_$e.txt
done
python3 -c "
import importlib,sys,os
sys.path.insert(0,'.')
b=importlib.import_module('single-view-3d-reconstruction_amd.build')
os.utime('single-view-3d-reconstruction_amd/csrc/conv3d_bf16.hip')
b.build()"
