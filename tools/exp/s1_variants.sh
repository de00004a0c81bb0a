#!/bin/bash
# stage1.hip measurement builds on the GPU box: kernel times of the four passes with the tail / the MFMAs switched off
# usage: bash tools/exp/s1_variants.sh   (writes gpurun_out/s1_exp_<n>.txt)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for e in 0 1 2; do
  SVR_S1_EXP=$e python3 -c "
import importlib,sys,os
sys.path.insert(0,'.')
b=importlib.import_module('single-view-3d-reconstruction_amd.build')
os.utime('single-view-3d-reconstruction_amd/csrc/stage1.hip')
b.build()" || exit 1
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_s1e$e -o s1e$e -- python3 $R/bench.py --no-cpu-baseline --no-fwd-only --no-query --no-diag --steps 3 --warmup 1 > /dev/null 2> $R/gpurun_out/s1e${e}.err)
  f=$(find $R/gpurun_out/prof_s1e$e -name "*kernel_stats.csv" | head -1)
  echo "== S1_EXP=$e" | tee $R/gpurun_out/s1_exp_$e.txt
  grep stage1_kernel $f | awk -F, '{print $1, $2, $4}' | tee -a $R/gpurun_out/s1_exp_$e.txt
done
# restore the production build
python3 -c "
import importlib,sys,os
sys.path.insert(0,'.')
b=importlib.import_module('single-view-3d-reconstruction_amd.build')
os.utime('single-view-3d-reconstruction_amd/csrc/stage1.hip')
b.build()"
