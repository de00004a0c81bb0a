#!/bin/bash
# conv3d_bwdw_bf16.hip measurement builds on the GPU box: weight-gradient kernel times with parts of the brick loop removed
# (1 = no MFMAs, 2 = no global loads, 3 = no split arithmetic; results are wrong in all three)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
rebuild() {
  SVR_WG_EXP=$1 python3 -c "
import importlib,sys,os
sys.path.insert(0,'.')
b=importlib.import_module('single-view-3d-reconstruction_amd.build')
os.utime('single-view-3d-reconstruction_amd/csrc/conv3d_bwdw_bf16.hip')
b.build()" || exit 1
}
for e in ${WG_EXPS:-0 1 2 3}; do
  rebuild $e
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_wge$e -o wge$e -- python3 $R/bench.py --no-cpu-baseline --no-fwd-only --no-query --no-diag --no-f32-backward --steps 3 --warmup 1 > /dev/null 2> $R/gpurun_out/wge${e}.err)
  echo "== SVR_WG_EXP=$e"
  python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/prof_wge$e/*kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f)) if 'bwd_weight_x3' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
for r in rows[-8:]:
    print("   grid=(%d,%s) %8.1f us" % (int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']), r['Grid_Size_Y'], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
PY
done
rebuild ""
