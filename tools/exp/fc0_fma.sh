#!/bin/bash
# usage (GPU box): bash tools/exp/fc0_fma.sh  -- the fused gather -> fc_0 kernel with ATen's rounding of the corner sum (FC_FMA=0)
# against the v_pk_fma_f32 chain (FC_FMA=1): stand-alone kernel time, then the step (same box)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in 0 1; do
  SVR_FC_FMA=$v python -c 'import __graft_entry__ as g; g.build()' > /dev/null 2>&1
  echo "FC_FMA=$v: $(timeout -k 10 200 python tools/exp/bench_fc0.py 2>/dev/null | tail -1)"
  echo "FC_FMA=$v step: $(timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-query --no-diag --no-fwd-only --no-f32-backward 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), d['step_ms'], d['roofline']['ms_per_launch'])")"
done
python -c 'import __graft_entry__ as g; g.build()' > /dev/null 2>&1
