#!/bin/bash
# Same-box comparison of the backward scatter's forms inside the step (GPU box): float atomics vs the atomic-free forms
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
one() { python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-query --no-diag --no-fwd-only 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],3), round(d['step_ms']['median'],3), d['scatter_forms'])"; }
for i in 1 2; do
  one "default            "
  SVR_PROJ_TWO_PASS=1 one "two-pass L4        "
  SVR_PROJ_TWO_PASS=1 SVR_PROJ_TWO_PASS_MIN_DIM=1 one "two-pass L4+L5     "
  SVR_PROJ_TWO_PASS=1 SVR_PROJ_TWO_PASS_MIN_DIM=1 SVR_PULL_MAX_WALK=128 one "two-pass + pull L3 "
  SVR_PULL_MAX_WALK=128 one "pull L3 only       "
done
