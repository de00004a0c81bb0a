#!/bin/bash
# usage: bash tools/ab_env.sh VAR A B   -- same-box A/B of the step time for two values of an environment switch
one() { env $1=$2 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-query --no-diag --no-fwd-only --no-f32-backward 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1=$2', round(d['ms_per_step'],3), round(d['step_ms']['median'],3))"; }
for i in 1 2; do one $1 $2; one $1 $3; done
