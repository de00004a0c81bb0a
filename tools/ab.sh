#!/bin/bash
# Same-box A/B of the step time: HEAD (exported into _ab/, built there) against the working tree.  Boxes differ by ~2 %
# (clocks, DVFS), so a change below that only shows in a same-box comparison.
#   here:        bash tools/ab.sh prepare          (git archive HEAD -> _ab/, build)
#   on the box:  gpurun -- 'bash tools/ab.sh run'  (two alternating rounds of bench.py --steps 20)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
cd "$R"
if [ "$1" = "prepare" ]; then
  rm -rf _ab && mkdir -p _ab && git archive HEAD | tar -x -C _ab
  (cd _ab && python3 -c "
import importlib,sys
sys.path.insert(0,'.')
print(importlib.import_module('single-view-3d-reconstruction_amd.build').build())")
  exit 0
fi
one() { python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-query --no-diag --no-fwd-only 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],3), round(d['step_ms']['median'],3))"; }
for i in 1 2; do (cd _ab && one "HEAD   "); one "working"; done
