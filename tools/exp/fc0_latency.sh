#!/bin/bash
# usage (GPU box): bash tools/exp/fc0_latency.sh -- is the fused gather -> fc_0 kernel bound by memory latency?  Same kernel, same
# instruction stream, three point sets: uniform (fine levels from Infinity Cache / HBM), tiny (cube of edge 0.1: L2 resident),
# same (one point: every load hits the same lines); then the tile order contiguous per XCD.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for d in uniform tiny same; do echo "$(timeout -k 10 200 python tools/exp/bench_fc0.py $d 2>/dev/null | tail -1)"; done
echo "SVR_FC0_XCD=1: $(SVR_FC0_XCD=1 timeout -k 10 200 python tools/exp/bench_fc0.py uniform 2>/dev/null | tail -1)"
echo "SVR_FC0_STAGE=0: $(SVR_FC0_STAGE=0 timeout -k 10 200 python tools/exp/bench_fc0.py uniform 2>/dev/null | tail -1)"
