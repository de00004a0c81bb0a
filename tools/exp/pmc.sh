#!/bin/bash
# usage: bash tools/exp/pmc.sh <exe> <tag> "<counters pass 1>" "<counters pass 2>" ...
# One rocprofv3 --pmc pass per argument.  gfx950 has 8 SQ, 4 TCC and 2 GRBM slots per pass (MI355X_MICROARCH.md
# "rocprofv3 PMC slots"; FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2): a pass that asks for more dies inside the
# profiler with "error code 38: Request exceeds the capabilities of the hardware to collect" (round 1,
# gpurun_out/pmc_tn_4.log).  So every pass is budget-checked BEFORE it runs, a failing pass stops the script, and each
# profiler run sits behind a wall-clock timeout (the program stays directly after `--`).
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
EXE=$R/$1; TAG=$2; shift 2
budget_ok() {   # $1 = counter list of one pass
  local sq=0 tcc=0 grbm=0 other=0 c
  for c in $1; do
    case $c in
      FETCH_SIZE) tcc=$((tcc+3));; WRITE_SIZE) tcc=$((tcc+2));;
      SQ_*) sq=$((sq+1));; TCC_*) tcc=$((tcc+1));; GRBM_*) grbm=$((grbm+1));; *) other=$((other+1));;
    esac
  done
  if [ $sq -gt 8 ] || [ $tcc -gt 4 ] || [ $grbm -gt 2 ] || [ $other -gt 4 ]; then
    echo "pmc.sh: pass '$1' exceeds the per-pass counter budget (SQ $sq/8, TCC $tcc/4, GRBM $grbm/2, other $other/4): split it"
    return 1
  fi
}
for C in "$@"; do budget_ok "$C" || exit 2; done
i=0
for C in "$@"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_${TAG}_$i -o p -- $EXE > $R/gpurun_out/pmc_${TAG}_$i.log 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then echo "pmc.sh: pass $i ('$C') failed, rc=$rc"; tail -8 $R/gpurun_out/pmc_${TAG}_$i.log; exit 1; fi
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/pmc_${TAG}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
