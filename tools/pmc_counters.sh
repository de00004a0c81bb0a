#!/bin/bash
# Per-kernel averages of arbitrary PMC counters: one rocprofv3 pass per counter GROUP (quote a group: "A B C"), --kernel-trace
# only, the program directly after `--` (MI355X_MICROARCH.md, rocprofv3 PMC slots).
#   usage (GPU box): bash tools/pmc_counters.sh <kernel-name substring> "<group 1>" ["<group 2>" ...]
#   -> gpurun_out/pmc_counters_<n>/ and a table on stdout
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
PAT=$1; shift
budget_ok() {   # $1 = counter list of one pass: 8 SQ / 4 TCC / 2 GRBM slots (an oversubscribed pass is rocprofv3's "error code 38")
  local sq=0 tcc=0 grbm=0 other=0 c
  for c in $1; do
    case $c in
      FETCH_SIZE) tcc=$((tcc+3));; WRITE_SIZE) tcc=$((tcc+2));;
      SQ_*) sq=$((sq+1));; TCC_*) tcc=$((tcc+1));; GRBM_*) grbm=$((grbm+1));; *) other=$((other+1));;
    esac
  done
  if [ $sq -gt 8 ] || [ $tcc -gt 4 ] || [ $grbm -gt 2 ] || [ $other -gt 4 ]; then
    echo "pmc_counters: pass '$1' exceeds the per-pass counter budget (SQ $sq/8, TCC $tcc/4, GRBM $grbm/2, other $other/4): split it"
    return 1
  fi
}
for G in "$@"; do budget_ok "$G" || exit 2; done
n=0
for G in "$@"; do
  n=$((n + 1))
  rm -rf "$R/gpurun_out/pmc_counters_$n"
  timeout -k 10 420 rocprofv3 --kernel-trace --pmc $G --output-format csv -d "$R/gpurun_out/pmc_counters_$n" -o p -- \
      python3 "$R/bench.py" --no-cpu-baseline --no-fwd-only --no-query --no-f32-backward --no-diag --steps 2 --warmup 1 > "$R/gpurun_out/pmc_counters_$n.log" 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then echo "pmc_counters: group $n ($G) failed (rc=$rc)"; tail -5 "$R/gpurun_out/pmc_counters_$n.log"; exit 1; fi
done
python3 - "$R" "$PAT" <<'PY'
import collections, csv, glob, sys
R, pat = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{R}/gpurun_out/pmc_counters_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"    {c:32s} {sum(v) / len(v):16.1f}   (n={len(v)})")
PY
