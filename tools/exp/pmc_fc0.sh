#!/bin/bash
# usage (GPU box): bash tools/exp/pmc_fc0.sh <tag>   -- separate rocprofv3 --pmc passes over the fused gather->fc_0 micro-benchmark
# (tools/exp/bench_fc0.py); SQ counters only, <= 8 per pass (MI355X_MICROARCH.md "rocprofv3 PMC slots"); fails fast.
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-fc0}
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_${TAG}_$i -o p -- python3 $R/tools/exp/bench_fc0.py > $R/gpurun_out/pmc_${TAG}_$i.log 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then echo "pmc_fc0: pass $i ('$C') failed, rc=$rc"; tail -8 $R/gpurun_out/pmc_${TAG}_$i.log; exit 1; fi
done
python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/pmc_${TAG}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        for key in ("gather_fc0_kernel", "gather_fwd_fused_kernel", "linear_nt_h3_kernel"):
            if key in n:
                agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
for k, d in out.items():
    if "SQ_WAVE_CYCLES" in d:
        wc = d["SQ_WAVE_CYCLES"]
        d["wait_any_frac_of_wave_cycles"] = d.get("SQ_WAIT_ANY", 0) / wc
        d["wait_inst_any_frac_of_wave_cycles"] = d.get("SQ_WAIT_INST_ANY", 0) / wc
    if "SQ_BUSY_CYCLES" in d and "SQ_VALU_MFMA_BUSY_CYCLES" in d:
        d["mfma_busy_frac_of_busy_cycles"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / d["SQ_BUSY_CYCLES"]
    if "TCC_HIT_sum" in d:
        d["l2_hit_rate"] = d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
json.dump(out, open("$R/gpurun_out/${TAG}_pmc_summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
