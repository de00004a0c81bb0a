#!/bin/bash
# HBM traffic of the fused gather kernels: two separate rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE) of a short
# bench run, as MI355X_MICROARCH.md prescribes.  usage (GPU box): bash tools/pmc_traffic.sh  -> gpurun_out/traffic.json
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_traffic_$C -o p -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/pmc_traffic_$C.log 2>&1
done
python3 - <<PY
import csv, glob, json, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/pmc_traffic_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        for key in ("gather_fwd_fused_kernel", "gather_bwd_fused_kernel"):
            if key in n:
                agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, d in agg.items():
    out[k] = {c + "_KB": sum(v) / len(v) for c, v in d.items()}
    out[k]["dispatches_averaged"] = len(next(iter(d.values())))
f = out.get("gather_fwd_fused_kernel", {})
if f:
    fetch = f["FETCH_SIZE_KB"] * 1024
    out["fetch_bytes_corrected_x2"] = 2 * fetch
    out["write_bytes"] = f["WRITE_SIZE_KB"] * 1024
    out["hbm_bytes_per_launch"] = 2 * fetch + f["WRITE_SIZE_KB"] * 1024
json.dump(out, open("$R/gpurun_out/traffic.json", "w"), indent=1)
print(json.dumps(out))
PY
