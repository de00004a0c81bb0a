#!/bin/bash
# usage (GPU box): bash tools/exp/calibrate_fetch.sh -> gpurun_out/fetch_calibration.json: FETCH_SIZE (KB, as rocprofv3 reports it) per
# algorithmic byte for streaming and for gather-shaped reads with known byte counts
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/pmc_calib
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_calib -o p -- python3 $R/tools/exp/calibrate_fetch.py > $R/gpurun_out/calib.log 2>&1 || { echo "rocprofv3 failed"; tail -5 $R/gpurun_out/calib.log; exit 1; }
python3 - <<PY
import csv, glob, json, collections
meta = json.loads([l for l in open("$R/gpurun_out/calib.log") if l.startswith("{")][-1])
rows = []
for f in glob.glob("$R/gpurun_out/pmc_calib/**/*counter_collection.csv", recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == "FETCH_SIZE"]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
big = [r for r in rows if float(r["Counter_Value"]) > 2e5]          # the 2 GiB kernels (KB units)
res = {"note": "FETCH_SIZE in the counter's KB / known bytes read; 1.0 = the counter is exact, 0.5 = it reports half (the guide's x2 correction)",
       "kernels": []}
for r in big:
    res["kernels"].append({"kernel": r["Kernel_Name"][:70], "FETCH_SIZE_KB": float(r["Counter_Value"])})
# reduce kernels first (3 x sum), then 3 gathers per C
sums = [k for k in res["kernels"] if "reduce" in k["kernel"].lower()]
gath = [k for k in res["kernels"] if "index" in k["kernel"].lower() or "gather" in k["kernel"].lower()]
if sums:
    res["stream_ratio"] = sum(k["FETCH_SIZE_KB"] for k in sums) / len(sums) * 1024 / meta["stream_bytes"]
per = len(gath) // max(len(meta["cases"]), 1)
for i, c in enumerate(meta["cases"]):
    ks = gath[i * per:(i + 1) * per]
    if ks:
        c["ratio"] = sum(k["FETCH_SIZE_KB"] for k in ks) / len(ks) * 1024 / c["gather_read_bytes"]
res["cases"] = meta["cases"]
json.dump(res, open("$R/gpurun_out/fetch_calibration.json", "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "kernels"}, indent=1))
print(len(res["kernels"]), "kernels over 200 MB:", collections.Counter(k["kernel"][:40] for k in res["kernels"]))
PY
