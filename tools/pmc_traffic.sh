#!/bin/bash
# HBM traffic of the fused gather kernels: two SEPARATE rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE) of a short
# bench run, as MI355X_MICROARCH.md (sections HBM, rocprofv3 PMC slots) prescribes: one TCC counter per pass,
# --kernel-trace only, the program directly after `--`.  Fails fast on a non-zero rocprofv3 exit; every profiler run
# sits behind its own wall-clock timeout.
#   usage (GPU box): bash tools/pmc_traffic.sh [tag] [extra bench.py args]
#   -> gpurun_out/gather_traffic.json  (copy to profiles/gather_traffic.json; bench.py reads that)
#   -> gpurun_out/pmc_traffic_<COUNTER>/   (counter CSVs; keep the gather rows under profiles/)
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r03}; shift || true
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf "$R/gpurun_out/pmc_traffic_$C"
  timeout -k 10 420 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$R/gpurun_out/pmc_traffic_$C" -o p -- \
      python3 "$R/bench.py" --no-cpu-baseline --no-fwd-only --no-query --no-f32-backward --steps 2 --warmup 1 "$@" > "$R/gpurun_out/pmc_traffic_$C.log" 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then
    echo "pmc_traffic: rocprofv3 --pmc $C failed (rc=$rc)"; tail -15 "$R/gpurun_out/pmc_traffic_$C.log"; exit 1
  fi
done
python3 - "$R" "$TAG" "$@" <<'PY'
import collections, csv, glob, json, sys
R, tag, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
def arg(name, default):
    return type(default)(extra[extra.index(name) + 1]) if name in extra else default
agg = collections.defaultdict(lambda: collections.defaultdict(list))
rows = collections.defaultdict(list)
for f in glob.glob(f"{R}/gpurun_out/pmc_traffic_*/**/*counter_collection.csv", recursive=True):
    rd = csv.DictReader(open(f))
    for r in rd:
        n = r["Kernel_Name"]
        for key in ("gather_fc0_kernel", "gather_bwd_proj_kernel", "gather_fwd_fused_kernel", "gather_bwd_fused_kernel", "gather_bwd_pull_kernel<16", "gather_bwd_pull_kernel<32",
                    "gather_bwd_pull_kernel<64", "stage1_kernel<0>", "stage1_kernel<1>", "stage1_kernel<2>", "stage1_kernel<3>"):
            # (stage1_kernel<MODE, H3>: the key keeps the mode only)
            if key in n or (key.startswith("stage1_kernel<") and key[:-1] + "," in n):
                agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
                rows[r["Counter_Name"]].append(r)
for cname, rr in rows.items():                      # the gather rows of each pass, for profiles/
    with open(f"{R}/gpurun_out/{tag}_pmc_{cname.lower()}_gather_rows.csv", "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rr[0].keys()))
        w.writeheader()
        w.writerows(rr)
out = {"round": tag, "batch": arg("--batch", 8), "grid": arg("--grid", 128), "points": arg("--points", 50000),
       "dist": arg("--dist", "uniform")}
for k, d in agg.items():
    out[k] = {c + "_KB": sum(v) / len(v) for c, v in d.items()}
    out[k]["dispatches_averaged"] = len(next(iter(d.values())))
f = out.get("gather_fwd_fused_kernel", {})
if "FETCH_SIZE_KB" in f and "WRITE_SIZE_KB" in f:
    fetch = f["FETCH_SIZE_KB"] * 1024
    out["fetch_bytes_raw"] = fetch
    out["fetch_bytes_corrected_x2"] = 2 * fetch     # gfx950: FETCH_SIZE counts 128-B requests at 64 B (guide, section HBM)
    out["write_bytes"] = f["WRITE_SIZE_KB"] * 1024
    out["hbm_bytes_per_launch"] = 2 * fetch + f["WRITE_SIZE_KB"] * 1024
g = out.get("gather_fc0_kernel", {})
if "FETCH_SIZE_KB" in g and "WRITE_SIZE_KB" in g:
    out["fused_fetch_bytes_corrected_x2"] = 2 * g["FETCH_SIZE_KB"] * 1024
    out["fused_write_bytes"] = g["WRITE_SIZE_KB"] * 1024
    out["fused_hbm_bytes_per_launch"] = 2 * g["FETCH_SIZE_KB"] * 1024 + g["WRITE_SIZE_KB"] * 1024
hbm = 0.0
for k in list(out):
    if k.startswith("gather_bwd_") and "proj" not in k and isinstance(out[k], dict) and "WRITE_SIZE_KB" in out[k] and "FETCH_SIZE_KB" in out[k]:
        hbm += 2 * out[k]["FETCH_SIZE_KB"] * 1024 + out[k]["WRITE_SIZE_KB"] * 1024     # the kernels of svr_gather_trilinear_bwd
if hbm:
    out["gather_bwd_hbm_bytes"] = hbm
pv = out.get("gather_bwd_proj_kernel")
if pv and "FETCH_SIZE_KB" in pv and "WRITE_SIZE_KB" in pv:          # projected scatter: per launch (two launches per step)
    out["proj_hbm_bytes_per_launch"] = 2 * pv["FETCH_SIZE_KB"] * 1024 + pv["WRITE_SIZE_KB"] * 1024
    out["proj_atomic_bytes_per_launch"] = pv["WRITE_SIZE_KB"] * 1024
for i, name in enumerate(("stats", "apply", "bwd_reduce", "bwd_apply")):      # recomputed first stage (stage1.hip)
    k = out.get(f"stage1_kernel<{i}>")
    if k and "FETCH_SIZE_KB" in k and "WRITE_SIZE_KB" in k:
        out[f"stage1_{name}_hbm_bytes"] = 2 * k["FETCH_SIZE_KB"] * 1024 + k["WRITE_SIZE_KB"] * 1024
if "gather_bwd_fused_kernel" in out:
    out["gather_bwd_write_bytes"] = out["gather_bwd_fused_kernel"].get("WRITE_SIZE_KB", 0.0) * 1024   # = float atomics issued
out["source"] = (f"separate rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE) of `bench.py --steps 2 --warmup 1 {' '.join(extra)}` "
                 f"(tools/pmc_traffic.sh), averaged over the dispatches of each kernel; rows kept in profiles/{tag}_pmc_*_gather_rows.csv")
json.dump(out, open(f"{R}/gpurun_out/gather_traffic.json", "w"), indent=1)
print(json.dumps(out))
PY
