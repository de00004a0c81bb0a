"""Print the fields of a bench.py JSON line that matter when comparing runs (usage: python tools/bench_summary.py file.json ...)."""
import json
import sys

for path in sys.argv[1:]:
    try:
        d = json.loads([l for l in open(path) if l.startswith("{")][-1])
    except Exception as e:      # noqa: BLE001
        print(path, "unreadable:", e)
        continue
    print("==", path)
    for k in ("value", "ms_per_step", "step_ms", "step_ms_list", "step_max_over_median", "host_enqueue_ms", "allocator",
              "scatter_forms", "all_reduce_ms", "backward_exact_f32", "fwd_only", "encoder_ms_per_step", "mlp_gemm_ms_per_step"):
        if k in d:
            v = d[k]
            if isinstance(v, dict):
                v = {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a != "note"}
            print(" ", k, v)
    r = d["roofline"]
    print("  roofline", {k: r.get(k) for k in ("ms_per_launch", "ms_per_launch_stats", "ms_per_launch_single_stream", "frac",
                                              "traffic_over_compulsory", "mfma_frac_of_f16_peak")})
    q = d.get("query_path") or {}
    for k in ("dtype_f32", "dtype_bf16", "lattice", "config1_bf16"):
        if q.get(k):
            print("  query", k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in q[k].items() if a not in ("workload",)})
    for k in d.get("roofline_kernels", []):
        print("   ", k["kernel"][:90], "| ms", round(k.get("ms_per_step", k.get("ms_per_launch", 0)), 3), "| frac",
              None if k.get("frac") is None else round(k["frac"], 3))
    if "cpu_baseline" in d:
        print("  cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
