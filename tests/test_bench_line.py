"""bench.py's stdout contract: the LAST stdout line is one compact JSON object the driver can parse -- below
4 KB at world 1 and at world 8, carrying the contract's keys plus `roofline` and `cpu_baseline` -- and the full record goes to
bench_detail.json.  report() / compact() / emit() run here on canned timings (no GPU)."""
import argparse
import io
import json
import os
import sys
from contextlib import redirect_stderr, redirect_stdout

import pytest

import bench


class _Timer:
    def __init__(self, ev):
        self.ev = {k: [None] * len(v) for k, v in ev.items()}
        self._t = ev

    def times(self, k):
        return list(self._t.get(k, []))

    def ms_per_step(self, k, steps):
        return sum(self.times(k)) / max(steps, 1)

    def ms_per_launch(self, k):
        v = self.times(k)
        return sum(v) / max(len(v), 1)


class _Arena:
    grown = 3

    def nbytes(self):
        return 4_400_000_000


def _canned(world):
    a = argparse.Namespace(gpus=world, steps=20, warmup=5, batch=8, grid=128, points=50000, dist="uniform",
                           detail=None)
    step_ms = [15.3 + 0.01 * i for i in range(a.steps)]
    kt = _Timer({"gather_fc0_run": [2.7 + 0.001 * i for i in range(a.steps)],
                 "all_reduce_mean": [0.05 + 0.2 * (world > 1)] * a.steps})
    D = bench.DIAG_STEPS
    kd = _Timer({"gather_fc0_run": [2.6] * D, "gather_bwd": [1.3] * D, "gather_project_bwd": [1.5, 0.5] * D,
                 "linear_fwd:400000x256x256": [0.3, 0.3] * D, "linear_bwd_data:400000x256x256": [0.4, 0.4] * D,
                 "linear_bwd_data:400000x256x800": [0.69] * D, "linear_bwd_weight:400000x256x800": [0.87] * D,
                 "conv_fwd:8x64x64x64:16->32": [0.4] * D, "conv_bwd_data:8x64x64x64:32->16": [0.33] * D,
                 "conv_bwd_weight:8x64x64x64:16->32": [0.4] * D, "stage1_fwd:8x128x128x128x16": [0.47] * D,
                 "stage1_bwd:8x128x128x128x16": [0.78] * D, "bn_fwd:8x64x64x64x32": [0.05] * D, "bn_bwd:8x64x64x64x32": [0.1] * D})
    ranks = [{"rank": r, "local_rank": r, "device": r, "device_name": "AMD Instinct MI355X", "wall_ms_per_step": 15.4 + 0.01 * r,
              "step_ms": bench._stats(step_ms), "host_enqueue_ms": bench._stats([3.5] * a.steps),
              "all_reduce_ms": bench._stats(kt.times("all_reduce_mean")), "gather_fc0_ms": bench._stats(kt.times("gather_fc0_run")),
              "hipMalloc_calls_in_timed_region": 0, "hipMalloc_calls_per_step": [0] * a.steps, "hipMalloc_sites": None,
              "hipFree_calls_in_timed_region": 0, "reserved_bytes_grown_in_timed_region": 0, "reserved_bytes": 10_600_000_000}
             for r in range(world)]
    forms = {"level1": "pull (atomic-free)", "level2": "pull (atomic-free)", "level3": "item-order float atomics"}
    res = bench.report(a, world, 15.4e-3 * a.steps, 123.456789, step_ms, [3.5] * a.steps, kt, kd, ranks, forms, 5.7, None,
                       _Arena(), 3)
    res["rccl"] = {"world_size": world, "launched_by_torchrun": world > 1, "backend": "nccl (RCCL)", "device_count": 8,
                   "devices": {str(r): r for r in range(world)}}
    res["roofline_gather_rows"] = bench.gather_rows_roofline(a, bench._stats([2.11, 2.13, 2.12, 2.2, 2.12]))
    for key in bench.ALT_BACKWARD:
        res[key] = {"ms_per_step": 24.9, "value": 1.6e7, "unit": "query-points/s", "note": bench.ALT_BACKWARD[key][1]}
    if world == 1:
        res["cpu_baseline"] = {"value": 6099.7, "unit": "query-points/s", "cores": 128, "kind": "port", "value_median": 6050.1,
                               "fwd_only": {"value": 19000.0, "value_median": 18500.0, "unit": "query-points/s"},
                               "cpu_model": "AMD EPYC 9575F 64-Core Processor", "sample_short": "1 sample (B=1, 128^3, 50000 pts), "
                               "1 warm-up + 3 reps, best; torch CPU ops, 128 threads", "sample": "x" * 400}
    return res


@pytest.mark.parametrize("world", [1, 8])
def test_compact_line_is_short_and_complete(world, tmp_path):
    res = _canned(world)
    out, err = io.StringIO(), io.StringIO()
    detail = str(tmp_path / "bench_detail.json")
    with redirect_stdout(out), redirect_stderr(err):
        bench.emit(res, detail)
    lines = [l for l in out.getvalue().splitlines() if l.strip()]
    assert len(lines) == 1                                   # ONE JSON line on stdout
    line = lines[-1]
    assert len(line) < 3072, len(line)
    c = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "step_ms", "hipMalloc_calls_in_timed_region", "rccl"):
        assert k in c, k
    assert c["n_gpus"] == world and c["config"]["global_batch"] == 8 * world and c["config"]["parallelism"] == f"dp{world}"
    assert set(c["config"]) >= {"workload", "global_batch", "parallelism", "arithmetic"} and "model" not in c["config"]
    r = c["roofline"]
    for k in ("kernel", "bound", "unit", "peak", "achieved", "frac", "traffic", "compulsory_bytes", "ms_per_launch", "mfma_frac"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and len(r["kernel"]) <= 60
    if r["frac"] is not None:
        assert 0 < r["frac"] <= 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
        assert "stored" in r["traffic_source"]
    g = c["roofline_gather_rows"]                            # the stand-alone gather (rows to HBM) beside the fused kernel
    assert g["ms_per_launch"] == pytest.approx(2.12) and g["bound"] == "hbm" and len(g["kernel"]) <= 60
    if g["frac"] is not None:
        assert 0 < g["frac"] <= 1 and abs(g["frac"] - g["achieved"] / g["peak"]) < 1e-3
    assert c["value"] == pytest.approx(world * 8 * 50000 / 15.4e-3, rel=1e-3)
    if world == 1:
        b = c["cpu_baseline"]
        assert set(b) >= {"value", "unit", "cores", "kind", "sample", "value_median", "cpu_model"} and b["kind"] == "port"
        assert "rank_ms_per_step" not in c["rccl"]
    else:
        assert len(c["rccl"]["rank_ms_per_step"]) == world and c["rccl"]["world_size"] == world
        assert c["rccl"]["all_reduce_ms_median"] > 0
    assert c["backward_exact_f32"]["ms_per_step"] == pytest.approx(24.9)
    # the arithmetic disclosure follows the code's switches (it said "conv_in exact f32" after stage 1 moved to f16x3)
    from svr_amd import ops
    arith = c["config"]["arithmetic"]
    assert ("conv_in f16x3" in arith) == bool(ops.stage1_arith())
    assert f"bwd gemm {ops.BACKWARD_GEMM}" in arith and f"fwd conv {ops.FORWARD_CONV}" in arith
    # the full record round-trips and holds what left the line
    full = json.loads(open(detail).read())
    assert "roofline_kernels" in full and "step_ms_list" in full and len(full["ranks"]) == world
    assert err.getvalue().startswith("bench_detail {")
