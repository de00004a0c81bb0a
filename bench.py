#!/usr/bin/env python
"""Headline benchmark: query-points/sec, forward+backward(+Adam), IF-Net 128^3 grid x 50k points.

    python bench.py --gpus N --steps K --warmup W

N = 1 runs in this process.  N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ...
bench.py --gpus N` (RANK / WORLD_SIZE in the environment; WORLD_SIZE must equal N), or -- without a launcher
environment -- this process starts that launcher itself as a CHILD (before anything touches the GPU) and exits with
its code, so `python bench.py --gpus 8` can never silently measure one GPU.

One "step" = one full training step of the hot path on one synthetic batch per GPU: 3D conv encoder + 6-level
trilinear gather + point MLP + BCE loss, backward of all of it, gradient all-reduce (N>1) and the Adam update.
Workload at every N: BASELINE.json configs[2] per GPU (128^3 grid, 50 000 points, batch 8; configs[3] is the same
per-GPU shard x 8 GPUs) -> weak scaling.  Inputs are resident in HBM before the timed region.  Prints ONE JSON line
on rank 0.

Roofline block: `roofline` is the north-star kernel -- the forward gather, since round 2 FUSED with
fc_0 (gather_fc0.hip: the feature rows never reach HBM).  `achieved` / `frac` are HBM bytes actually moved per launch
(rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/gather_traffic.json, measured on this workload) / the live
HIP-event time of the launch, against the 8 TB/s HBM peak -- a fraction <= 1.  The ALGORITHMIC rate (the
corner reads, most of them cache hits) is reported beside it against the L2 roof, the compulsory-traffic fraction and
the kernel's MFMA fraction too (it is bound by the vector-L1 rate of the corner reads, neither by HBM nor by MFMA).  `roofline_kernels` carries the other dominant kernels: the
backward scatter against the float-atomic roof and the point-MLP GEMMs against the f16 / bf16 MFMA peak.
"""
import argparse
import gc
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GATHER_BYTES_PER_POINT_F32 = 93000          # 7*8*369 reads + 7*369 writes + 12 B coords
FUSED_BYTES_PER_POINT_F32 = 7 * 8 * 369 * 4 + 12 + 256 * 4   # fused gather+fc_0: corner reads + coords + the h0 row
FUSED_KEPT_COLUMNS = 800                    # training: rows of the levels that are not projected (+ padding) are kept
FC0_K_FUSED = 2592                          # fused reduction length (2583 feature columns + 9 zero columns)
GATHER_BYTES_PER_POINT_BF16 = 46506         # the same elements at 2 B + 12 B coords
GATHER_BWD_BYTES_PER_POINT_F32 = 92776      # 2583 gradient reads + 12 B + 7*8*368 RMW (counted once)
HBM_PEAK_GBPS = 8000.0                       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured achievable)
HBM_ACHIEVABLE_GBPS = 6290.0
L2_PEAK_GBPS = 34500.0                       # MI355X_MICROARCH.md section L2: ~34.5 TB/s aggregate
ATOMIC_PEAK_GBPS = 1300.0                    # MI355X_MICROARCH.md "Global float atomics": ~1.3 TB/s of added bytes
MFMA_F16_PEAK_TFLOPS = 2500.0                # dense f16 / bf16 MFMA peak (spec)
MLP_FLOP_PER_POINT = 1585152                 # forward FLOP per query point of fc_0..fc_out (f32-equivalent)
DIAG_STEPS = 2                               # untimed single-stream steps behind the timed region (per-kernel times)
SETUP_STEPS = 2                              # untimed steps before the W warm-up steps (allocator pools, scatter-form decision)
# the same step under other backward arithmetic, measured behind the timed region: key -> (ops.BACKWARD_* mode, note)
ALT_BACKWARD = {
    "backward_exact_f32": ("f32", "the same step with every backward GEMM / convolution on the exact-f32 MFMA kernels "
                                  "(ops.BACKWARD_* = 'f32') instead of the f32-level scaled f16 split"),
    "backward_bf16x3": ("bf16x3", "the same step with every backward GEMM / convolution on the 3-term bf16 split (16 mantissa bits "
                                  "per operand, ~1.5e-5 per product: the default of rounds 1-3; ops.BACKWARD_* = 'bf16x3', "
                                  "SVR_BACKWARD=bf16x3)"),
}
SPLIT_PRODUCTS = 3                           # f16x3 / bf16x3: three MFMA products per f32-equivalent product


def synth_batch(seed, B, D, N, device, dist="uniform"):
    import torch
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = (torch.rand(B, 1, D, D, D, generator=g) < 0.05).float()
    if dist == "surface":
        # secondary, locality-friendly distribution: points on random planes + N(0, 0.01 / 0.1) noise,
        # like data_processing/mesh_occupancies.py:14-17 samples around the mesh surface
        origin = torch.rand(B, 8, 1, 3, generator=g) - 0.5
        u = torch.randn(B, 8, 1, 3, generator=g)
        v = torch.randn(B, 8, 1, 3, generator=g)
        ab = torch.rand(B, 8, N // 8, 2, generator=g) - 0.5
        p = origin + ab[..., :1] * u * 0.3 + ab[..., 1:] * v * 0.3
        sigma = torch.where(torch.rand(B, 8, N // 8, 1, generator=g) < 0.5, 0.01, 0.1)
        pts = (p + torch.randn(B, 8, N // 8, 3, generator=g) * sigma).reshape(B, -1, 3).clamp(-0.5, 0.5)
        if pts.shape[1] < N:
            pts = torch.cat([pts, torch.rand(B, N - pts.shape[1], 3, generator=g) - 0.5], 1)
    else:
        pts = torch.rand(B, N, 3, generator=g) - 0.5
    occ = (torch.rand(B, N, generator=g) < 0.5).float()
    return {"input": x.to(device), "points": pts.to(device), "occupancies": occ.to(device)}


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(D, N, reps=3):
    """The CPU oracle (= the reference's op sequence in stock torch CPU ops, oracle/ifnet_oracle.py) on ONE sample
    of the same workload on this host's cores: 1 warm-up + `reps` timed repetitions, forward and backward timed
    separately, min and median reported.  ~8 s per repetition on the GPU box."""
    import statistics

    import torch
    from oracle import ifnet_oracle as O
    threads = torch.get_num_threads()
    b = synth_batch(103, 1, D, N, "cpu")
    st = O.make_leaf_state(O.name_seeded_state(128))
    fwd, tot = [], []
    for rep in range(reps + 1):
        for v in st.values():
            v.grad = None
        t0 = time.perf_counter()
        out = O.training_step(st, b, 128)
        t1 = time.perf_counter()
        out["loss"].backward()
        t2 = time.perf_counter()
        if rep > 0:                                       # repetition 0 is the warm-up
            fwd.append(t1 - t0)
            tot.append(t2 - t0)
    return {"value": N / min(tot), "unit": "query-points/s", "cores": threads, "kind": "port",
            "value_median": N / statistics.median(tot),
            "fwd_only": {"value": N / min(fwd), "value_median": N / statistics.median(fwd), "unit": "query-points/s"},
            "cpu_model": _cpu_model(),
            "sample_short": f"1 sample (B=1, {D}^3, {N} pts), 1 warm-up + {reps} reps, best; torch CPU ops, {threads} threads",
            "sample": f"1 sample of the workload (B=1, {D}^3 grid, {N} points; x B = one GPU batch), 1 warm-up + {reps} timed "
                      f"repetitions: fwd+bwd {', '.join(f'{t:.2f}' for t in tot)} s, fwd {', '.join(f'{t:.2f}' for t in fwd)} s; "
                      f"value = best, value_median = median; torch {torch.__version__} CPU ops with {threads} threads "
                      "(oracle/ifnet_oracle.py, the reference's op sequence)"}


def _relaunch_under_torchrun(a):
    """`python bench.py --gpus N` without a launcher environment: start N ranks as a child process (nothing in THIS
    process has touched the GPU yet) and pass its exit code on."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("GPU_MAX_HW_QUEUES", "8")
    return subprocess.call(cmd, env=env)


def _malloc_sites(snap):
    """--alloc-trace: every segment_alloc (= hipMalloc) of the recorded history with the frames of this repository on the
    Python stack of the request that caused it."""
    out = []
    for trace in snap.get("device_traces", []):
        for i, ev in enumerate(trace):
            if ev.get("action") != "segment_alloc":
                continue
            frames = ev.get("frames") or []
            if not frames:
                for ev2 in trace[i + 1:i + 4]:
                    if ev2.get("action") == "alloc":
                        frames = ev2.get("frames") or []
                        break
            mine = [f"{os.path.relpath(f['filename'], ROOT)}:{f['line']} {f['name']}" for f in frames if ROOT in f.get("filename", "")]
            out.append({"bytes": ev.get("size"), "stream": ev.get("stream"), "frames": mine[:5]})
    return out


def _stats(v):
    """min / median / p90 / max / mean of a list of milliseconds."""
    if not v:
        return None
    import statistics
    w = sorted(v)
    return {"n": len(w), "min": w[0], "median": statistics.median(w), "p90": w[min(len(w) - 1, int(round(0.9 * (len(w) - 1))))],
            "max": w[-1], "mean": sum(w) / len(w)}


class _KernelTimer:
    """HIP-event brackets around selected ops on the stream they are enqueued on (torch's current stream)."""

    def __init__(self, torch):
        self.torch = torch
        self.ev = {}

    def wrap(self, mod, name, key=None):
        orig = getattr(mod, name)
        torch = self.torch

        def timed(*args, **kw):
            k = key(*args, **kw) if key else name
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig(*args, **kw)
            e1.record()
            self.ev.setdefault(k, []).append((e0, e1))
            return r

        setattr(mod, name, timed)
        return orig

    def times(self, k):
        return [e0.elapsed_time(e1) for e0, e1 in self.ev.get(k, [])]

    def ms_per_step(self, k, steps):
        """Time of the key's launches per step: the brackets are step-major, so they split into `steps` equal runs; the MINIMUM
        over the steps is reported (a transient inside one bracket -- an allocator refill in the first single-stream step put
        66 ms into one BatchNorm bracket of profiles/r04_bench_v3_detail.json -- must not become a kernel's time)."""
        v = self.times(k)
        if not v:
            return 0.0
        if steps <= 1 or len(v) % steps:
            return sum(v) / max(steps, 1)
        n = len(v) // steps
        return min(sum(v[i * n:(i + 1) * n]) for i in range(steps))

    def ms_per_launch(self, k):
        v = self.times(k)
        return sum(v) / max(len(v), 1)


def _conv_key(op, x, co, ci):
    B, D, H, W = x.shape[0], x.shape[1], x.shape[2], x.shape[3]
    return f"{op}:{B}x{D}x{H}x{W}:{ci}->{co}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8, help="samples per GPU")
    ap.add_argument("--grid", type=int, default=128)
    ap.add_argument("--points", type=int, default=50000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fwd-only", action="store_true")
    ap.add_argument("--no-query", action="store_true", help="skip the query-path (f32 / bf16 storage / lattice) measurement")
    ap.add_argument("--no-diag", action="store_true", help="skip the untimed single-stream steps behind the timed region")
    ap.add_argument("--no-f32-backward", action="store_true",
                    help="skip the extra steps with the exact-f32 / f32-level (bf16x6) backward kernels")
    ap.add_argument("--detail", default=os.path.join(ROOT, "bench_detail.json"),
                    help="where the full record goes (per-kernel roofline table, per-step lists, per-rank dicts, query path); "
                         "stdout carries only the compact line")
    ap.add_argument("--alloc-trace", action="store_true",
                    help="record the caching allocator's history over the timed region and report the call sites of every hipMalloc")
    ap.add_argument("--backward", choices=["production", "f32"], default="production",
                    help="arithmetic of the backward GEMMs / convolutions: the production bf16x3 split, or exact-f32 MFMA "
                         "everywhere (ops.BACKWARD_GEMM / BACKWARD_CONV / BACKWARD_CONV_WEIGHT = 'f32'): what the step costs "
                         "without the split")
    ap.add_argument("--dist", choices=["uniform", "surface"], default="uniform",
                    help="query-point distribution; uniform (default) is the reported worst case")
    a = ap.parse_args()

    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ     # under torch.distributed.run
    if a.gpus > 1 and not launched:
        sys.exit(_relaunch_under_torchrun(a))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {a.gpus}")

    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # main + two side streams + RCCL's (see the package's __init__)
    # RCCL prints its version banner to file descriptor 1 whatever NCCL_DEBUG says (profiles/r04_bench_torchrun_world1.json:
    # five lines in front of the JSON).  The compact line must be the only thing on stdout: Python's sys.stdout moves to a
    # duplicate of the real stdout and descriptor 1 itself is pointed at stderr, so native libraries print there.
    sys.stdout.flush()
    sys.stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    import svr_amd  # noqa: F401
    from svr_amd import ops
    from svr_amd.dp import DataParallelTrainer
    from svr_amd.model import ifnet as _ifn
    from svr_amd.trainer import ImplicitRefinementTrainer
    from oracle import ifnet_oracle as O            # name-seeded weights only (checker-side helper)

    if a.backward == "f32":
        ops.BACKWARD_GEMM = ops.BACKWARD_CONV = ops.BACKWARD_CONV_WEIGHT = "f32"
    torch.cuda.set_device(local_rank)
    if launched:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        if dist.get_world_size() != a.gpus:
            sys.exit(f"bench.py: process group has {dist.get_world_size()} ranks, --gpus {a.gpus}")
    dev = torch.device("cuda", local_rank)

    trainer = ImplicitRefinementTrainer()
    trainer.ifnet.load_state_dict(O.name_seeded_state(128), strict=False)   # random init of the architecture
    trainer = trainer.to(dev).train()
    opt = torch.optim.Adam(trainer.ifnet.parameters(), lr=trainer.hparams.lr, fused=True)
    dp = DataParallelTrainer(trainer, optimizer=opt)
    batch = synth_batch(103 + rank, a.batch, a.grid, a.points, dev, a.dist)

    def sync():
        if launched:
            dist.barrier()
        torch.cuda.synchronize()

    # set-up, not warm-up: the first steps of a process size the step arena and the allocator pools, and the per-level
    # scatter form is decided from the statistics of the step PULL_DECISION_LAG (3) steps back (ifnet.SCATTER_FORM "auto")
    # -- SETUP_STEPS + the W warm-up steps of the contract run before the timed region
    for _ in range(SETUP_STEPS):
        dp.step(batch)
    sync()
    gc.collect()      # (see below: in FRONT of the warm-up steps, so that the GPU does not sit idle for the ~0.1 s of the collection
    gc.freeze()       #  right before the timed region -- the first timed step then also pays for the clock ramp)
    for _ in range(a.warmup):
        dp.step(batch)
    # live HIP-event timing over the timed region, on the stream the work is enqueued on: the roofline kernel (the fused
    # gather -> fc_0 launch alone, ops.gather_fc0_run; ops.gather_fwd where the fused kernel does not apply), the gradient
    # all-reduce, and one event per step boundary (per-step GPU times without a host synchronisation)
    kt = _KernelTimer(torch)
    restore = [(ops, "gather_fwd", kt.wrap(ops, "gather_fwd")), (ops, "gather_fc0_run", kt.wrap(ops, "gather_fc0_run")),
               (dp.bucket, "all_reduce_mean", kt.wrap(dp.bucket, "all_reduce_mean"))]
    mem0 = torch.cuda.memory_stats(dev)
    arena = trainer.ifnet.ifnet_feature_extractor._arena
    grown0 = arena.grown
    # Python's cyclic collector: a gen-2 collection over the module / autograd object graph stalls the enqueueing thread for
    # 40-200 ms (one 54.7 ms step in a 200-step run, profiles/r04_bench_200steps_gc_default.json; INTEGRATION.md section 6
    # tells a training loop to do the same).  Everything alive after the set-up steps was moved to the permanent generation
    # (above, in front of the warm-up steps); the collector stays on for what the timed steps allocate.
    sync()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    host_ms, malloc_at = [], []
    if a.alloc_trace:
        torch.cuda.memory._record_memory_history(max_entries=1000000, stacks="python")
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(a.steps):
        h0 = time.perf_counter()
        loss_t = dp.step(batch)["loss"].detach()     # (not the graph: a kept loss tensor keeps its step's activations alive)
        marks[i + 1].record()
        host_ms.append((time.perf_counter() - h0) * 1e3)
        malloc_at.append(torch.cuda.memory_stats(dev).get("num_device_alloc", 0))      # host-side counter, no device call
    sync()
    dt = time.perf_counter() - t0
    gc.unfreeze()
    mem1 = torch.cuda.memory_stats(dev)
    malloc_sites = None
    if a.alloc_trace:
        malloc_sites = _malloc_sites(torch.cuda.memory._snapshot())
        torch.cuda.memory._record_memory_history(enabled=None)
    for mod, name, orig in restore:
        setattr(mod, name, orig)
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps)]
    forms = {}
    for key, h in _ifn._pull_hint.items():
        forms[f"level{key[1]}"] = "pull (atomic-free)" if h["use"] else "item-order float atomics"
    # the other kernels of `roofline_kernels` (encoder, scatter, GEMMs) and the roofline kernel stand-alone: DIAG_STEPS
    # extra, untimed steps with the step's streams collapsed into one (ifnet.OVERLAP_BACKWARD off, SVR_NO_SIDE_STREAM), so
    # that a HIP-event bracket holds one kernel's own time -- in the measured step those kernels run side by side and
    # their brackets stretch 1.3-3.5x
    kd = _KernelTimer(torch)
    if not a.no_diag:
        w = kd.wrap
        diag = [(ops, "gather_bwd", w(ops, "gather_bwd")), (ops, "gather_project_bwd", w(ops, "gather_project_bwd")),
                (ops, "gather_fc0_run", w(ops, "gather_fc0_run")),
                # linear_fwd(x (M,K), w (N,K)); linear_bwd_data(dy (M,N), w (N,K)); linear_bwd_weight(dy (M,N), x (M,K))
                (ops, "linear_fwd", w(ops, "linear_fwd", lambda x, w, *r, **k: f"linear_fwd:{x.shape[0]}x{w.shape[0]}x{x.shape[-1]}")),
                (ops, "linear_bwd_data", w(ops, "linear_bwd_data", lambda dy, w, *r, **k: f"linear_bwd_data:{dy.shape[0]}x{dy.shape[-1]}x{w.shape[-1]}")),
                (ops, "linear_bwd_weight", w(ops, "linear_bwd_weight", lambda dy, x, *r, **k: f"linear_bwd_weight:{dy.shape[0]}x{dy.shape[-1]}x{x.shape[-1]}")),
                # encoder: conv forward / backward-data / weight gradient keyed by volume and channels; BatchNorm passes
                (ops, "conv3d_k3_fwd", w(ops, "conv3d_k3_fwd", lambda x, wt, *r, **k: _conv_key("conv_fwd", x, wt.shape[0], wt.shape[1]))),
                (ops, "conv3d_c1_fwd_stats", w(ops, "conv3d_c1_fwd_stats", lambda x, wt, *r, **k: _conv_key("conv_in_fwd", x, wt.shape[0], 1))),
                (ops, "conv3d_k3_bwd_data", w(ops, "conv3d_k3_bwd_data", lambda dy, wt, *r, **k: _conv_key("conv_bwd_data", dy, wt.shape[0], wt.shape[1]))),
                (ops, "conv3d_k3_bwd_weight", w(ops, "conv3d_k3_bwd_weight", lambda x, dy, *r, **k: _conv_key("conv_bwd_weight", x, dy.shape[4], x.shape[4]))),
                (ops, "stage1_fwd", w(ops, "stage1_fwd", lambda x, wt, *r, **k: "stage1_fwd:" + "x".join(str(v) for v in x.shape[:4]) + f"x{wt.shape[0]}")),
                (ops, "stage1_bwd", w(ops, "stage1_bwd", lambda x, wp, *r, **k: "stage1_bwd:" + "x".join(str(v) for v in x.shape[:4]) + f"x{wp.shape[2]}")),
                (ops, "bn_forward", w(ops, "bn_forward", lambda x, *r, **k: "bn_fwd:" + "x".join(str(v) for v in x.shape))),
                (ops, "bn_backward", w(ops, "bn_backward", lambda x, *r, **k: "bn_bwd:" + "x".join(str(v) for v in x.shape)))]
        prev_overlap, _ifn.OVERLAP_BACKWARD = _ifn.OVERLAP_BACKWARD, False
        prev_env = os.environ.get("SVR_NO_SIDE_STREAM")
        os.environ["SVR_NO_SIDE_STREAM"] = "1"
        try:
            for _ in range(DIAG_STEPS):
                dp.step(batch)
            sync()
        finally:
            _ifn.OVERLAP_BACKWARD = prev_overlap
            if prev_env is None:
                os.environ.pop("SVR_NO_SIDE_STREAM", None)
            else:
                os.environ["SVR_NO_SIDE_STREAM"] = prev_env
            for mod, name, orig in diag:
                setattr(mod, name, orig)
    # the stand-alone feature gather (svr_gather_trilinear_fwd: the (B*N, 2592) feature rows written to HBM -- the kernel the
    # north star's "fraction of the HBM roofline on the trilinear feature gather" is stated on; callers that want the rows
    # themselves use it, the step uses the fused kernel above): a few launches behind the timed region, HIP events on the
    # launch stream.  Present in the PMC passes of tools/pmc_traffic.sh too (they keep the diagnostics on).
    unfused_ms = None
    if not a.no_diag and rank == 0:
        with torch.no_grad():
            ext = trainer.ifnet.ifnet_feature_extractor
            lv = ext.encode_levels(batch["input"])
            pts_u = ops.morton_order(batch["points"].float().contiguous(), want_sorted=True)[1]    # as IFNet.forward visits them
            rows = ops.gather_fwd(lv, pts_u, ext._layout, ext._disp, ext._align)
            ku = _KernelTimer(torch)
            orig_g = ku.wrap(ops, "gather_fwd")
            for _ in range(5):
                ops.gather_fwd(lv, pts_u, ext._layout, ext._disp, ext._align, out=rows)
            torch.cuda.synchronize()
            ops.gather_fwd = orig_g
            unfused_ms = _stats(ku.times("gather_fwd"))
            del rows, lv
    # what the step costs WITHOUT the bf16x3 split in the backward: a few extra steps behind the timed
    # region with the backward GEMMs / convolutions on the exact-f32 MFMA kernels (rank 0's clock; never the headline)
    alt_ms = {}
    if not a.no_diag and not a.no_f32_backward and a.backward == "production":
        prev = (ops.BACKWARD_GEMM, ops.BACKWARD_CONV, ops.BACKWARD_CONV_WEIGHT)
        for key, (mode, _) in ALT_BACKWARD.items():
            if mode not in ops.BACKWARD_MODES or mode in prev:
                continue
            ops.BACKWARD_GEMM = ops.BACKWARD_CONV = ops.BACKWARD_CONV_WEIGHT = mode
            try:
                for _ in range(2):
                    dp.step(batch)
                sync()
                t1 = time.perf_counter()
                for _ in range(5):
                    dp.step(batch)
                sync()
                alt_ms[key] = (time.perf_counter() - t1) / 5 * 1e3
            finally:
                ops.BACKWARD_GEMM, ops.BACKWARD_CONV, ops.BACKWARD_CONV_WEIGHT = prev
    loss = float(loss_t)
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if launched:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)
    # per-rank record (N > 1: gathered on rank 0, so that one run of the driver's 8-GPU node is diagnosable on its own)
    mine = {"rank": rank, "local_rank": local_rank, "device": torch.cuda.current_device(),
            "device_name": torch.cuda.get_device_name(dev), "wall_ms_per_step": dt / a.steps * 1e3,
            "step_ms": _stats(step_ms), "host_enqueue_ms": _stats(host_ms),
            "all_reduce_ms": _stats(kt.times("all_reduce_mean")),
            "gather_fc0_ms": _stats(kt.times("gather_fc0_run") or kt.times("gather_fwd")),
            "hipMalloc_calls_in_timed_region": mem1.get("num_device_alloc", 0) - mem0.get("num_device_alloc", 0),
            "hipMalloc_calls_per_step": [b - a_ for a_, b in zip([mem0.get("num_device_alloc", 0)] + malloc_at[:-1], malloc_at)],
            "hipMalloc_sites": malloc_sites,
            "hipFree_calls_in_timed_region": mem1.get("num_device_free", 0) - mem0.get("num_device_free", 0),
            "reserved_bytes_grown_in_timed_region": mem1.get("reserved_bytes.all.current", 0) - mem0.get("reserved_bytes.all.current", 0),
            "reserved_bytes": mem1.get("reserved_bytes.all.current", 0)}
    ranks = [mine]
    if launched and world > 1:
        ranks = [None] * world
        dist.all_gather_object(ranks, mine)

    # forward only (no_grad, training-mode BatchNorm: the same kernels as the step's forward half), same batch
    fwd_ms = None
    if not a.no_fwd_only:
        with torch.no_grad():
            trainer.training_step(batch, 0)
            sync()
            t1 = time.perf_counter()
            for _ in range(a.steps):
                trainer.training_step(batch, 0)
            sync()
            fwd_ms = (time.perf_counter() - t1) / a.steps * 1e3

    # query path alone (cached pyramid -> gather + point MLP forward, no grad; the dense-grid-inference kernels):
    # default f32 storage against the bf16-storage throughput mode (north_star "bf16 occupancy logits"), and the f32 path
    # on the reference's dense lattice (model/ifnet.py:202-229: C-order lattice incl. the +-0.5 planes, 32 768-point chunks)
    query = None
    if not a.no_query and rank == 0:
        query = {}
        net = trainer.ifnet
        pts = batch["points"]
        with torch.no_grad():
            f32_gather = "gather_fc0_run" if _ifn.FUSE_FC0 else "gather_fwd"
            bf16_gather = "gather_fc0_bf16_run" if _ifn.FUSE_FC0_BF16 else "gather_fwd_bf16"
            for name, storage, gname in (("f32", "f32", f32_gather), ("bf16", "bf16", bf16_gather)):
                levels = net.encode(batch["input"], storage)
                z = net.query(levels, pts, spatial_sort=True)
                kq = _KernelTimer(torch)
                orig = kq.wrap(ops, gname)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(a.steps):
                    z = net.query(levels, pts, spatial_sort=True)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t1) / a.steps * 1e3
                setattr(ops, gname, orig)
                query[name] = {"ms": ms, "gather_ms": kq.ms_per_launch(gname), "gather_op": gname, "logits": z.float()}
                del levels
            # dense lattice: one sample's pyramid, the lattice of its grid resolution in the reference's chunks
            levels = net.encode(batch["input"][:1])
            lat = _ifn.make_3d_grid((-0.5,) * 3, (0.5,) * 3, (a.grid,) * 3).to(dev)
            chunk = 2048 * 16
            prep = net.prepare_query(levels, chunk)

            def lattice_pass():
                outs = []
                for pi in torch.split(lat, chunk):
                    outs.append(torch.sigmoid(net.query(levels, pi.unsqueeze(0), prepared=prep)))
                return torch.cat(outs, 1)
            lattice_pass()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            reps = max(1, min(a.steps, 3))
            for _ in range(reps):
                lattice_pass()
            torch.cuda.synchronize()
            lms = (time.perf_counter() - t1) / reps * 1e3
            query["lattice"] = {"ms": lms, "points": int(lat.shape[0]), "chunk": chunk, "value": lat.shape[0] / (lms * 1e-3),
                                "unit": "query-points/s",
                                "workload": f"one sample, {a.grid}^3 lattice incl. the +-0.5 planes (model/ifnet.py:202-229), "
                                            f"{chunk}-point chunks, cached pyramid, fc_0 prepared once, sigmoid, values on device"}
            del levels, lat
            # BASELINE configs[1] as stated: 64^3 grid, 10 000 points, batch 4, bf16 storage, gather + MLP kernels only
            g1 = torch.Generator(device="cpu").manual_seed(102)
            x1 = (torch.rand(4, 1, 64, 64, 64, generator=g1) < 0.05).float().to(dev)
            p1 = (torch.rand(4, 10000, 3, generator=g1) - 0.5).to(dev)
            lv1 = net.encode(x1, "bf16")
            prep1 = net.prepare_query(lv1, 10000)
            net.query(lv1, p1, spatial_sort=True, prepared=prep1)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            reps1 = 20
            for _ in range(reps1):
                net.query(lv1, p1, spatial_sort=True, prepared=prep1)
            torch.cuda.synchronize()
            ms1 = (time.perf_counter() - t1) / reps1 * 1e3
            query["config1_bf16"] = {"ms": ms1, "points": 40000, "value": 40000 / (ms1 * 1e-3), "unit": "query-points/s",
                                     "workload": "BASELINE configs[1]: 64^3 pyramid (batch 4) in bf16 storage, 10 000 points per sample, "
                                                 "Morton sort + fused gather -> fc_0 + fc_1, fc_2, fc_out; launch bound at this size"}
            del lv1, prep1
        zf, zb = query["f32"].pop("logits"), query["bf16"].pop("logits")
        query["bf16"]["logits_rel_dev_vs_f32_storage"] = float((zb - zf).abs().max() / zf.abs().max())
    if launched:
        dist.barrier()

    if rank == 0:
        res = report(a, world, dt, loss, step_ms, host_ms, kt, kd, ranks, forms, fwd_ms, query, arena, grown0)
        if unfused_ms:
            res["roofline_gather_rows"] = gather_rows_roofline(a, unfused_ms)
        res["rccl"] = {"world_size": world, "launched_by_torchrun": launched, "backend": "nccl (RCCL)" if launched else None,
                       "device_count": torch.cuda.device_count(),
                       "devices": {str(r["rank"]): r["device"] for r in ranks}}
        for key, ms in alt_ms.items():
            res[key] = {"ms_per_step": ms, "value": world * a.batch * a.points / (ms * 1e-3), "unit": "query-points/s",
                        "note": ALT_BACKWARD[key][1] + ": 5 steps behind the timed region"}
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(a.grid, a.points)
        emit(res, a.detail)
    if launched:
        dist.barrier()
        dist.destroy_process_group()


def _sig(v, n=5):
    """Floats of the compact line: n significant digits."""
    if isinstance(v, float):
        return float(f"{v:.{n}g}")
    return v


def compact(res):
    """The ONE line the driver parses (the 21 KB line of round 3 left BENCH_r03.parsed null): the contract's
    keys + a short roofline / cpu_baseline / backward-arithmetic / allocator / RCCL summary, <= 3 KB at any world size.
    Everything else (per-kernel table, per-step lists, per-rank dicts, query path) is bench_detail.json."""
    r, st, cfg = res["roofline"], res.get("step_ms") or {}, res["config"]
    out = {k: res[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                               "scaling", "vs_baseline", "dtype", "data")}
    out["step_ms"] = {k: st.get(k) for k in ("min", "median", "max")}
    out["config"] = {"workload": cfg["workload_short"], "global_batch": cfg["global_batch"], "parallelism": cfg["parallelism"],
                     "points": cfg["points"], "arithmetic": cfg["arithmetic_short"]}
    out["roofline"] = {"kernel": r["kernel_short"], "bound": r["bound"], "unit": r["unit"], "peak": r["peak"],
                       "achieved": r["achieved"], "frac": r["frac"], "traffic": r["traffic"],
                       "traffic_source": "stored PMC (profiles/gather_traffic.json; FETCH x2 uncalibrated for gathers: upper bound)"
                                         if r["traffic"] else None,
                       "compulsory_bytes": r["compulsory_bytes_per_launch"], "algorithmic_bytes": r["algorithmic_bytes_per_launch"],
                       "ms_per_launch": r["ms_per_launch"], "mfma_frac": r.get("mfma_frac_of_f16_peak")}
    g = res.get("roofline_gather_rows")
    if g:
        out["roofline_gather_rows"] = {"kernel": g["kernel_short"], "bound": "hbm", "unit": "GB/s", "peak": g["peak"],
                                       "achieved": g["achieved"], "frac": g["frac"], "traffic": g["traffic"],
                                       "ms_per_launch": g["ms_per_launch"]}
    c = res.get("cpu_baseline")
    if c:
        out["cpu_baseline"] = {"value": c["value"], "value_median": c["value_median"], "unit": c["unit"], "cores": c["cores"],
                               "kind": c["kind"], "cpu_model": c["cpu_model"], "fwd_only_value": c["fwd_only"]["value"],
                               "sample": c["sample_short"]}
    for key in ALT_BACKWARD:
        if key in res:
            out[key] = {"ms_per_step": res[key]["ms_per_step"]}
    if res.get("fwd_only"):
        out["fwd_only_ms"] = res["fwd_only"]["ms_per_step"]
    out["hipMalloc_calls_in_timed_region"] = res["allocator"]["hipMalloc_calls_in_timed_region"]
    rc = res.get("rccl") or {}
    out["rccl"] = {"world_size": rc.get("world_size"), "torchrun": rc.get("launched_by_torchrun"),
                   "device_count": rc.get("device_count")}
    if res["n_gpus"] > 1:
        out["rccl"]["rank_ms_per_step"] = [_sig(x["wall_ms_per_step"], 4) for x in res.get("ranks", [])]
    ar = res.get("all_reduce_ms")
    if ar:
        out["rccl"]["all_reduce_ms_median"] = ar["median"]
    out["detail"] = "bench_detail.json"

    def rnd(o):
        if isinstance(o, dict):
            return {k: rnd(v) for k, v in o.items()}
        if isinstance(o, list):
            return [rnd(v) for v in o]
        return _sig(o)
    return rnd(out)


def emit(res, detail_path):
    """bench_detail.json (+ the same object on stderr) first, then the compact line as the LAST and ONLY JSON line on stdout."""
    line = json.dumps(compact(res), separators=(",", ":"))
    assert len(line) < 3072, f"compact bench line grew to {len(line)} bytes"
    full = json.dumps(res)
    try:
        with open(detail_path, "w") as f:
            f.write(full + "\n")
    except OSError as e:
        print(f"bench.py: cannot write {detail_path}: {e}", file=sys.stderr)
    print("bench_detail " + full, file=sys.stderr, flush=True)
    print(line, flush=True)


def _ifn_mod():
    from svr_amd.model import ifnet
    return ifnet


def _backward_arithmetic():
    from svr_amd import ops
    modes = {ops.BACKWARD_GEMM, ops.BACKWARD_CONV, ops.BACKWARD_CONV_WEIGHT}
    names = {"bf16x3": "bf16x3 split (~1.5e-5 per product)", "f16x3s": "scaled 3-product f16 split (f32-level)", "f32": "exact f32 MFMA"}
    return " / ".join(names.get(m, m) for m in sorted(modes))


def _arithmetic():
    """The step's arithmetic, read from the switches the kernels are selected by (ops.FORWARD_* / BACKWARD_* /
    stage1_arith): (long form for bench_detail.json, short form for the compact line)."""
    from svr_amd import ops
    s1 = "f16x3 (recomputed, |x| < 65504)" if ops.stage1_arith() else "exact f32 (recomputed)"
    long = (f"f32 storage everywhere; forward GEMMs: {ops.FORWARD_GEMM}, forward convs: {ops.FORWARD_CONV} (3-product f16 split "
            f"on the f16 MFMA: f32-level, ~3e-7 of f64); conv_in (stage 1): {s1}; backward dX/dW GEMMs, conv backward-data and "
            f"conv weight gradients: {_backward_arithmetic()}; BN, gather/scatter, loss: exact f32")
    short = (f"f32 storage; fwd gemm {ops.FORWARD_GEMM}, fwd conv {ops.FORWARD_CONV}, conv_in {'f16x3' if ops.stage1_arith() else 'f32'}; "
             f"bwd gemm {ops.BACKWARD_GEMM}, bwd conv {ops.BACKWARD_CONV}, wgrad {ops.BACKWARD_CONV_WEIGHT}; BN/gather/scatter f32")
    return long, short


def gather_rows_roofline(a, ms):
    """Roofline entry of the stand-alone gather (rows to HBM): counter-measured HBM bytes of profiles/gather_traffic.json over
    the live median launch time.  Its traffic is dominated by the 4.1 GB of rows it must write (compulsory, streaming -- the
    part of the counter that the calibration of profiles/r04_fetch_calibration.json finds exact)."""
    npts = a.batch * a.points
    t = _traffic(a)
    traffic = t.get("hbm_bytes_per_launch")
    med = ms["median"]
    rate = traffic / (med * 1e-3) / 1e9 if traffic else None
    return {"kernel": "gather_fwd_fused_kernel (svr_gather_trilinear_fwd: all 6 levels, (B*N, 2592) f32 rows written to HBM)",
            "kernel_short": "gather_fwd_fused_kernel (stand-alone gather, rows to HBM)",
            "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBPS, "achieved": rate,
            "frac": (rate / HBM_PEAK_GBPS) if rate else None, "traffic": traffic,
            "write_bytes": t.get("write_bytes"), "fetch_bytes_corrected_x2": t.get("fetch_bytes_corrected_x2"),
            "rows_bytes_written": npts * 2592 * 4,
            "algorithmic_bytes_per_launch": npts * GATHER_BYTES_PER_POINT_F32,
            "ms_per_launch": med, "ms_per_launch_stats": ms,
            "note": "5 launches behind the timed region on the main stream (nothing beside them); not part of the step, which "
                    "runs the fused gather -> fc_0 kernel of `roofline` instead"}


def _traffic(a):
    tfile = os.path.join(ROOT, "profiles", "gather_traffic.json")
    if os.path.exists(tfile):
        try:
            t = json.load(open(tfile))
            if t.get("batch") == a.batch and t.get("grid") == a.grid and t.get("points") == a.points \
                    and t.get("dist", "uniform") == a.dist:
                return t
        except Exception:
            pass
    return {}


def report(a, world, dt, loss, step_ms, host_ms, kt, kd, ranks, forms, fwd_ms, query, arena, grown0):
    npts = a.batch * a.points
    value = world * npts * a.steps / dt
    fused = bool(kt.times("gather_fc0_run"))
    gkey = "gather_fc0_run" if fused else "gather_fwd"
    live = _stats(kt.times(gkey))                      # the launch inside the measured (multi-stream) step
    alone = _stats(kd.times(gkey))                     # the same launch in the single-stream steps behind it
    gather_ms = live["median"] if live else 0.0
    kept = FUSED_KEPT_COLUMNS * 4 if fused else 0
    alg_bytes = npts * ((FUSED_BYTES_PER_POINT_F32 + kept) if fused else GATHER_BYTES_PER_POINT_F32)
    alg_rate = alg_bytes / (gather_ms * 1e-3) / 1e9 if gather_ms > 0 else None
    # compulsory traffic: every pyramid volume once + coordinates + what the kernel must write once
    # (unfused: the feature rows; fused: the h0 rows and the kept columns)
    chans, d = [1, 16, 32, 64, 128, 128], a.grid
    vol_elems, level_elems = 0, []
    for i, c in enumerate(chans):
        vol_elems += c * d ** 3
        level_elems.append(c * d ** 3)
        if i >= 1:
            d = max(d // 2, 1)
    out_bytes = (256 * 4 + kept) if fused else 2583 * 4
    compulsory = a.batch * (vol_elems * 4 + a.points * 12 + a.points * out_bytes)
    t = _traffic(a)
    traffic = t.get("fused_hbm_bytes_per_launch") if fused else t.get("hbm_bytes_per_launch")
    bwd_atomic_bytes, bwd_hbm_bytes = t.get("gather_bwd_write_bytes"), t.get("gather_bwd_hbm_bytes")
    proj_hbm, proj_atomic, tsrc = t.get("proj_hbm_bytes_per_launch"), t.get("proj_atomic_bytes_per_launch"), t.get("source")
    hbm_rate = traffic / (gather_ms * 1e-3) / 1e9 if (traffic and gather_ms > 0) else None
    roofline = {
        "kernel_short": "gather_fc0_kernel (fused 6-level trilinear gather -> fc_0)" if fused else "gather_fwd_fused_kernel (6 levels)",
        "kernel": ("gather_fc0_kernel (svr_gather_fc0_run: all 6 levels gathered slab by slab into LDS and multiplied into "
                   "fc_0's 64 x 256 tile, one launch; the W split / slab table launches of svr_gather_fc0_prepare are outside "
                   "the bracket)") if fused else
                  "gather_fwd_fused_kernel (svr_gather_trilinear_fwd, all 6 levels in one launch)",
        "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBPS,
        # HBM bytes actually moved (PMC) / the MEDIAN live launch time in the timed region: a true fraction of the 8 TB/s
        # roof (null without counters); min / max / mean of the brackets and the stand-alone time are beside it
        "achieved": hbm_rate, "frac": (hbm_rate / HBM_PEAK_GBPS) if hbm_rate else None,
        "frac_of_achievable_6290": (hbm_rate / HBM_ACHIEVABLE_GBPS) if hbm_rate else None,
        "traffic": traffic, "traffic_source": tsrc, "ms_per_launch": gather_ms, "ms_per_launch_stats": live,
        "ms_per_launch_single_stream": alone,
        "frac_at_max_launch": (traffic / (live["max"] * 1e-3) / 1e9 / HBM_PEAK_GBPS) if (traffic and live) else None,
        "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_GBps": alg_rate,
        "algorithmic_frac_of_l2_peak_34500": (alg_rate / L2_PEAK_GBPS) if alg_rate else None,
        "compulsory_bytes_per_launch": compulsory,
        "compulsory_frac_of_hbm_peak": (compulsory / (gather_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if gather_ms > 0 else None,
        "traffic_over_compulsory": (traffic / compulsory) if traffic else None,
        "note": "ms_per_launch = median HIP-event bracket of the kernel launch over the timed region (the side stream's plan "
                "sorts may run beside it); algorithmic bytes are mostly L1/L2/Infinity-Cache hits, so they are priced against "
                "the L2 roof; frac is counter-measured HBM traffic against the HBM roof"}
    if fused and gather_ms > 0:
        tf = npts * 2.0 * 256 * FC0_K_FUSED * SPLIT_PRODUCTS / (gather_ms * 1e-3) / 1e12
        roofline["mfma_TFLOPs"] = tf
        roofline["mfma_frac_of_f16_peak"] = tf / MFMA_F16_PEAK_TFLOPS
        roofline["note"] += ("; the kernel also carries fc_0's product (3 f16 MFMA products per f32 product): "
                             "mfma_frac_of_f16_peak.  It is bound by the vector-L1 rate of the corner reads "
                             "(64 B/clk/CU), not by HBM or the matrix cores -- the separate kernels it replaces took "
                             "2.05 (gather, HBM-write bound) + 2.13 ms (fc_0)")
    st = _stats(step_ms)
    res = {
        "metric": "query-points/sec fwd+bwd (128^3 grid, 50k pts)", "value": value, "unit": "query-points/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "setup_steps": SETUP_STEPS, "ms_per_step": dt / a.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[2] per GPU: {a.grid}^3 grid, {a.points} query points, batch "
                               f"{a.batch}/GPU, full 3D conv encoder + 6-level trilinear gather + occupancy MLP, "
                               "fwd+bwd+grad all-reduce+Adam",
                   "workload_short": f"configs[2] per GPU: {a.grid}^3 grid, {a.points} pts, batch {a.batch}/GPU, encoder + gather + MLP, "
                                     "fwd+bwd+allreduce+Adam",
                   "global_batch": world * a.batch, "parallelism": f"dp{world}", "loss": loss, "points": a.dist,
                   "deterministic_scatter": bool(_ifn_mod().DETERMINISTIC),
                   "arithmetic": _arithmetic()[0], "arithmetic_short": _arithmetic()[1]},
        # every timed step on its own (HIP events at the step boundaries on the main stream; no host synchronisation inside
        # the region): a transient shows as max >> median, uniform contention as a shifted median
        "step_ms": st, "step_ms_list": [round(v, 3) for v in step_ms],
        "step_max_over_median": (st["max"] / st["median"]) if st else None,
        "host_enqueue_ms": _stats(host_ms),
        "allocator": {"hipMalloc_calls_in_timed_region": ranks[0]["hipMalloc_calls_in_timed_region"],
                      "hipMalloc_calls_per_step": ranks[0]["hipMalloc_calls_per_step"],
                      "hipMalloc_sites": ranks[0]["hipMalloc_sites"],
                      "hipFree_calls_in_timed_region": ranks[0]["hipFree_calls_in_timed_region"],
                      "reserved_bytes_grown_in_timed_region": ranks[0]["reserved_bytes_grown_in_timed_region"],
                      "reserved_bytes": ranks[0]["reserved_bytes"],
                      "arena_bytes": arena.nbytes(), "arena_buffers_grown_in_timed_region": arena.grown - grown0,
                      "note": "the step's large cross-stream buffers (kept columns 1.28 GB, their gradient, gradient volumes, "
                              "scatter plans) live in a per-module arena allocated in the set-up steps"},
        "scatter_forms": forms,
        "all_reduce_ms": ranks[0]["all_reduce_ms"],
        "roofline": roofline,
    }
    res["ranks"] = ranks
    if fwd_ms is not None:
        res["fwd_only"] = {"value": npts / (fwd_ms * 1e-3), "unit": "query-points/s", "ms_per_step": fwd_ms,
                           "note": "rank 0, forward + loss under no_grad, training-mode BatchNorm"}
    if query is not None:
        for name, bpp in (("f32", GATHER_BYTES_PER_POINT_F32), ("bf16", GATHER_BYTES_PER_POINT_BF16)):
            q = query[name]
            if q["gather_op"] == "gather_fc0_run":
                bpp = FUSED_BYTES_PER_POINT_F32          # gather_ms then covers gather AND fc_0
            elif q["gather_op"] == "gather_fc0_bf16_run":
                bpp = 7 * 8 * 369 * 2 + 12 + 256 * 2     # bf16 corner reads + coordinates + the bf16 h0 row
            q["value"] = npts / (q["ms"] * 1e-3)
            q["unit"] = "query-points/s"
            if not q["gather_ms"]:
                q["gather_ms"] = None
                continue
            q["gather_algorithmic_GBps"] = npts * bpp / (q["gather_ms"] * 1e-3) / 1e9
            q["gather_algorithmic_frac_of_l2_peak_34500"] = q["gather_algorithmic_GBps"] / L2_PEAK_GBPS
        res["query_path"] = {"workload": f"cached {a.grid}^3 pyramid (batch {a.batch}), {npts} query points per pass: 6-level "
                                         "trilinear gather + point MLP forward, no grad, points visited in Morton order (dense-grid inference kernels)",
                             "dtype_f32": query["f32"], "dtype_bf16": query["bf16"], "lattice": query.get("lattice"),
                             "config1_bf16": query.get("config1_bf16"),
                             "note": "bf16 = separately named bf16-STORAGE mode (bf16 volumes / feature rows / activations, "
                                     "f32 accumulation; never the default, not held to the fp32 1e-4 gate)"}
    kernels = []
    bwd_ms = kd.ms_per_launch("gather_bwd")
    if bwd_ms > 0:
        # compulsory traffic of the scatter: the gradient rows once + the gradient volumes (levels 1..5) once
        bwd_comp = a.batch * (a.points * 2583 * 4 + (vol_elems - a.grid ** 3) * 4)
        if fused:   # the 128-channel levels are projected: this call scatters the 784 columns of levels 1-3 only
            bwd_comp = a.batch * (a.points * 784 * 4 + sum(level_elems[1:4]) * 4)
        rate = bwd_hbm_bytes / (bwd_ms * 1e-3) / 1e9 if bwd_hbm_bytes else None
        kernels.append({"kernel": "gather_bwd (svr_gather_trilinear_bwd: per level the atomic-free pull form or the atomic "
                                  "scatter over the joint item order, chosen from the point distribution)", "bound": "hbm",
                        "unit": "GB/s", "peak": HBM_PEAK_GBPS,
                        "achieved": rate, "frac": (rate / HBM_PEAK_GBPS) if rate else None, "ms_per_launch": bwd_ms,
                        "traffic": bwd_hbm_bytes, "float_atomic_bytes": bwd_atomic_bytes,
                        "algorithmic_bytes_per_launch": npts * (GATHER_BWD_BYTES_PER_POINT_F32 if not fused else 784 * 4 + 12 + 7 * 8 * 112 * 4),
                        "compulsory_bytes_per_launch": bwd_comp,
                        "compulsory_frac_of_hbm_peak": bwd_comp / (bwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                        "note": "achieved = HBM bytes of all scatter kernels (PMC FETCH_SIZE x2 + WRITE_SIZE) / live time; the "
                                "round-1 kernel sat at the ~1.3 TB/s float-atomic rate (7.0 GB of atomics), the atomics left are "
                                "float_atomic_bytes"})
    proj_ms = kd.ms_per_step("gather_project_bwd", DIAG_STEPS)
    if proj_ms > 0:
        kernels.append({"kernel": "gather_bwd_proj_kernel (svr_gather_project_bwd: fc_0's input gradient rows scattered into "
                                  "(voxel, displacement, 256) slabs for the two 128-channel levels, incl. the slab memset)",
                        "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBPS,
                        "achieved": (2 * proj_hbm / (proj_ms * 1e-3) / 1e9) if proj_hbm else None,
                        "frac": (2 * proj_hbm / (proj_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if proj_hbm else None,
                        "float_atomic_bytes_per_step": (2 * proj_atomic) if proj_atomic else None,
                        "ms_per_step": proj_ms, "launches_per_step": len(kd.ev["gather_project_bwd"]) / DIAG_STEPS,
                        "algorithmic_bytes_per_step": 2 * npts * 7 * (256 * 4 + 12), "traffic": (2 * proj_hbm) if proj_hbm else None,
                        "note": "reads each dh0 row (1 KB) once per displacement and level; float atomics only at run ends"})
    # every GEMM of the step, keyed by the shape it was CALLED with: point-MLP layers have M = points, the two
    # projected levels add voxel-row GEMMs (M = B*S^3); fc_0's backward runs over the kept columns only
    what = {"linear_fwd": "forward (f16x3)", "linear_bwd_data": "dX", "linear_bwd_weight": "dW + bias gradient"}
    mlp_ms, enc_ms = 0.0, {"stage1_fwd": 0.0, "stage1_bwd": 0.0, "conv_fwd": 0.0, "conv_in_fwd": 0.0, "conv_bwd_data": 0.0,
                           "conv_bwd_weight": 0.0, "bn_fwd": 0.0, "bn_bwd": 0.0}
    for key in sorted(kd.ev):
        op, _, shape = key.partition(":")
        calls = len(kd.ev[key]) / DIAG_STEPS
        ms = kd.ms_per_step(key, DIAG_STEPS)
        if ms <= 0:
            continue
        if op in what:
            M, N, K = (int(v) for v in shape.split("x"))
            mlp_ms += ms
            tf = 2.0 * M * N * K * calls * SPLIT_PRODUCTS / (ms * 1e-3) / 1e12
            who = "point-MLP" if M == npts else "projected-level voxel"
            kernels.append({"kernel": f"{who} GEMM {what[op]} M={M} N={N} K={K}", "bound": "mfma", "unit": "TFLOP/s",
                            "peak": MFMA_F16_PEAK_TFLOPS, "achieved": tf, "frac": tf / MFMA_F16_PEAK_TFLOPS,
                            "ms_per_step": ms, "calls_per_step": calls, "f32_equivalent_TFLOPs": tf / SPLIT_PRODUCTS})
        elif op in ("conv_fwd", "conv_bwd_data", "conv_bwd_weight", "conv_in_fwd"):
            # encoder convolutions (40.8 GFLOP / sample forward at 128^3): 2 * 27 * Ci * Co FLOP per voxel,
            # three split products issued per f32 product except conv_in's exact-f32 MFMA form
            vol, ch = shape.split(":")
            Bv, Dv, Hv, Wv = (int(v) for v in vol.split("x"))
            ci, co = (int(v) for v in ch.split("->"))
            enc_ms[op] += ms
            flop = 2.0 * 27 * ci * co * Bv * Dv * Hv * Wv * calls
            if op == "conv_in_fwd":          # Ci = 1: HBM bound (writes the 16-channel volume), exact f32
                byts = Bv * Dv * Hv * Wv * (1 + co) * 4 * calls
                rate = byts / (ms * 1e-3) / 1e9
                kernels.append({"kernel": f"encoder conv_in forward + BatchNorm statistics {vol} 1->{co}", "bound": "hbm", "unit": "GB/s",
                                "peak": HBM_PEAK_GBPS, "achieved": rate, "frac": rate / HBM_PEAK_GBPS, "ms_per_step": ms,
                                "calls_per_step": calls, "algorithmic_bytes_per_step": byts})
                continue
            split = 1 if (ci == 1 or co == 1) else SPLIT_PRODUCTS
            tf = flop * split / (ms * 1e-3) / 1e12
            names = {"conv_fwd": "forward (f16x3)", "conv_bwd_data": "backward-data", "conv_bwd_weight": "weight gradient"}
            kernels.append({"kernel": f"encoder conv {names[op]} {vol} {ci}->{co}", "bound": "mfma", "unit": "TFLOP/s",
                            "peak": MFMA_F16_PEAK_TFLOPS if split > 1 else 157.3, "achieved": tf,
                            "frac": tf / (MFMA_F16_PEAK_TFLOPS if split > 1 else 157.3), "ms_per_step": ms, "calls_per_step": calls,
                            "f32_equivalent_TFLOPs": tf / split})
        elif op in ("stage1_fwd", "stage1_bwd"):
            # recomputed first stage (stage1.hip): HBM bound by construction.  Algorithmic bytes per voxel of the 16-channel
            # volume: forward = x twice (4 B each) + y written (64 B) + pooled and argmax (64 / 8 + 16 / 8 B); backward = x twice
            # + dy twice (64 B each) + dpooled and argmax twice ((64 + 16) / 8 B each)
            dims = [int(v) for v in shape.split("x")]
            vox = dims[0] * dims[1] * dims[2] * dims[3]
            enc_ms[op] += ms
            per_vox = (8 + 64 + 8 + 2) if op == "stage1_fwd" else (8 + 128 + 20)
            byts = vox * per_vox * calls
            # counter-measured HBM bytes of the two passes (profiles/gather_traffic.json, same workload) where available
            tkeys = ("stage1_stats_hbm_bytes", "stage1_apply_hbm_bytes") if op == "stage1_fwd" else ("stage1_bwd_reduce_hbm_bytes", "stage1_bwd_apply_hbm_bytes")
            pmc = sum(t[k] for k in tkeys) * calls if all(k in t for k in tkeys) else None
            rate = (pmc if pmc else byts) / (ms * 1e-3) / 1e9
            what1 = ("forward: conv_in -> ReLU -> BatchNorm statistics, then again -> y, pooled, argmax" if op == "stage1_fwd" else
                     "backward: BatchNorm reduce, then BatchNorm apply -> conv_in weight / bias gradient (no dconv tensor)")
            kernels.append({"kernel": f"encoder stage 1 recomputed ({what1}) {shape}", "bound": "hbm", "unit": "GB/s",
                            "peak": HBM_PEAK_GBPS, "achieved": rate, "frac": rate / HBM_PEAK_GBPS, "ms_per_step": ms,
                            "calls_per_step": calls, "algorithmic_bytes_per_step": byts, "traffic": pmc,
                            "traffic_source": t.get("source") if pmc else None,
                            "mfma_f32_TFLOPs": (vox * (2 if op == "stage1_fwd" else 2 + 2 * 32 / 27.0) * 2 * 27 * 16 * calls) / (ms * 1e-3) / 1e12})
        elif op in ("bn_fwd", "bn_bwd"):
            # BatchNorm + pool passes: HBM bound.  Algorithmic bytes per element of the (B,D,H,W,C) volume: forward =
            # statistics read (skipped where the producing conv delivers them) + apply read + write (+ pooled write / 8 and
            # its argmax); backward = reduce pass (x, dy reads) + apply pass (x, dy reads, dx write)
            dims = [int(v) for v in shape.split("x")]
            elems = 1
            for v in dims:
                elems *= v
            enc_ms[op] += ms
            per_elem = (4 + 4 + 4 + 0.5 + 0.125) if op == "bn_fwd" else (8 + 8 + 4)
            byts = elems * per_elem * calls
            rate = byts / (ms * 1e-3) / 1e9
            kernels.append({"kernel": f"encoder BatchNorm{'+pool forward' if op == 'bn_fwd' else ' backward (+unpool, +ReLU mask)'} {shape}",
                            "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBPS, "achieved": rate, "frac": rate / HBM_PEAK_GBPS,
                            "ms_per_step": ms, "calls_per_step": calls, "algorithmic_bytes_per_step": byts})
    res["mlp_gemm_ms_per_step"] = mlp_ms
    res["encoder_ms_per_step"] = dict(enc_ms, total=sum(enc_ms.values()),
                                      note="single-stream steps; conv FLOPs: SURVEY 8(a6), 40.8 GFLOP / sample forward")
    res["roofline_kernels_note"] = ("MFMA entries: achieved counts the 3 split products actually issued on the f16 / bf16 "
                                    "matrix cores (the f32-equivalent rate is a third of it); ms = HIP-event brackets "
                                    f"around the C-ABI calls, from {DIAG_STEPS} untimed single-stream steps behind the timed "
                                    "region (in the measured step the backward runs on three streams and the brackets "
                                    "of concurrently running kernels stretch)")
    res["roofline_kernels"] = kernels
    return res


if __name__ == "__main__":
    main()
