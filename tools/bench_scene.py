"""Config-5 per-GPU shard (BASELINE configs[4]) as a timed step: rgb (4,3,256,256) -> UNet (stock
MIOpen ops) -> unproject -> project(128^3, kernel 3, sigma 1.5) -> IF-Net, 50 000 points, fwd + bwd + Adam.
usage: python tools/bench_scene.py [--steps K] [--warmup W] [--subsample-points n]   -> one JSON line"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import svr_amd  # noqa: E402,F401
from oracle import ifnet_oracle as O  # noqa: E402  (name-seeded weights only)
from oracle import scene_oracle as S  # noqa: E402
from svr_amd.trainer import SceneNetTrainer, default_hparams  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=8, help="the step arena and the scatter-form decision (fixed lag 3) settle in the first steps")
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--points", type=int, default=50000)
ap.add_argument("--gc", default="default", choices=["default", "freeze", "off"],
                help="Python's cyclic collector over the timed steps: untouched, gc.freeze() after the warm-up (the modules / "
                     "graphs / caches alive by then leave the collector's generations), or disabled")
ap.add_argument("--inflight", type=int, default=2, help="steps the host may run ahead of the GPU (0: unbounded)")
ap.add_argument("--graph", action="store_true", help="capture the whole step into a HIP graph (svr_amd.graphs.GraphedStep)")
ap.add_argument("--unet-backend", default="hip", choices=["hip", "stock"])
ap.add_argument("--miopen-benchmark", action="store_true", help="torch.backends.cudnn.benchmark: let MIOpen time its solvers")
a = ap.parse_args()
torch.backends.cudnn.benchmark = bool(a.miopen_benchmark)
dims = (128, 128, 128)
g = torch.Generator(device="cpu").manual_seed(105)
rgb = torch.rand(a.batch, 3, 256, 256, generator=g) * 2 - 1
target = torch.rand(a.batch, 240, 320, generator=g) * 5 + 0.5
pts = torch.rand(a.batch, a.points, 3, generator=g) - 0.5
occ = (torch.rand(a.batch, a.points, generator=g) < 0.5).float()
tr = SceneNetTrainer(default_hparams(), dims=dims)
tr.unet.load_state_dict(S.name_seeded_like(tr.unet.state_dict(), 1.0, "unet."), strict=False)
tr.ifnet.load_state_dict(O.name_seeded_state(128), strict=False)
tr = tr.cuda().train()
tr.unet.backend = a.unet_backend
if a.graph:
    h = tr.hparams
    opt = torch.optim.Adam([{"params": tr.unet.parameters(), "lr": h.lr}, {"params": tr.project.parameters(), "lr": 10 * h.lr},
                            {"params": tr.ifnet.parameters()}], lr=h.lr, capturable=True)
else:
    opt = tr.configure_optimizers()[0][0]
batch = {"rgb": rgb.cuda(), "depthmap_target": target.cuda(), "points": pts.cuda(), "occupancies": occ.cuda()}


phase_ms = []      # host time of the eager step's phases (forward, backward, optimizer): where a slow step lost its time


def step():
    t = [time.perf_counter()]
    opt.zero_grad(set_to_none=True)
    loss = tr.training_step(batch, 0)["loss"]
    t.append(time.perf_counter())
    loss.backward()
    t.append(time.perf_counter())
    opt.step()
    t.append(time.perf_counter())
    phase_ms.append([round((b - a_) * 1e3, 2) for a_, b in zip(t, t[1:])])
    return loss.detach()      # (not the graph: a kept loss tensor keeps its step's activations alive while the next step allocates)


if a.graph:
    from svr_amd.graphs import GraphedStep
    gs = GraphedStep(tr, opt, batch, warmup=a.warmup)

    def step():                      # noqa: F811
        return gs.run(batch)
for _ in range(a.warmup):
    step()
torch.cuda.synchronize()
import gc  # noqa: E402
if a.gc == "freeze":
    gc.collect()
    gc.freeze()
elif a.gc == "off":
    gc.collect()
    gc.disable()
gc_events = []
_gc_t = [0.0]


def _gc_cb(phase, info):      # how long each collection of the timed region took, and of which generation
    if phase == "start":
        _gc_t[0] = time.perf_counter()
    else:
        gc_events.append((info["generation"], round((time.perf_counter() - _gc_t[0]) * 1e3, 2)))


gc.callbacks.append(_gc_cb)
# every timed step on its own: HIP events at the step boundaries, the host's enqueue time and the caching allocator's
# hipMalloc count per step (the only kept round-3 artefact said 106.5 ms eager and nobody could tell why)
dev = torch.device("cuda")
marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
done = []
host_ms, mallocs = [], []
m_prev = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
t0 = time.perf_counter()
marks[0].record()
for i in range(a.steps):
    if a.inflight and len(done) >= a.inflight:      # like DataParallelTrainer: at most `inflight` steps queued ahead of the GPU
        done[-a.inflight].synchronize()
    h0 = time.perf_counter()
    loss = step()
    marks[i + 1].record()
    done.append(marks[i + 1])
    host_ms.append((time.perf_counter() - h0) * 1e3)
    m = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
    mallocs.append(m - m_prev)
    m_prev = m
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps)]
srt = sorted(step_ms)
print(json.dumps({"workload": f"BASELINE configs[4] per-GPU shard: UNet -> unproject -> project(128^3) -> IF-Net, batch {a.batch}, "
                              f"{a.points} points, fwd+bwd+Adam", "ms_per_step": dt * 1e3,
                  "query_points_per_s": a.batch * a.points / dt, "loss": float(loss), "unet_backend": a.unet_backend,
                  "hip_graph": bool(a.graph), "steps": a.steps, "warmup": a.warmup, "inflight": a.inflight,
                  "step_ms": {"min": srt[0], "median": srt[len(srt) // 2], "max": srt[-1]},
                  "step_ms_list": [round(v, 2) for v in step_ms],
                  "host_enqueue_ms": {"median": sorted(host_ms)[len(host_ms) // 2], "max": max(host_ms)},
                  "host_ms_list": [round(v, 1) for v in host_ms],
                  "host_phase_ms_fwd_bwd_opt": phase_ms[-a.steps:] if not a.graph else None,
                  "hipMalloc_calls_per_step": mallocs, "gc": a.gc,
                  "gc_collections_ms": {"gen2": [t for g_, t in gc_events if g_ == 2],
                                        "gen1_total": round(sum(t for g_, t in gc_events if g_ == 1), 2),
                                        "gen0_total": round(sum(t for g_, t in gc_events if g_ == 0), 2)},
                  "reserved_GB": torch.cuda.memory_stats(dev).get("reserved_bytes.all.current", 0) / 1e9}))
