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


def step():
    opt.zero_grad(set_to_none=True)
    loss = tr.training_step(batch, 0)["loss"]
    loss.backward()
    opt.step()
    return loss


if a.graph:
    from svr_amd.graphs import GraphedStep
    gs = GraphedStep(tr, opt, batch, warmup=a.warmup)

    def step():                      # noqa: F811
        return gs.run(batch)
for _ in range(a.warmup):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
print(json.dumps({"workload": f"BASELINE configs[4] per-GPU shard: UNet -> unproject -> project(128^3) -> IF-Net, batch {a.batch}, "
                              f"{a.points} points, fwd+bwd+Adam", "ms_per_step": dt * 1e3,
                  "query_points_per_s": a.batch * a.points / dt, "loss": float(loss), "unet_backend": a.unet_backend,
                  "hip_graph": bool(a.graph)}))
