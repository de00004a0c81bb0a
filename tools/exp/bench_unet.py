"""UNet (reference model/unet.py:15-118) fwd / fwd+bwd time at the config-5 shard shape (batch 4, 3x256x256, 1 output channel),
hand-kernel backend; per-layer HIP-event brackets of the conv blocks with --layers.
   python tools/exp/bench_unet.py [--batch 4] [--reps 20] [--layers] [--backend hip|stock]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch

import svr_amd  # noqa: F401
from svr_amd.model import unet as U

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--layers", action="store_true")
ap.add_argument("--backend", default="hip")
a = ap.parse_args()
dev = torch.device("cuda")
torch.manual_seed(0)
net = U.Unet(channels_in=3, channels_out=1, backend=a.backend).to(dev).train()
x = (torch.rand(a.batch, 3, 256, 256, device=dev) * 2 - 1)
tgt = torch.rand(a.batch, 1, 256, 256, device=dev)


def fwd():
    with torch.no_grad():
        return net(x)


def step():
    for p in net.parameters():
        p.grad = None
    y = net(x)
    loss = (y - tgt).abs().mean()
    loss.backward()
    return loss


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, (time.perf_counter() - t0) / reps * 1e3


out = {"batch": a.batch, "backend": a.backend}
out["fwd_ms_gpu"], out["fwd_ms_wall"] = timeit(fwd, a.reps)
out["step_ms_gpu"], out["step_ms_wall"] = timeit(step, a.reps)
if a.layers and a.backend == "hip":
    ev = []
    from svr_amd import ops
    F = U._ConvBlockIgemmFn if (ops.UNET_IGEMM and ops.BACKWARD_GEMM == "f16x3s") else U._ConvBlockFn
    of, ob = F.forward, F.backward

    def tf(ctx, *args):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = of(ctx, *args)
        e1.record()
        ev.append(("fwd", tuple(args[0].shape), tuple(args[2].shape), e0, e1))
        return r

    def tb(ctx, dy):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = ob(ctx, dy)
        e1.record()
        ev.append(("bwd", tuple(dy.shape), ctx.cfg[-1], e0, e1))
        return r
    F.forward, F.backward = staticmethod(tf), staticmethod(tb)
    step()
    torch.cuda.synchronize()
    ev.clear()
    step()
    torch.cuda.synchronize()
    out["layers"] = [{"pass": p, "shape": s, "weight": w, "ms": round(e0.elapsed_time(e1), 4)} for p, s, w, e0, e1 in ev]
    out["conv_blocks_ms"] = round(sum(r["ms"] for r in out["layers"]), 3)
print(json.dumps(out))
