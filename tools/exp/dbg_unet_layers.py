"""Each conv block of the full UNet at (B, 256, 256) through the implicit-GEMM path against stock torch ops on the GPU."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import torch.nn.functional as F

import svr_amd  # noqa: F401
from svr_amd import ops
from svr_amd.model import unet as U

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
torch.manual_seed(0)
nf = 32
enc = [(3, nf, 256), (nf, 2 * nf, 128), (2 * nf, 4 * nf, 64), (4 * nf, 8 * nf, 32), (8 * nf, 8 * nf, 16), (8 * nf, 8 * nf, 8),
       (8 * nf, 8 * nf, 4), (8 * nf, 8 * nf, 2)]
dec = [(8 * nf, 0, 8 * nf, 1), (8 * nf, 8 * nf, 8 * nf, 2), (8 * nf, 8 * nf, 8 * nf, 4), (8 * nf, 8 * nf, 8 * nf, 8),
       (8 * nf, 8 * nf, 4 * nf, 16), (4 * nf, 4 * nf, 2 * nf, 32), (2 * nf, 2 * nf, nf, 64), (nf, nf, 1, 128)]


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


_made = []
_orig = ops.Conv2dPlanes.__init__
def _init(self, *a, **k):
    _orig(self, *a, **k)
    _made.append(self)
ops.Conv2dPlanes.__init__ = _init


def run(C0, C1, Cout, H, k, stride, act, up):
    s0 = torch.randn(B, H, H, C0, device="cuda").requires_grad_(True)
    s1 = torch.randn(B, H, H, C1, device="cuda").requires_grad_(True) if C1 else None
    w = (torch.randn(Cout, C0 + C1, k, k, device="cuda") / (k * k * (C0 + C1)) ** 0.5).requires_grad_(True)
    b = torch.randn(Cout, device="cuda").requires_grad_(True)
    y = U._ConvBlockIgemmFn.apply(s0, s1, w, b, k, stride, act, up)
    x = torch.cat((s0, s1), 3) if C1 else s0
    x = x.permute(0, 3, 1, 2).double()
    x = F.leaky_relu(x, 0.2) if act == 1 else (F.relu(x) if act == 2 else x)
    if up:
        x = F.interpolate(x, scale_factor=2, mode="bilinear")
    ref = F.conv2d(x, w.double(), b.double(), stride=stride, padding=1).permute(0, 2, 3, 1)
    pl = _made[-1]
    print("   amax word", pl.buf[:4].view(torch.int32).view(torch.float32).item(), "true", float(w.abs().max()), "buf ptr", hex(pl.buf.data_ptr()))
    e = (y.double() - ref).abs()
    bad = (e > 1e-5 * ref.abs().max()).nonzero()
    if bad.shape[0]:
        import collections
        print("   bad outputs", bad.shape[0], "of", e.numel(), "channels", sorted(collections.Counter(bad[:, 3].tolist()).items())[:20],
              "rows", sorted(collections.Counter(bad[:, 1].tolist()).items())[:10], "first", bad[:5].tolist())
    dy = torch.randn_like(y)
    ins = [t for t in (s0, s1, w, b) if t is not None]
    g = torch.autograd.grad(y, ins, dy)
    gr = torch.autograd.grad(ref, ins, dy.double())
    print(f"C0={C0} C1={C1} Cout={Cout} H={H} k={k} up={up}: y {rel(y.double(), ref):.2e}  grads " +
          " ".join(f"{rel(a.double(), r.double()):.2e}" for a, r in zip(g, gr)), flush=True)


for i, (ci, co, h) in enumerate(enc):
    run(ci, 0, co, h, 4, 2, 0 if i == 0 else 1, False)
for (c0, c1, co, h) in dec:
    run(c0, c1, co, h, 3, 1, 2, True)
