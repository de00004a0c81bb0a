"""Fused gather->fc_0 (gather_fc0.hip) against the two separate kernels at the config-3 shape (128^3, 50k points, B=8)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import svr_amd  # noqa
from svr_amd import ops

B, N, D = 8, 50000, 128
chans = [1, 16, 32, 64, 128, 128]
torch.manual_seed(0)
vols, d = [], D
for i, c in enumerate(chans):
    vols.append(torch.randn(B, d, d, d, c, device="cuda"))
    if i >= 1:
        d //= 2
dist = sys.argv[1] if len(sys.argv) > 1 else "uniform"
pts = torch.rand(B, N, 3, device="cuda") - 0.5
if dist == "surface":
    v = torch.randn(B, N, 3, device="cuda")
    pts = (v / v.norm(dim=2, keepdim=True) * 0.35 + 0.01 * torch.randn(B, N, 3, device="cuda")).clamp(-0.5, 0.5)
if dist == "tiny":      # all points inside a cube of edge 0.1: every level's working set is L2 resident (latency experiment)
    pts = pts * 0.1
if dist == "same":      # one point: every load hits the same lines (issue / LDS / barrier / MFMA floor of the kernel)
    pts = torch.zeros_like(pts) + torch.tensor([0.1, 0.2, 0.3], device="cuda")
_, pts = ops.morton_order(pts.contiguous(), want_sorted=True)
layout = ops.FeatureLayout(chans)
disp = float(np.float32(0.0722))
w = torch.randn(256, layout.row_stride, device="cuda") / 30
w[:, layout.width:] = 0
bias = torch.randn(256, device="cuda")
# DVFS experiments (MI355X_MICROARCH.md "DVFS give-back": matrix instructions on random operands lower the clock the chip
# holds; on zeros they do not): zero weights / zero volumes keep every instruction and every address, only the operand bits change
if "zerow" in sys.argv[2:]:
    w.zero_()
if "zerov" in sys.argv[2:]:
    for v in vols:
        v.zero_()


def timeit(f, n=10):
    f(); f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


rows = [None]
def sep():
    rows[0] = ops.gather_fwd(vols, pts, layout, disp, False)
    return ops.linear_fwd(rows[0], w, bias, relu=True)
t_g = timeit(lambda: ops.gather_fwd(vols, pts, layout, disp, False))
t_sep = timeit(sep)
t_inf = timeit(lambda: ops.gather_fc0_fwd(vols, pts, layout, disp, False, w, bias))
t_keep = timeit(lambda: ops.gather_fc0_fwd(vols, pts, layout, disp, False, w, bias, keep_levels=(0, 1, 2, 3)))
a = sep()
b, _ = ops.gather_fc0_fwd(vols, pts, layout, disp, False, w, bias)
print(f"{dist}: gather {t_g:.3f} ms, gather+fc_0 separate {t_sep:.3f} ms, fused (no rows) {t_inf:.3f} ms, fused (levels 0-3 kept) {t_keep:.3f} ms, "
      f"max dev {float((a - b).abs().max() / a.abs().max()):.2e}")
