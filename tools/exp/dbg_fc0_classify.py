"""For every (row, displacement) item of the fused kernel's kept columns that differs from the unfused gather: does it hold the
values of ANOTHER item (row', j') of the same level (-> the lane used another item's geometry), a partial sum, or garbage?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import svr_amd  # noqa
from svr_amd import ops
B, N, D = 1, 640, 32
chans = [1, 16, 32, 64, 128, 128]
torch.manual_seed(0)
vols, d = [], D
for i, c in enumerate(chans):
    vols.append(torch.randn(B, d, d, d, c, device="cuda"))
    if i >= 1:
        d //= 2
pts = torch.rand(B, N, 3, device="cuda") - 0.5
_, pts = ops.morton_order(pts.contiguous(), want_sorted=True)
layout = ops.FeatureLayout(chans)
disp = float(np.float32(0.0722))
w = torch.randn(256, layout.row_stride, device="cuda") / 30
w[:, layout.width:] = 0
bias = torch.randn(256, device="cuda")
rows = ops.gather_fwd(vols, pts, layout, disp, False)
h0, kept = ops.gather_fc0_fwd(vols, pts, layout, disp, False, w, bias, keep_levels=(0, 1, 2, 3, 4, 5))
torch.cuda.synchronize()
for l, c in enumerate(chans):
    if c < 16:
        continue
    a = rows[:, layout.col[l]:layout.col[l] + 7 * c].view(N, 7, c)
    b = kept[:, layout.col[l]:layout.col[l] + 7 * c].view(N, 7, c)
    bad = (a != b).any(2)
    if not bad.any():
        print(f"level {l}: equal")
        continue
    flat = a.reshape(N * 7, c)
    out = []
    for r, j in bad.nonzero().tolist()[:40]:
        v = b[r, j]
        nch = int((a[r, j] != v).sum())
        m = (flat == v).all(1).nonzero().flatten().tolist()
        src = [(x // 7, x % 7) for x in m][:3]
        out.append(f"(row {r} [{r % 64} in tile, lane-row {r % 16}] j {j}: {nch}/{c} ch differ; equals item {src})")
    print(f"level {l} (C={c}): {int(bad.sum())} bad items of {N * 7}:\n   " + "\n   ".join(out[:24]))
