"""Where do the fused kernel's kept columns differ from the unfused gather's?  (row in tile, level, displacement, channel)"""
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
for rep in range(2):
    h0, kept = ops.gather_fc0_fwd(vols, pts, layout, disp, False, w, bias, keep_levels=(0, 1, 2, 3, 4, 5))
    torch.cuda.synchronize()
    for l, c in enumerate(chans):
        a = rows[:, layout.col[l]:layout.col[l] + 7 * c].view(N, 7, c)
        b = kept[:, layout.col[l]:layout.col[l] + 7 * c].view(N, 7, c)
        bad = (a != b)
        if bad.any():
            idx = bad.nonzero()
            r, j, ch = idx[:, 0], idx[:, 1], idx[:, 2]
            print(f"rep {rep} level {l} (C={c}): {int(bad.sum())} of {bad.numel()} differ; rows%64: {sorted(set((r % 64).tolist()))[:20]} j: {sorted(set(j.tolist()))} "
                  f"ch%4: {sorted(set((ch % 4).tolist()))} ch//4: {sorted(set((ch // 4).tolist()))[:12]} first: row {int(r[0])} j {int(j[0])} ch {int(ch[0])} {float(a[r[0], j[0], ch[0]])} vs {float(b[r[0], j[0], ch[0]])}")
        else:
            print(f"rep {rep} level {l}: equal")
