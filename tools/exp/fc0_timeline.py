"""In-kernel timeline of the fused gather -> fc_0 kernel (measurement build: SVR_FC0_MEASURE=1).  Lane 0 of every wave of 8 tiles
in the middle of the launch stamps s_memtime at: producers -- slab loop top (1), produce() done (2), barrier passed (3);
consumers -- every k-step (20), slab consumed (21), barrier passed (22), main loop done (23), epilogue done (24).
Prints, per role, where a tile's cycles go."""
import ctypes as C
import os
import sys
import collections

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import svr_amd  # noqa
from svr_amd import ops, _lib

B, N, D = 8, 50000, 128
chans = [1, 16, 32, 64, 128, 128]
torch.manual_seed(0)
vols, d = [], D
for i, c in enumerate(chans):
    vols.append(torch.randn(B, d, d, d, c, device="cuda"))
    if i >= 1:
        d //= 2
pts = torch.rand(B, N, 3, device="cuda") - 0.5
if "same" in sys.argv[1:]:
    pts = torch.zeros_like(pts) + torch.tensor([0.1, 0.2, 0.3], device="cuda")
_, pts = ops.morton_order(pts.contiguous(), want_sorted=True)
layout = ops.FeatureLayout(chans)
disp = float(np.float32(0.0722))
w = torch.randn(256, layout.row_stride, device="cuda") / 30
w[:, layout.width:] = 0
bias = torch.randn(256, device="cuda")
TILES, NST = 8, 1024
buf = torch.zeros(TILES * 8 * NST, dtype=torch.int64, device="cuda")
lib = _lib.lib()
f = C.CDLL(_lib.LIB_PATH).svr_gather_fc0_stamps      # (same library instance: dlopen of a loaded path returns it)
f.argtypes = [C.c_void_p]
for _ in range(3):
    ops.gather_fc0_fwd(vols, pts, layout, disp, False, w, bias, keep_levels=(0, 1, 2, 3))
torch.cuda.synchronize()
assert f(C.c_void_p(buf.data_ptr())) == 0
ops.gather_fc0_fwd(vols, pts, layout, disp, False, w, bias, keep_levels=(0, 1, 2, 3))
torch.cuda.synchronize()
f(C.c_void_p(0))
st = buf.cpu().numpy().astype(np.uint64).reshape(TILES, 8, NST)
MASK = (1 << 48) - 1
for tile in range(TILES):
    t0 = min(int(st[tile, w_, 0] & MASK) for w_ in range(8) if st[tile, w_, 0])
    for wave in (0, 4):
        ev = [(int(v >> 56), int((v >> 48) & 255), int(v & MASK) - t0) for v in st[tile, wave] if v]
        role = "consumer" if wave < 4 else "producer"
        tot = ev[-1][2] - ev[0][2]
        acc = collections.Counter()
        prev = ev[0]
        for e in ev[1:]:
            acc[(prev[0], e[0])] += e[2] - prev[2]
            prev = e
        print(f"tile {tile} wave {wave} ({role}): {tot} cycles; " + ", ".join(f"{a}->{b}: {v} ({100 * v / tot:.0f}%)" for (a, b), v in sorted(acc.items())))
        if tile == 2:
            # per slab detail
            per = collections.defaultdict(dict)
            prev = ev[0]
            for e in ev[1:]:
                per[e[1]][(prev[0], e[0])] = per[e[1]].get((prev[0], e[0]), 0) + e[2] - prev[2]
                prev = e
            for s_ in sorted(per):
                print("    slab", s_, {f"{a}->{b}": v for (a, b), v in per[s_].items()})
