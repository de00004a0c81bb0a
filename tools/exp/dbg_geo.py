"""Measurement build: dump tile 0's level-3 geometry table (FC_GEO_LDS) twice and compare with the expected entries."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import svr_amd  # noqa
from svr_amd import ops, _lib
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
bias = torch.randn(256, device="cuda")
lib = C.CDLL(_lib.LIB_PATH)
lib.svr_gather_fc0_stamps.argtypes = [C.c_void_p]
lib.svr_gather_fc0_dbg_points.argtypes = [C.c_void_p]
def expected(level, tile):
    S = vols[level].shape[1]; Cc = vols[level].shape[4]
    p = pts[0, tile * 64: tile * 64 + 64].cpu().numpy().astype(np.float32)
    exp_off = np.zeros((64, 9, 2), np.uint32); exp_w = np.zeros((64, 9, 2), np.float32)
    for r in range(64):
        for axis in range(3):
            for var in range(3):
                g = np.float32(2.0) * p[r, 2 - axis]
                if var == 1: g = np.float32(g + np.float32(-disp))
                if var == 2: g = np.float32(g + np.float32(disp))
                i = np.float32((np.float32(g + np.float32(1)) * np.float32(S) - np.float32(1)) / np.float32(2))
                i0f = np.floor(i); w0 = np.float32(np.float32(i0f + 1) - i); w1 = np.float32(i - i0f); i0 = int(i0f)
                v0 = 0 <= i0 < S; v1 = 0 <= i0 + 1 < S
                c0 = min(max(i0, 0), S - 1); c1 = min(max(i0 + 1, 0), S - 1)
                mul = Cc * 4 if axis == 0 else (S * Cc * 4 if axis == 1 else S * S * Cc * 4)
                exp_off[r, axis * 3 + var] = (c0 * mul, c1 * mul)
                exp_w[r, axis * 3 + var] = (w0 if v0 else 0, w1 if v1 else 0)
    return exp_off, exp_w

lib.svr_gather_fc0_dbg_level.argtypes = [C.c_int]
for level in (2, 3, 4):
    for tile in range(0, 10):
        buf = torch.zeros(64 * 36 + 16, dtype=torch.int32, device="cuda")
        lib.svr_gather_fc0_stamps(C.c_void_p(buf.data_ptr()))
        lib.svr_gather_fc0_dbg_points(C.c_void_p(pts.data_ptr() + tile * 64 * 12))
        lib.svr_gather_fc0_dbg_level(chans[level] if level < 5 else -1)
        ops.gather_fc0_fwd(vols, pts, layout, disp, False, w, bias)
        torch.cuda.synchronize()
        lib.svr_gather_fc0_stamps(C.c_void_p(0))
        t = buf[:64 * 36].cpu().numpy().view(np.uint32).reshape(64, 36).copy()
        exp_off, exp_w = expected(level, tile)
        off = t[:, :18].reshape(64, 9, 2); wt = t[:, 18:].view(np.float32).reshape(64, 9, 2)
        bo = np.argwhere(off != exp_off); bw = np.argwhere(wt != exp_w)
        if len(bo) or len(bw):
            print(f"level {level} tile {tile}: offsets wrong at {len(bo)} entries, weights at {len(bw)}; rows {sorted(set(bo[:, 0].tolist()) | set(bw[:, 0].tolist()))} "
                  f"entries {sorted(set(bo[:, 1].tolist()) | set(bw[:, 1].tolist()))}")
            if len(bw):
                r, a, c = bw[0]; print("    weight", r, a, c, wt[r, a], exp_w[r, a])
        else:
            print(f"level {level} tile {tile}: table as expected")
