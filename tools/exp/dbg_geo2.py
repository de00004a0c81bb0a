"""Measurement build with -DFC_GEO_CHECK=1: the kernel compares every corner offset from the LDS geometry table with the register
(ds_bpermute) form and records mismatches."""
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
buf = torch.zeros(8 + 200 * 4, dtype=torch.int64, device="cuda")
lib.svr_gather_fc0_stamps(C.c_void_p(buf.data_ptr()))
rows = ops.gather_fwd(vols, pts, layout, disp, False)
h0, kept = ops.gather_fc0_fwd(vols, pts, layout, disp, False, w, bias, keep_levels=(0, 1, 2, 3, 4, 5))
torch.cuda.synchronize()
print("kept columns equal to the unfused gather:", bool(torch.equal(rows[:, :layout.width], kept[:, :layout.width])))
lib.svr_gather_fc0_stamps(C.c_void_p(0))
b = buf.cpu().numpy().view(np.uint64)
n = int(b[0])
print("mismatches:", n)
for k in range(min(n, 40)):
    o = b[8 + 4 * k: 12 + 4 * k]
    tile = int(o[0] >> np.uint64(32)); w0 = int(o[0] & np.uint64(0xffffffff))
    print(f"tile {tile} level {w0 >> 24} pw {(w0 >> 16) & 255} lane {(w0 >> 8) & 255} it {(w0 >> 4) & 15} value#{w0 & 15} lds {int(o[1] >> np.uint64(32))} reg {int(o[1] & np.uint64(0xffffffff))} j0 {int(o[2] >> np.uint64(32))} LP/NJ {int(o[2]) >> 8 & 255}/{int(o[2]) & 255}")
