"""Does the fc_0 weight-gradient GEMM overlap with the gather backward / conv backward on a second stream?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import svr_amd
from svr_amd import ops

B, D, N = 8, 128, 50000
chans = [1, 16, 32, 64, 128, 128]
dev = "cuda"
torch.manual_seed(0)
vols, d = [], D
for i, c in enumerate(chans):
    vols.append(torch.randn(B, d, d, d, c, device=dev))
    if i >= 1:
        d //= 2
pts = torch.rand(B, N, 3, device=dev) - 0.5
order = ops.morton_order(pts).long()
p = pts.reshape(-1, 3)[order].view(B, N, 3).contiguous()
layout = ops.FeatureLayout(chans)
disp = 0.0722
gfeat = torch.randn(B * N, layout.row_stride, device=dev)
gv = [None] + [torch.zeros_like(v) for v in vols[1:]]
lo = [None] * 6
for l in (3, 4, 5):
    lo[l] = ops.voxel_order(p, tuple(vols[l].shape[1:4]))
M = B * N
x = torch.randn(M, 2592, device=dev)
dy = torch.randn(M, 256, device=dev)
xc = torch.randn(B, 64, 64, 64, 32, device=dev)
dc = torch.randn(B, 64, 64, 64, 32, device=dev)
wc = torch.randn(32, 32, 3, 3, 3, device=dev) * 0.05
side = torch.cuda.Stream()


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def gb():
    ops.gather_bwd(vols, gv, p, gfeat, layout, disp, False, level_orders=lo)


def tn():
    ops.linear_bwd_weight(dy, x)


def convs():
    for _ in range(3):
        ops.conv3d_k3_bwd_data(dc, wc)
        ops.conv3d_k3_bwd_weight(xc, dc)


def both(main_fn):
    def f():
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            tn()
        main_fn()
        torch.cuda.current_stream().wait_stream(side)
    return f


print("gather_bwd alone  %.3f ms" % timeit(gb))
print("tn (fc_0 dW) alone %.3f ms" % timeit(tn))
print("convs alone       %.3f ms" % timeit(convs))
print("gather_bwd || tn  %.3f ms" % timeit(both(gb)))
print("convs || tn       %.3f ms" % timeit(both(convs)))
