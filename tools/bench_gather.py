"""Micro-benchmark of the gather forward / backward per level at the config-3 shape (B=8, 128^3, N=50k)."""
import os, sys, time
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
if "--surface" in sys.argv:      # the clustered distribution of bench.py --dist surface (points around random planes)
    from bench import synth_batch
    pts = synth_batch(103, B, 8, N, dev, "surface")["points"].contiguous()
else:
    pts = torch.rand(B, N, 3, device=dev) - 0.5
order = ops.morton_order(pts).long()
pts_sorted = pts.reshape(-1, 3)[order].view(B, N, 3).contiguous()
layout = ops.FeatureLayout(chans)
disp = 0.0722


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


feat = torch.empty(B * N, layout.row_stride, device=dev)
gfeat = torch.randn(B * N, layout.row_stride, device=dev)
if "--fwd-only" in sys.argv:
    for name, p in (("random", pts), ("sorted", pts_sorted)):
        print(name, "fwd all levels: %.3f ms" % timeit(lambda: ops.gather_fwd(vols, p, layout, disp, False, out=feat), n=10))
    sys.exit(0)
for name, p in (() if "--pull-only" in sys.argv else (("random", pts), ("sorted", pts_sorted))):
    print(name, "fwd all levels: %.3f ms" % timeit(lambda: ops.gather_fwd(vols, p, layout, disp, False, out=feat)))
    for l in range(1, 6):
        gv = [None] * 6
        gv[l] = torch.zeros_like(vols[l])
        t = timeit(lambda: ops.gather_bwd(vols, gv, p, gfeat, layout, disp, False))
        print(f"  bwd level {l} (C={chans[l]}, S={vols[l].shape[1]}): {t:.3f} ms")
    gv = [None] + [torch.zeros_like(v) for v in vols[1:]]
    print("  bwd levels 1-5 fused: %.3f ms" % timeit(lambda: ops.gather_bwd(vols, gv, p, gfeat, layout, disp, False)))
    lo = [None] * 6
    for l in (3, 4, 5):
        lo[l] = ops.voxel_order(p, tuple(vols[l].shape[1:4]))
    print("  voxel_order x3: %.3f ms" % timeit(lambda: [ops.voxel_order(p, tuple(vols[l].shape[1:4])) for l in (3, 4, 5)]))
    for l in (3, 4, 5):
        gv1 = [None] * 6
        gv1[l] = torch.zeros_like(vols[l])
        print(f"  bwd level {l} with its voxel order: %.3f ms" % timeit(lambda: ops.gather_bwd(vols, gv1, p, gfeat, layout, disp, False, level_orders=lo)))
    print("  bwd levels 1-5 fused + level orders: %.3f ms" % timeit(lambda: ops.gather_bwd(vols, gv, p, gfeat, layout, disp, False, level_orders=lo)))

# ---- pull form (atomic-free) vs atomics, per level, on the Morton-sorted point set (what the training step sees)
p = pts_sorted
print("pull-form scatter (sorted points), SVR_PULL_VARIANT =", os.environ.get("SVR_PULL_VARIANT", "0"))
plans = [None] * 6
for l in (1, 2, 3):
    dims = tuple(vols[l].shape[1:4])
    t = timeit(lambda: ops.pull_plan(p, dims, chans[l], layout.col[l], layout.row_stride, disp, False))
    plans[l] = ops.pull_plan(p, dims, chans[l], layout.col[l], layout.row_stride, disp, False)
    pl = [None] * 6
    pl[l] = plans[l]
    gv1 = [None] * 6
    gv1[l] = torch.empty_like(vols[l])
    tk = timeit(lambda: ops.gather_bwd(vols, gv1, p, gfeat, layout, disp, False, level_plans=pl))
    print(f"  level {l} (C={chans[l]}, S={dims[0]}): plan {t:.3f} ms, pull scatter {tk:.3f} ms, longest walk {int(plans[l].stats[0])}")
lo = [None] * 6
for l in (1, 2, 3, 4, 5):
    dims = tuple(vols[l].shape[1:4])
    t = timeit(lambda: ops.item_order(p, dims, disp, False))
    lo[l] = ops.item_order(p, dims, disp, False)
    if l <= 3:
        plans_keep = plans[l]
    l1 = [None] * 6
    l1[l] = lo[l]
    gv1 = [None] * 6
    gv1[l] = torch.zeros_like(vols[l])
    tk = timeit(lambda: ops.gather_bwd(vols, gv1, p, gfeat, layout, disp, False, level_orders=l1))
    print(f"  level {l} (C={chans[l]}, S={dims[0]}): item order {t:.3f} ms, atomic scatter over items {tk:.3f} ms")
lo45 = [None, None, None, None, lo[4], lo[5]]
gv = [None] + [torch.empty_like(v) if plans[l] is not None else torch.zeros_like(v) for l, v in enumerate(vols) if l >= 1]
gz = [None] + [torch.zeros_like(v) for v in vols[1:]]
print("  all levels atomic over item orders: %.3f ms" % timeit(lambda: ops.gather_bwd(vols, gz, p, gfeat, layout, disp, False, level_orders=lo)))
lo = lo45
print("  levels 1-3 pull + 4-5 atomic (item orders): %.3f ms" % timeit(
    lambda: ops.gather_bwd(vols, gv, p, gfeat, layout, disp, False, level_orders=lo, level_plans=plans)))
