"""Micro-benchmark of the projected backward scatter (gather_bwd_proj_kernel) at the config-3 shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import svr_amd
from svr_amd import ops
B, N = 8, 50000
torch.manual_seed(0)
pts = torch.rand(B, N, 3, device="cuda") - 0.5
order = ops.morton_order(pts).long()
pts = pts.reshape(-1, 3)[order].view(B, N, 3).contiguous()
dh = torch.randn(B * N, 256, device="cuda")
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for S in (16, 8):
    items = ops.item_order(pts, (S, S, S), 0.0722, False, with_j=True)
    t_o = timeit(lambda: ops.item_order(pts, (S, S, S), 0.0722, False, with_j=True))
    t = timeit(lambda: ops.gather_project_bwd(pts, dh, (S, S, S), items, 0.0722, False))
    print(f"S={S}: item order {t_o:.3f} ms, projected scatter (incl. dP memset) {t:.3f} ms")
    plan = ops.project_plan(pts, (S, S, S), 0.0722, False)
    t_p = timeit(lambda: ops.project_plan(pts, (S, S, S), 0.0722, False))
    t2 = timeit(lambda: ops.gather_project_bwd(pts, dh, (S, S, S), plan, 0.0722, False))
    a = ops.gather_project_bwd(pts, dh, (S, S, S), items, 0.0722, False)
    b = ops.gather_project_bwd(pts, dh, (S, S, S), plan, 0.0722, False)
    slots = int(plan.sidx[-1])
    print(f"S={S}: plan {t_p:.3f} ms, two-pass scatter {t2:.3f} ms, {slots} slots of {plan.slots}, "
          f"max dev {float((a - b).abs().max() / a.abs().max()):.2e}")
