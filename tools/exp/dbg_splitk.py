"""linear_bwd_data_splitk against linear_bwd_data and float64 at the voxel-GEMM shapes; times of both."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import svr_amd  # noqa
from svr_amd import ops
torch.manual_seed(0)
for M in (4096, 32768, 262144 // 8, 8 * 4096, 8 * 512, 1000):
    N, K = 1792, 128
    dy = torch.randn(M, N, device="cuda") * 1e-4
    w = torch.randn(N, K, device="cuda") / N ** 0.5          # (N, K): linear_bwd_data's layout
    ref = dy.double() @ w.double()
    a = ops.linear_bwd_data(dy, w)
    b = ops.linear_bwd_data_splitk(dy, w.t().contiguous())
    rel = lambda x: float((x.double() - ref).abs().max() / ref.abs().max())
    am = b._svr_amax.view(torch.float32).item()
    def t(fn):
        for _ in range(3): fn()
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 20 * 1e3
    wt = w.t().contiguous()
    print(f"M={M}: rel err old {rel(a):.2e} splitk {rel(b):.2e}; amax word {am:.4e} true {float(b.abs().max()):.4e}; "
          f"us old {t(lambda: ops.linear_bwd_data(dy, w)):.0f} splitk {t(lambda: ops.linear_bwd_data_splitk(dy, wt)):.0f}", flush=True)
