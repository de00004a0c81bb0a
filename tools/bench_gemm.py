"""Micro-benchmark of the MLP GEMMs at the config-3 shape (M = 400 000 rows)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import svr_amd
from svr_amd import ops

M = 400000
dev = "cuda"
torch.manual_seed(0)


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for (N, K) in ((256, 2592), (256, 256)):
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev)
    dy = torch.randn(M, N, device=dev)
    fl = 2.0 * M * N * K / 1e12
    for mode in ("f32", "bf16x6", "f16x3"):
        t = timeit(lambda: ops.linear_fwd(x, w, b, relu=True, mode=mode))
        print(f"N={N} K={K}: fwd      {mode:7s} {t:7.3f} ms  {fl/t*1e3:7.1f} TF/s")
    for mode in ("f32", "bf16x3"):
        t = timeit(lambda: ops.linear_bwd_data(dy, w, mask=x, mode=mode))
        print(f"N={N} K={K}: bwd_data {mode:7s} {t:7.3f} ms  {fl/t*1e3:7.1f} TF/s")
        t = timeit(lambda: ops.linear_bwd_weight(dy, x, mode=mode))
        print(f"N={N} K={K}: bwd_wgt  {mode:7s} {t:7.3f} ms  {fl/t*1e3:7.1f} TF/s")
