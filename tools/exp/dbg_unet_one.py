import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch, torch.nn.functional as F
import svr_amd  # noqa
from svr_amd import ops
B = 2
torch.manual_seed(0)
def rel(a, b): return float((a - b).abs().max() / b.abs().max())
for (C, Cout, H) in ((256, 64, 64), (256, 64, 32), (256, 128, 64), (128, 64, 64), (256, 32, 64), (512, 64, 64)):
    x = torch.randn(B, H, H, C, device="cuda")
    w = torch.randn(Cout, C, 3, 3, device="cuda") / (9 * C) ** 0.5
    b = torch.randn(Cout, device="cuda")
    pl = ops.Conv2dPlanes(w, 1, True)
    y = ops.conv2d_fwd(x, None, 3, 1, 0, pl, b)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=1).permute(0, 2, 3, 1)
    e = (y.double() - ref).abs()
    bad = (e > 1e-5 * ref.abs().max()).nonzero()
    print(f"C={C} Cout={Cout} H={H}: rel {rel(y.double(), ref):.2e} bad {bad.shape[0]} first {bad[:6].tolist()}", flush=True)
