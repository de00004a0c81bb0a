import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch, torch.nn.functional as F
import svr_amd  # noqa
from svr_amd import ops
from svr_amd.model import unet as U
B = 2
torch.manual_seed(0)
def run(C0, C1, Cout, H, tag):
    s0 = torch.randn(B, H, H, C0, device="cuda")
    s1 = torch.randn(B, H, H, C1, device="cuda")
    w = torch.randn(Cout, C0 + C1, 3, 3, device="cuda") / (9 * (C0 + C1)) ** 0.5
    b = torch.randn(Cout, device="cuda")
    V = ops.conv2d_virtual(s0, s1, 2, True)
    x = F.interpolate(F.relu(torch.cat((s0, s1), 3).permute(0, 3, 1, 2).double()), scale_factor=2, mode="bilinear")
    print(tag, "V err", float((V.double() - x.permute(0, 2, 3, 1)).abs().max()))
    pl = ops.Conv2dPlanes(w, 1, True)
    y = ops.conv2d_fwd(V, None, 3, 1, 0, pl, b)
    ref = F.conv2d(x, w.double(), b.double(), padding=1).permute(0, 2, 3, 1)
    e = (y.double() - ref).abs()
    bad = (e > 1e-5 * ref.abs().max()).nonzero()
    print(tag, f"y rel {float(e.max() / ref.abs().max()):.2e} bad {bad.shape[0]} of {e.numel()} first {bad[:8].tolist()} last {bad[-3:].tolist()}", flush=True)
    refV = F.conv2d(V.permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=1).permute(0, 2, 3, 1)
    print(tag, f"y vs conv of the kernel's own V: {float((y.double() - refV).abs().max() / refV.abs().max()):.2e}")
run(128, 128, 64, 32, "dconv6 alone")
run(256, 256, 128, 16, "dconv5")
run(128, 128, 64, 32, "dconv6 after")
