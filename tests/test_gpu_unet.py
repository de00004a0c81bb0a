"""GPU parity of the hand-kernel UNet (fused im2col + split-precision MFMA GEMMs + BatchNorm kernels,
conv2d.hip) against the CPU oracle restatement of the reference's model/unet.py (oracle/scene_oracle.py: stock torch CPU
ops; pinned through tests/golden/scene_cfg5small.npz, which test_gpu_scene_parity.py checks end to end)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import scene_oracle as S
from tests import _golden as G

pytestmark = pytest.mark.gpu


def _ops():
    import svr_amd  # noqa: F401
    from svr_amd import ops
    return ops


@pytest.mark.parametrize("path", ["igemm", "explicit"])
@pytest.mark.parametrize("B,H,W,C0,C1,Cout,k,stride,act,up", [
    (2, 9, 7, 8, 0, 32, 4, 2, 1, False), (1, 8, 8, 3, 0, 32, 4, 2, 0, False), (2, 5, 6, 16, 16, 32, 3, 1, 2, True),
    (1, 1, 1, 32, 0, 64, 3, 1, 2, True), (2, 4, 3, 8, 24, 1, 3, 1, 2, True), (3, 2, 2, 32, 0, 32, 4, 2, 1, False),
    # reduction splits (few tiles, long reductions), 128-wide tiles, two sources straddling a k-step, odd sizes at stride 2
    (1, 4, 4, 256, 256, 256, 3, 1, 2, True), (2, 16, 16, 64, 0, 128, 4, 2, 1, False), (2, 7, 5, 20, 12, 72, 4, 2, 1, False),
    (1, 24, 20, 32, 0, 160, 3, 1, 2, False), (2, 2, 2, 64, 0, 256, 4, 2, 1, False),
    # 1..4 output channels: the plain-FMA kernels (last decoder layer)
    (2, 6, 5, 16, 0, 3, 3, 1, 2, True), (1, 8, 6, 12, 4, 2, 4, 2, 1, False), (1, 5, 5, 3, 0, 4, 3, 1, 0, False),
    (2, 40, 36, 32, 32, 1, 3, 1, 2, True)])
def test_conv_block_forward_backward(B, H, W, C0, C1, Cout, k, stride, act, up, path):
    """act -> [x2 bilinear upsample] -> conv on cat(src0, src1): the block's output and all four gradients against stock
    torch CPU ops in float64 -- as implicit GEMMs (conv2d_igemm.hip, the default) and through the explicit patch matrix."""
    ops = _ops()
    from svr_amd.model import unet as U
    _ConvBlockFn = U._ConvBlockIgemmFn if path == "igemm" else U._ConvBlockFn
    if path == "explicit" and (k * k * (C0 + C1)) % 16 and Cout > 1:
        pytest.skip("explicit path: the patch matrix's width must be a multiple of 16")
    g = torch.Generator().manual_seed(B * 100 + H + C0)
    s0 = torch.randn(B, C0, H, W, generator=g, dtype=torch.float64).requires_grad_(True)
    s1 = torch.randn(B, C1, H, W, generator=g, dtype=torch.float64).requires_grad_(True) if C1 else None
    w = (torch.randn(Cout, C0 + C1, k, k, generator=g, dtype=torch.float64) / (k * k * (C0 + C1)) ** 0.5).requires_grad_(True)
    b = torch.randn(Cout, generator=g, dtype=torch.float64).requires_grad_(True)
    x = torch.cat((s0, s1), 1) if C1 else s0
    x = F.leaky_relu(x, 0.2) if act == 1 else (F.relu(x) if act == 2 else x)
    if up:
        x = F.interpolate(x, scale_factor=2, mode="bilinear")
    ref = F.conv2d(x, w, b, stride=stride, padding=1)
    dy = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    grads = torch.autograd.grad(ref, [t for t in (s0, s1, w, b) if t is not None], dy)
    cl = lambda t: t.detach().float().permute(0, 2, 3, 1).contiguous().cuda().requires_grad_(True)   # noqa: E731
    g0, g1 = cl(s0), (cl(s1) if C1 else None)
    wg, bg = w.detach().float().cuda().requires_grad_(True), b.detach().float().cuda().requires_grad_(True)
    y = _ConvBlockFn.apply(g0, g1, wg, bg, k, stride, act, up)
    assert G.rel_err(y.detach().cpu().permute(0, 3, 1, 2).numpy(), ref.detach().numpy()) < 3e-6
    y.backward(dy.float().permute(0, 2, 3, 1).contiguous().cuda())
    got = [g0.grad.cpu().permute(0, 3, 1, 2)] + ([g1.grad.cpu().permute(0, 3, 1, 2)] if C1 else []) + [wg.grad.cpu(), bg.grad.cpu()]
    for a, r in zip(got, grads):
        # explicit path: the library's default backward split on the patch matrix; igemm: always the scaled f16 split
        assert G.rel_err(a.numpy(), r.numpy()) < (5e-6 if path == "igemm" else 5e-5), (a.shape,)


def test_batched_weight_preparation_gives_the_same_planes():
    """ops.conv2d_prepare_many (all layers of a network in three launches) leaves the same bits as preparing layer by layer:
    amax words, forward planes, backward planes (4 parity classes at stride 2); 1-4-channel layers keep the f32 weight."""
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    shapes = [(32, 3, 4, 2, False), (64, 32, 4, 2, True), (256, 512, 3, 1, True), (72, 20, 3, 1, True), (1, 64, 3, 1, True)] + \
             [(32, 16, 3, 1, True)] * 14                     # 19 layers: two batches
    ws = [(torch.randn(co, c, k, k, generator=g) * 10.0 ** (i % 5 - 2)).cuda() for i, (co, c, k, s, b) in enumerate(shapes)]
    many = ops.conv2d_prepare_many([(w, s, b) for w, (co, c, k, s, b) in zip(ws, shapes)])
    for w, (co, c, k, s, b), pl in zip(ws, shapes, many):
        one = ops.Conv2dPlanes(w, s, want_bwd=b)
        assert pl.small == one.small and pl.has_bwd == one.has_bwd
        if one.small:
            assert torch.equal(pl.w, one.w)
            continue
        nbytes = int(ops._lib.lib().svr_conv2d_planes_bytes(co, c, k))
        fwd = 4 * co * k * k * ((c + 15) // 16 * 16)
        used = fwd if not b else (fwd + 255) // 256 * 256 + 4 * c * k * k * ((co + 15) // 16 * 16)
        assert used <= nbytes
        a = pl.buf[pl._amax - pl.buf.data_ptr():][:4]
        p = pl.buf[pl._planes - pl.buf.data_ptr():][:used]
        assert torch.equal(a, one.buf[:4]) and torch.equal(p, one.buf[256:256 + used])


@pytest.mark.parametrize("variant,B,H,W", [("full", 2, 256, 256), ("mini", 2, 48, 64)])
def test_unet_matches_oracle(variant, B, H, W):
    import svr_amd  # noqa: F401
    from svr_amd.model import UNetMini, Unet
    g = torch.Generator().manual_seed(31)
    rgb = torch.rand(B, 3, H, W, generator=g) * 2 - 1
    m = (Unet if variant == "full" else UNetMini)(channels_in=3, channels_out=1)
    st = S.name_seeded_like(m.state_dict(), 1.0, "unet.")
    m.load_state_dict(st, strict=False)
    m = m.cuda().train()
    out = m(rgb.cuda())
    assert tuple(out.shape) == (B, 1, H, W)
    ref_st = {k: v.clone().requires_grad_(not k.endswith(("running_mean", "running_var"))) for k, v in st.items()}
    ref = S.unet_forward(ref_st, rgb, variant, True)
    e = G.rel_err(out.detach().cpu().numpy(), ref.detach().numpy())
    print(variant, "unet output rel err", e)
    assert e < 2e-5
    w = torch.randn(ref.shape, generator=g)
    (ref * w).sum().backward()
    (out * w.cuda()).sum().backward()
    top = max(v.grad.norm().item() for v in ref_st.values() if v.grad is not None)
    for name, p in m.named_parameters():
        r = ref_st[name].grad.double()
        q = p.grad.detach().cpu().double()
        if r.norm().item() < 1e-3 * top:          # conv bias directly in front of BatchNorm: true gradient 0
            assert q.norm().item() < 2e-3 * top, name
            continue
        n = abs(q.norm().item() - r.norm().item()) / r.norm().item()
        med = float((q - r).abs().median() / r.abs().max())
        assert n < 5e-3 and med < 2e-3, (name, n, med)
    for name, bbuf in m.named_buffers():
        if "running" in name:
            assert G.rel_err(bbuf.cpu().numpy(), ref_st[name].detach().numpy()) < 1e-4, name
    # the stock-op backend of the same module gives the same output (A/B switch)
    m.backend = "stock"
    with torch.no_grad():
        stock = m(rgb.cuda())
    assert G.rel_err(stock.cpu().numpy(), ref.detach().numpy()) < 1e-4


def test_whole_step_hip_graph_replays_the_eager_step():
    """svr_amd.graphs.GraphedStep: the captured step (forward, backward, side-stream sorts, Adam) replayed on new inputs
    gives the eager step's loss and parameters (float atomics in the 128-channel scatter: 1e-5 / one Adam flip)."""
    import copy
    import svr_amd  # noqa: F401
    from oracle import ifnet_oracle as O
    from svr_amd.graphs import GraphedStep
    from svr_amd.trainer import ImplicitRefinementTrainer

    def make():
        tr = ImplicitRefinementTrainer()
        tr.ifnet.load_state_dict(O.name_seeded_state(128), strict=False)
        tr = tr.cuda().train()
        return tr, torch.optim.Adam(tr.ifnet.parameters(), lr=1e-4, capturable=True)

    def batch(seed):
        g = torch.Generator().manual_seed(seed)
        return {"input": (torch.rand(2, 1, 32, 32, 32, generator=g) < 0.05).float().cuda(),
                "points": (torch.rand(2, 3000, 3, generator=g) - 0.5).cuda(),
                "occupancies": (torch.rand(2, 3000, generator=g) < 0.5).float().cuda()}

    tr_e, opt_e = make()
    tr_g, opt_g = make()
    gs = GraphedStep(tr_g, opt_g, batch(0), warmup=1)           # one eager warm-up step (creates the optimizer state), then capture
    tr_g.load_state_dict(copy.deepcopy(tr_e.state_dict()))     # undo the warm-up step: both trainers start from the same state
    for st in opt_g.state.values():
        for k, v in st.items():
            if torch.is_tensor(v):
                v.zero_()
    for seed in (1, 2, 3):
        b = batch(seed)
        opt_e.zero_grad(set_to_none=True)
        le = tr_e.training_step(b, 0)["loss"]
        le.backward()
        opt_e.step()
        lg = gs.run(b)
        assert abs(le.item() - lg.item()) < 1e-4 * abs(le.item()), (seed, le.item(), lg.item())
    for (n, pe), (_, pg) in zip(tr_e.named_parameters(), tr_g.named_parameters()):
        assert float((pe - pg).abs().max()) <= 6.1e-4, n             # three Adam steps of lr 1e-4: at most one sign flip each
