"""GPU edge cases of the hot path against the CPU oracle: the reference's real (non-cubic) grid size,
single points, empty point sets, points on / outside the volume boundary, batch of one."""
import numpy as np
import pytest
import torch

from oracle import ifnet_oracle as O
from tests import _golden as G

pytestmark = pytest.mark.gpu


def _pair(net_res=128):
    import svr_amd  # noqa: F401
    from svr_amd.model import IFNet
    st = O.name_seeded_state(net_res)
    m = IFNet(net_res=net_res)
    m.load_state_dict(st, strict=False)
    return m.cuda().train(), st


def test_real_dataset_grid_size_139x104x112():
    """The reference's actual grid (trainer/trainer_ifnet.py:23): odd sizes, MaxPool floor 139->69->34->17->8."""
    m, st = _pair()
    g = torch.Generator().manual_seed(5)
    x = (torch.rand(1, 1, 139, 104, 112, generator=g) < 0.03).float()
    pts = (torch.rand(1, 3000, 3, generator=g) - 0.5) * 1.02
    with torch.no_grad():
        got = m(x.cuda(), pts.cuda()).cpu()
        ref = O.ifnet_forward({k: v.clone() for k, v in st.items()}, x, pts, 128, training=True)
    assert G.rel_err(got.numpy(), ref.numpy()) < 1e-4


def test_boundary_and_outside_points():
    m, st = _pair()
    g = torch.Generator().manual_seed(6)
    x = (torch.rand(2, 1, 16, 16, 16, generator=g) < 0.2).float()
    special = torch.tensor([[-0.5, -0.5, -0.5], [0.5, 0.5, 0.5], [0.0, 0.5, -0.5], [0.75, 0.0, 0.0], [-3.0, 2.0, 0.1],
                            [0.4999999, -0.4999999, 0.0], [0.0, 0.0, 0.0]])
    pts = torch.cat([special, torch.rand(57, 3, generator=g) - 0.5]).unsqueeze(0).repeat(2, 1, 1)
    with torch.no_grad():
        got = m(x.cuda(), pts.cuda()).cpu()
        ref = O.ifnet_forward({k: v.clone() for k, v in st.items()}, x, pts, 128, training=True)
    assert G.rel_err(got.numpy(), ref.numpy()) < 1e-4


def test_single_value_batchnorm_raises_like_torch():
    """B=1 on a 16^3 grid leaves one value per channel at the last level: torch (and so the reference) raises."""
    m, _ = _pair()
    with pytest.raises(ValueError, match="Expected more than 1 value per channel"):
        m(torch.zeros(1, 1, 16, 16, 16).cuda(), torch.zeros(1, 4, 3).cuda())


@pytest.mark.parametrize("B,N,D", [(1, 1, 32), (3, 2, 16), (1, 129, 32)])
def test_tiny_point_sets_forward_backward(B, N, D):
    from svr_amd.trainer import bce_with_logits_sum_mean
    m, st = _pair()
    g = torch.Generator().manual_seed(7 + N)
    x = (torch.rand(B, 1, D, D, D, generator=g) < 0.2).float()
    pts = torch.rand(B, N, 3, generator=g) - 0.5
    occ = (torch.rand(B, N, generator=g) < 0.5).float()
    logits = m(x.cuda(), pts.cuda())
    loss = bce_with_logits_sum_mean(logits, occ.cuda())
    loss.backward()
    ref_st = O.make_leaf_state(st)
    ref = O.training_step(ref_st, {"input": x, "points": pts, "occupancies": occ})
    ref["loss"].backward()
    assert G.rel_err(logits.detach().cpu().numpy(), ref["logits"].detach().numpy()) < 1e-4
    assert abs(loss.item() - ref["loss"].item()) < 1e-5 * abs(ref["loss"].item())
    gw = m.fc_out.weight.grad.cpu().numpy().reshape(-1)
    assert G.rel_err(gw, ref_st["fc_out.weight"].grad.numpy().reshape(-1)) < 1e-4


def test_empty_point_set():
    m, _ = _pair()
    x = torch.zeros(2, 1, 16, 16, 16).cuda()
    with torch.no_grad():
        out = m(x, torch.zeros(2, 0, 3).cuda())
    assert tuple(out.shape) == (2, 0)


def test_spatial_sort_does_not_change_results():
    m, _ = _pair()
    g = torch.Generator().manual_seed(9)
    x = (torch.rand(2, 1, 32, 32, 32, generator=g) < 0.1).float().cuda()
    pts = (torch.rand(2, 5000, 3, generator=g) - 0.5).cuda()
    with torch.no_grad():
        m.eval()
        a = m(x, pts, spatial_sort=True)
        b = m(x, pts, spatial_sort=False)
    assert torch.equal(a, b)          # forward is order independent bit for bit (rows are independent)


def test_baseline_config2_gather_plus_mlp_only():
    """BASELINE configs[1] shape: six feature volumes at the 64^3 level sizes, 10k points, batch 4, the gather +
    point-MLP kernels only (no encoder), against the CPU oracle's grid_sample + conv1d chain: 1e-4 on the logits."""
    import svr_amd  # noqa: F401
    from svr_amd import ops
    B, D, N = 4, 64, 10000
    chans = [1, 16, 32, 64, 128, 128]
    g = torch.Generator().manual_seed(102)
    vols, d = [], D
    for i, c in enumerate(chans):
        vols.append(torch.randn(B, c, d, d, d, generator=g))
        if i >= 1:
            d //= 2
    pts = torch.rand(B, N, 3, generator=g) - 0.5
    st = O.name_seeded_state(128)
    ref = O.point_mlp(st, O.gather_features(vols, pts, 128))                      # (B, N)
    layout = ops.FeatureLayout(chans)
    rows = ops.gather_fwd([v.permute(0, 2, 3, 4, 1).contiguous().cuda() for v in vols], pts.cuda(), layout,
                          float(np.float32(0.0722)), False)
    perm = layout.reference_permutation()
    w0 = st["fc_0.weight"].squeeze(2)
    w0p = torch.cat([w0, w0.new_zeros(w0.shape[0], 1)], 1)[:, torch.where(perm >= 0, perm, torch.full_like(perm, w0.shape[1]))]
    h = ops.linear_fwd(rows, w0p.contiguous().cuda(), st["fc_0.bias"].cuda(), relu=True)
    h = ops.linear_fwd(h, st["fc_1.weight"].squeeze(2).contiguous().cuda(), st["fc_1.bias"].cuda(), relu=True)
    h = ops.linear_fwd(h, st["fc_2.weight"].squeeze(2).contiguous().cuda(), st["fc_2.bias"].cuda(), relu=True)
    z = ops.fc_out_fwd(h, st["fc_out.weight"].reshape(-1).contiguous().cuda(), st["fc_out.bias"].cuda()).view(B, N)
    assert G.rel_err(z.cpu().numpy(), ref.numpy()) < 1e-4


def test_f16x3_domain_is_loud_and_bf16x6_covers_it():
    """The f16-split forward is specified for |x| < 65504 (f16 range): beyond it the result must be non-finite (never a
    silently wrong number), and the bf16x6 forward (any f32 range) must still be right."""
    import svr_amd  # noqa: F401
    from svr_amd import ops
    g = torch.Generator().manual_seed(7)
    x = torch.randn(256, 64, generator=g)
    x[5, 3] = 1.0e6                                   # outside the f16 range
    w = torch.randn(32, 64, generator=g) / 8
    ref = x.double() @ w.double().t()
    y16 = ops.linear_fwd(x.cuda(), w.cuda(), None, relu=False, mode="f16x3").cpu()
    assert not torch.isfinite(y16[5]).all()           # loud: inf / nan in the affected row
    ok_rows = [r for r in range(256) if r != 5]
    assert G.rel_err(y16[ok_rows].numpy(), ref[ok_rows].numpy()) < 2e-6   # the other rows are unaffected
    y6 = ops.linear_fwd(x.cuda(), w.cuda(), None, relu=False, mode="bf16x6").cpu()
    assert G.rel_err(y6.numpy(), ref.numpy()) < 2e-6
