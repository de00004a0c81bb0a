"""GPU tests of the bf16-STORAGE throughput mode of the query path (bf16_path.hip; north_star "bf16 occupancy logits",
BASELINE configs[1]: 64^3 grid, 10k points, batch 4, grid_sample + MLP kernels only).

What is exact and what is not:
  * the sample geometry is the f32 code of the default path -> corner indices bit-exact (tested there), and because
    all gather arithmetic is f32 in ATen's order, the bf16 feature rows equal
    bf16(F.grid_sample(float(bf16 volumes))) BIT FOR BIT;
  * logits carry bf16's 8 mantissa bits: they are reported against BOTH reference answers --
    the reference run end to end in bf16 on the CPU and in fp32 (tests/golden/ifnet_bf16_*.npz, made from the imported
    reference by oracle/gen_golden.py --only-bf16).  The reference's own bf16 run differs from its fp32 run by
    2e-2 .. 9e-2 (rel = max|a-b| / max|b|); this mode (f32 encoder, bf16 storage, f32 accumulation) must be at least
    as close to fp32 as that.  It is never held to the fp32 path's 1e-4 gate."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ifnet_oracle as O
from tests import _golden as G

pytestmark = pytest.mark.gpu
DISP = float(np.float32(0.0722))


def _ops():
    import svr_amd  # noqa: F401
    from svr_amd import ops
    return ops


def _cl(v):
    return v.permute(0, 2, 3, 4, 1).contiguous().cuda()


@pytest.mark.parametrize("B,dims,N,spread,net_res", [(2, (16, 16, 16), 777, 1.0, 128), (1, (35, 26, 28), 500, 1.3, 128),
                                                    (3, (32, 32, 32), 1, 1.0, 128), (2, (16, 12, 20), 300, 1.1, 32)])
def test_bf16_gather_rows_are_bit_exact(B, dims, N, spread, net_res):
    ops = _ops()
    chans = O.level_channels(net_res)
    g = torch.Generator().manual_seed(5 + N)
    vols, d = [], list(dims)
    for i, c in enumerate(chans):
        vols.append(torch.randn(B, c, *d, generator=g).bfloat16())
        if i >= 1:
            d = [max(1, s // 2) for s in d]
    pts = (torch.rand(B, N, 3, generator=g) - 0.5) * spread
    layout = ops.FeatureLayout(chans)
    a = O.ARCH[net_res]
    rows = ops.gather_fwd_bf16([_cl(v) for v in vols], pts.cuda(), layout, float(np.float32(a["disp"])), a["align_corners"])
    assert rows.dtype == torch.bfloat16 and tuple(rows.shape) == (B * N, layout.row_stride)
    ref = O.gather_features([v.float() for v in vols], pts, net_res).bfloat16()       # f32 math on bf16 values, one rounding
    perm = layout.reference_permutation()
    valid = perm >= 0
    got = rows.cpu().view(B, N, -1)
    out = torch.empty(B, N, int(valid.sum()), dtype=torch.bfloat16)
    out[:, :, perm[valid]] = got[:, :, valid]
    assert torch.equal(out.permute(0, 2, 1), ref)
    assert torch.all(got[:, :, ~valid] == 0)


@pytest.mark.parametrize("M,N,K", [(1000, 256, 2592), (130, 512, 64), (4099, 256, 256)])
def test_bf16_linear_and_fc_out(M, N, K):
    ops = _ops()
    g = torch.Generator().manual_seed(M + K)
    x = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16()
    b = torch.randn(N, generator=g)
    ref = F.relu(x.double() @ w.double().t() + b.double())
    y = ops.linear_fwd_bf16(x.cuda(), w.cuda(), b.cuda(), relu=True)
    assert y.dtype == torch.bfloat16
    # f32 accumulation (1e-6) + ONE rounding of the result to bf16 (half an ulp = 2^-9 relative)
    err = (y.cpu().double() - ref).abs()
    assert bool((err <= 2.0 ** -8 * ref.abs() + 1e-5 * ref.abs().max()).all()), float((err / ref.abs().max()).max())
    yl = ops.linear_fwd_bf16(x.cuda(), w.cuda(), None, relu=False)
    refl = x.double() @ w.double().t()
    assert G.rel_err(yl.cpu().double().numpy(), refl.numpy()) < 2.0 ** -8
    wo = torch.randn(N, generator=g) / 16
    bo = torch.randn(1, generator=g)
    z = ops.fc_out_fwd_bf16(y, wo.cuda(), bo.cuda())
    zref = y.cpu().double() @ wo.double() + bo.double()
    assert G.rel_err(z.cpu().numpy(), zref.numpy()) < 1e-5


def test_baseline_config2_bf16_gather_plus_mlp():
    """BASELINE configs[1] (seed 102): six bf16 feature volumes at the 64^3 level shapes, 10 000 points,
    batch 4, gather + point MLP only.  Against the CPU chain of the reference's ops (grid_sample + conv1d,
    oracle/ifnet_oracle.py) in fp32 on the same bf16 volumes / bf16-rounded weights, and against that chain run in
    bf16 on the CPU; both deviations are printed."""
    import svr_amd  # noqa: F401
    from svr_amd import ops
    B, D, N = 4, 64, 10000
    chans = [1, 16, 32, 64, 128, 128]
    g = torch.Generator().manual_seed(102)
    vols, d = [], D
    for i, c in enumerate(chans):
        vols.append(torch.randn(B, c, d, d, d, generator=g).bfloat16())
        if i >= 1:
            d //= 2
    pts = torch.rand(B, N, 3, generator=g) - 0.5
    st = O.name_seeded_state(128)
    st_r = {k: (v.bfloat16().float() if k.endswith("weight") and k.startswith("fc_") and "fc_out" not in k else v) for k, v in st.items()}
    ref32 = O.point_mlp(st_r, O.gather_features([v.float() for v in vols], pts, 128))          # fp32 chain, bf16-valued operands
    st_b = {k: v.bfloat16() for k, v in st.items()}
    refbf = O.point_mlp(st_b, O.gather_features(vols, pts.bfloat16(), 128)).float()            # everything in bf16 on the CPU
    layout = ops.FeatureLayout(chans)
    rows = ops.gather_fwd_bf16([_cl(v) for v in vols], pts.cuda(), layout, DISP, False)
    perm = layout.reference_permutation()
    w0 = st["fc_0.weight"].squeeze(2)
    w0p = torch.cat([w0, w0.new_zeros(w0.shape[0], 1)], 1)[:, torch.where(perm >= 0, perm, torch.full_like(perm, w0.shape[1]))]
    h = ops.linear_fwd_bf16(rows, w0p.contiguous().bfloat16().cuda(), st["fc_0.bias"].cuda(), relu=True)
    h = ops.linear_fwd_bf16(h, st["fc_1.weight"].squeeze(2).contiguous().bfloat16().cuda(), st["fc_1.bias"].cuda(), relu=True)
    h = ops.linear_fwd_bf16(h, st["fc_2.weight"].squeeze(2).contiguous().bfloat16().cuda(), st["fc_2.bias"].cuda(), relu=True)
    z = ops.fc_out_fwd_bf16(h, st["fc_out.weight"].reshape(-1).contiguous().cuda(), st["fc_out.bias"].cuda()).view(B, N).cpu()
    e32, ebf = G.rel_err(z.numpy(), ref32.numpy()), G.rel_err(z.numpy(), refbf.numpy())
    print(f"config 2, bf16 storage: vs fp32 chain {e32:.3e}, vs bf16-on-CPU chain {ebf:.3e}; "
          f"bf16-on-CPU vs fp32 chain {G.rel_err(refbf.numpy(), ref32.numpy()):.3e}")
    # only the activations are rounded to bf16 between layers here (3 roundings of 2^-9, 256..2592-term dot products)
    assert e32 < 1e-2
    assert e32 <= G.rel_err(refbf.numpy(), ref32.numpy())      # at least as close to fp32 as the all-bf16 CPU chain


@pytest.mark.parametrize("case", ["bf16_cfg1", "bf16_b2"])
@pytest.mark.parametrize("mode", ["train", "eval"])
def test_bf16_query_against_both_reference_answers(case, mode):
    """IFNet.encode(x, storage='bf16') + query(): logits vs the reference's fp32 run and vs the reference's bf16 run."""
    import svr_amd  # noqa: F401
    from svr_amd.model import IFNet
    z = G.load("ifnet_" + case)
    net_res, seed, B, d0, d1, d2, N = (int(v) for v in z["meta"])
    x = torch.from_numpy(np.unpackbits(z["x_bits"])[:B * d0 * d1 * d2].astype(np.float32)).view(B, 1, d0, d1, d2)
    pts = torch.from_numpy(z["points"].copy())
    m = IFNet(net_res=net_res)
    m.load_state_dict(G.state(net_res, z=z), strict=False)
    m = m.cuda().train(mode == "train")
    f32 = m.query(m.encode(x.cuda()), pts.cuda()).cpu().numpy()
    bf = m.query(m.encode(x.cuda(), storage="bf16"), pts.cuda()).cpu().numpy()
    ref32, refbf = z[f"logits_f32_{mode}"], z[f"logits_bf16_{mode}"]
    assert G.rel_err(f32, ref32) < 1e-4                                   # the default path, for orientation
    e32, ebf, ref_gap = G.rel_err(bf, ref32), G.rel_err(bf, refbf), G.rel_err(refbf, ref32)
    print(f"{case} {mode}: bf16-storage mode vs reference fp32 {e32:.3e}, vs reference bf16 {ebf:.3e}; "
          f"reference bf16 vs reference fp32 {ref_gap:.3e}")
    assert e32 < 2e-2 and e32 <= ref_gap          # closer to the fp32 answer than the reference's own bf16 run
    assert ebf < 1.5 * ref_gap                    # and within the reference's bf16 noise of its bf16 answer


def test_bf16_dense_grid_inference():
    """evaluate_network_on_grid(storage='bf16') (SURVEY 8 f1 in the throughput mode) vs the fp32 cached-pyramid path."""
    import svr_amd  # noqa: F401
    from svr_amd.model import IFNet, evaluate_network_on_grid
    z = G.load("ifnet_b3")
    net_res, x, pts, _ = G.ifnet_inputs(z)
    m = IFNet(net_res=net_res)
    m.load_state_dict(G.state(net_res, z=z), strict=False)
    m = m.cuda().eval()
    a = evaluate_network_on_grid(m, x[:1].cuda(), (16, 16, 16), 1, points_batch_size=1000)
    b = evaluate_network_on_grid(m, x[:1].cuda(), (16, 16, 16), 1, points_batch_size=1000, storage="bf16")
    assert b.shape == a.shape and np.abs(a - b).max() < 2e-2 * max(np.abs(a).max(), 1e-6) + 5e-3


def test_query_spatial_sort_is_result_neutral():
    import svr_amd  # noqa: F401
    from svr_amd.model import IFNet
    z = G.load("ifnet_cfg1")
    net_res, x, pts, _ = G.ifnet_inputs(z)
    m = IFNet(net_res=net_res)
    m.load_state_dict(G.state(net_res, z=z), strict=False)
    m = m.cuda().eval()
    for storage in ("f32", "bf16"):
        lv = m.encode(x.cuda(), storage)
        assert torch.equal(m.query(lv, pts.cuda()), m.query(lv, pts.cuda(), spatial_sort=True)), storage


@pytest.mark.parametrize("B,dims,N,spread", [(2, (16, 16, 16), 777, 1.0), (1, (35, 26, 28), 500, 1.3), (3, (32, 32, 32), 1, 1.0),
                                             (2, (24, 24, 24), 1280, 1.05)])
def test_fused_bf16_gather_fc0_equals_the_two_bf16_kernels(B, dims, N, spread):
    """gather_fc0.hip's bf16-storage variant (svr_gather_fc0_bf16_*) against the two kernels it fuses (bf16 gather +
    svr_linear_fwd_bf16): the SAME bf16 feature values (the gather's arithmetic and single rounding) and the same bf16
    weights meet in f32 accumulators, only the summation order over K differs -> h0 agrees to one bf16 ulp of the row
    scale (1 / 128), and with the float64 product of the bit-exact bf16 rows to the same bound."""
    ops = _ops()
    chans = O.level_channels(128)
    g = torch.Generator().manual_seed(11 + N)
    vols, d = [], list(dims)
    for i, c in enumerate(chans):
        vols.append(torch.randn(B, c, *d, generator=g).bfloat16())
        if i >= 1:
            d = [max(1, s // 2) for s in d]
    pts = ((torch.rand(B, N, 3, generator=g) - 0.5) * spread).cuda()
    layout = ops.FeatureLayout(chans)
    vols_g = [_cl(v) for v in vols]
    w = (torch.randn(256, layout.row_stride, generator=g) / 30).cuda()
    w[:, layout.width:] = 0
    bias = torch.randn(256, generator=g).cuda()
    assert ops.gather_fc0_bf16_supported(vols_g, layout, DISP, False, 256)
    rows = ops.gather_fwd_bf16(vols_g, pts, layout, DISP, False)
    want = ops.linear_fwd_bf16(rows, ops.cast_bf16(w), bias, relu=True).float()
    prep = ops.gather_fc0_bf16_prepare(vols_g, layout, DISP, False, w)
    h0 = ops.gather_fc0_bf16_run(prep, pts, bias, relu=True)
    assert h0.dtype == torch.bfloat16 and tuple(h0.shape) == (B * N, 256)
    scale = float(want.abs().max())
    assert float((h0.float() - want).abs().max()) <= scale / 128
    ref = torch.relu(rows.double().cpu() @ ops.cast_bf16(w).double().cpu().t() + bias.double().cpu())
    assert float((h0.double().cpu() - ref).abs().max()) <= scale / 128
    # most entries are identical bits (the order of an f32 sum rarely moves a bf16 rounding)
    assert float((h0.float() == want).float().mean()) > 0.97
    # no ReLU / another point set on the same preparation
    h1 = ops.gather_fc0_bf16_run(prep, pts[:, : max(1, N // 2)].contiguous(), bias, relu=False)
    assert tuple(h1.shape) == (B * max(1, N // 2), 256)
    assert torch.equal(torch.relu(h1.float()).view(B, -1, 256), h0.float().view(B, N, 256)[:, : max(1, N // 2)])


def test_bf16_query_uses_the_fused_kernel_and_matches_the_unfused_mode():
    """IFNet.query on a bf16 pyramid: the fused path (default) against SVR_NO_FUSED_FC0_BF16's two-kernel path on the same
    pyramid -- logits within bf16 noise of each other (3e-3 of the logit scale), and the prepared form gives the fused bits."""
    import svr_amd  # noqa: F401
    from svr_amd.model import IFNet
    from svr_amd.model import ifnet as ifn
    m = IFNet(net_res=128)
    m.load_state_dict(O.name_seeded_state(128), strict=False)
    m = m.cuda().eval()
    g = torch.Generator().manual_seed(91)
    x = (torch.rand(2, 1, 32, 32, 32, generator=g) < 0.05).float().cuda()
    pts = (torch.rand(2, 3000, 3, generator=g) - 0.5).cuda()
    levels = m.encode(x, storage="bf16")
    assert ifn.FUSE_FC0_BF16
    z_fused = m.query(levels, pts)
    prep = m.prepare_query(levels, 3000)
    assert prep is not None and torch.equal(m.query(levels, pts, prepared=prep), z_fused)
    saved, ifn.FUSE_FC0_BF16 = ifn.FUSE_FC0_BF16, False
    try:
        z_two = m.query(levels, pts)
    finally:
        ifn.FUSE_FC0_BF16 = saved
    assert float((z_fused - z_two).abs().max()) <= 3e-3 * float(z_two.abs().max())
