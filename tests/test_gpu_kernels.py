"""GPU parity of the individual HIP kernels (through the C ABI) against the CPU oracle /
stock torch CPU ops on the same seeded inputs.  Integer work is bit-exact, floating point
within the tolerance written in each test."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ifnet_oracle as O
from tests import _golden as G

pytestmark = pytest.mark.gpu


def _ops():
    import svr_amd  # noqa: F401
    from svr_amd import ops
    return ops


def _cl(v):  # NCDHW -> channels-last (B,D,H,W,C) on the GPU
    return v.permute(0, 2, 3, 4, 1).contiguous().cuda()


def _ncdhw(v):  # channels-last GPU -> NCDHW CPU
    return v.cpu().permute(0, 4, 1, 2, 3).contiguous()


def _rand_levels(B, dims, chans, seed):
    g = torch.Generator().manual_seed(seed)
    vols = []
    d = list(dims)
    for i, c in enumerate(chans):
        vols.append(torch.randn(B, c, *d, generator=g))
        if i >= 1:
            d = [max(1, s // 2) for s in d]
    return vols


@pytest.mark.parametrize("wide", [False, True], ids=["offsets32", "offsets64"])
@pytest.mark.parametrize("B,dims,N,spread", [(2, (16, 16, 16), 777, 1.0), (1, (35, 26, 28), 500, 1.3), (3, (32, 32, 32), 1, 1.0)])
def test_gather_fwd_indices_and_features(B, dims, N, spread, wide):
    """`wide` forces the 64-bit-offset forward body (gather.hip gather_fwd_body<16..128>) that production only takes
    when a volume or the feature matrix has >= 2^31 elements (B >= 17 at the config-3 shape)."""
    ops = _ops()
    from svr_amd import _lib
    flags = _lib.GATHER_WIDE_OFFSETS if wide else 0
    chans = [1, 16, 32, 64, 128, 128]
    vols = _rand_levels(B, dims, chans, 5)
    g = torch.Generator().manual_seed(6)
    pts = (torch.rand(B, N, 3, generator=g) - 0.5) * spread
    layout = ops.FeatureLayout(chans)
    vols_g = [_cl(v) for v in vols]
    rows = ops.gather_fwd(vols_g, pts.cuda(), layout, float(np.float32(0.0722)), False, flags=flags)
    # bit-exact corner indices on every level
    for l, v in enumerate(vols):
        idx = ops.corner_indices(vols_g, pts.cuda(), layout, l, float(np.float32(0.0722)), False).cpu()
        ref, _ = O.corner_indices(pts, v.shape[2:], 128)
        assert torch.equal(idx, ref), f"level {l}"
    # features vs F.grid_sample (reference layout): bit for bit
    ref = O.gather_features(vols, pts, 128)                           # (B, sumC*7, N)
    perm = layout.reference_permutation()
    got = rows.cpu().view(B, N, -1)
    valid = perm >= 0
    got_ref_order = torch.empty(B, N, int(valid.sum()))
    got_ref_order[:, :, perm[valid]] = got[:, :, valid]
    got_ref_order = got_ref_order.permute(0, 2, 1)
    assert torch.equal(got_ref_order, ref)
    assert torch.all(got[:, :, ~valid] == 0)


@pytest.mark.parametrize("wide", [False, True], ids=["offsets32", "offsets64"])
def test_gather_fwd_align_corners_variant(wide):
    ops = _ops()
    from svr_amd import _lib
    chans = [1, 64, 128, 128]
    B, dims, N = 2, (16, 12, 20), 300
    vols = _rand_levels(B, dims, chans, 9)
    g = torch.Generator().manual_seed(10)
    pts = (torch.rand(B, N, 3, generator=g) - 0.5) * 1.1
    layout = ops.FeatureLayout(chans)
    rows = ops.gather_fwd([_cl(v) for v in vols], pts.cuda(), layout, float(np.float32(0.035)), True,
                          flags=_lib.GATHER_WIDE_OFFSETS if wide else 0)
    ref = O.gather_features(vols, pts, 32)
    perm = layout.reference_permutation()
    valid = perm >= 0
    got = rows.cpu().view(B, N, -1)
    out = torch.empty(B, N, int(valid.sum()))
    out[:, :, perm[valid]] = got[:, :, valid]
    assert torch.equal(out.permute(0, 2, 1), ref)


def test_gather_bwd_volume_and_point_grads():
    ops = _ops()
    chans = [1, 16, 32, 64, 128, 128]
    B, dims, N = 2, (16, 16, 16), 400
    vols = [v.requires_grad_(True) for v in _rand_levels(B, dims, chans, 11)]
    g = torch.Generator().manual_seed(12)
    pts = ((torch.rand(B, N, 3, generator=g) - 0.5) * 1.1).requires_grad_(True)
    ref = O.gather_features(vols, pts, 128)
    w = torch.randn(ref.shape, generator=g)
    (ref * w).sum().backward()
    layout = ops.FeatureLayout(chans)
    perm = layout.reference_permutation()
    gfeat = torch.zeros(B, N, layout.row_stride)
    valid = perm >= 0
    gfeat[:, :, valid] = w.permute(0, 2, 1)[:, :, perm[valid]]
    vols_g = [_cl(v.detach()) for v in vols]
    gvols = [torch.zeros_like(v) for v in vols_g]
    gp = ops.gather_bwd(vols_g, gvols, pts.detach().cuda(), gfeat.view(B * N, -1).cuda(), layout,
                        float(np.float32(0.0722)), False, want_gpoints=True)
    for l, v in enumerate(vols):
        assert G.rel_err(_ncdhw(gvols[l]).numpy(), v.grad.numpy()) < 1e-5, f"level {l}"
    assert G.rel_err(gp.cpu().numpy(), pts.grad.numpy()) < 1e-4


def _gfeat_from_reference_layout(w, layout, B, N):
    """(B, sumC*7, N) cotangent in the reference's row order -> (B*N, FS) rows in the internal column layout."""
    perm = layout.reference_permutation()
    valid = perm >= 0
    gfeat = torch.zeros(B, N, layout.row_stride)
    gfeat[:, :, valid] = w.permute(0, 2, 1)[:, :, perm[valid]]
    return gfeat.view(B * N, -1)


@pytest.mark.parametrize("net_res,chans,dims", [(128, [1, 16, 32, 64, 128, 128], (12, 10, 14)), (32, [1, 64, 128, 128], (9, 8, 8))])
def test_gather_bwd_deterministic_mode_is_bit_exact_and_reproducible(net_res, chans, dims):
    """SVR_GATHER_DETERMINISTIC: atomic-free scatter in the summation order of ATen's CPU grid_sampler_3d_backward
    (j, then n, then the 8 corners) -- the gradient volumes equal the reference's CPU autograd BIT FOR BIT and two
    runs give identical bytes (the production scatter uses float atomics: order dependent in the last bits)."""
    ops = _ops()
    from svr_amd import _lib
    B, N = 2, 300
    vols = [v.requires_grad_(True) for v in _rand_levels(B, dims, chans, 31)]
    g = torch.Generator().manual_seed(32)
    pts = (torch.rand(B, N, 3, generator=g) - 0.5) * 1.15
    ref = O.gather_features(vols, pts, net_res)
    w = torch.randn(ref.shape, generator=g)
    (ref * w).sum().backward()
    layout = ops.FeatureLayout(chans)
    gfeat = _gfeat_from_reference_layout(w, layout, B, N).cuda()
    vols_g = [_cl(v.detach()) for v in vols]
    a = O.ARCH[net_res]
    runs = []
    for _ in range(2):
        gvols = [torch.zeros_like(v) for v in vols_g]
        ops.gather_bwd(vols_g, gvols, pts.cuda(), gfeat, layout, float(np.float32(a["disp"])), a["align_corners"],
                       flags=_lib.GATHER_DETERMINISTIC)
        runs.append(gvols)
    for l, v in enumerate(vols):
        assert torch.equal(runs[0][l], runs[1][l]), f"level {l}: not reproducible"
        assert torch.equal(_ncdhw(runs[0][l]), v.grad), f"level {l}: not the CPU summation order"


@pytest.mark.parametrize("dims,align,N", [((16, 16, 16), False, 6000), ((9, 7, 11), False, 3000), ((8, 8, 8), True, 2500),
                                         ((5, 6, 4), True, 700)])
def test_voxel_order_is_a_permutation_and_scatter_under_it(dims, align, N):
    """svr_points_voxel_order + the per-level `level_orders` of the production scatter (row-major voxel order, two open
    register runs, face hand-over): the order is a per-sample permutation sorted by base voxel, and with DENSE points
    (N >= voxels, so runs and hand-overs dominate) the scatter under it equals the natural-order scatter and CPU
    autograd to 1e-5."""
    ops = _ops()
    B = 2
    chans = [1, 16, 32, 64, 128, 128]
    net_res = 32 if align else 128
    g = torch.Generator().manual_seed(41 + N)
    pts = (torch.rand(B, N, 3, generator=g) - 0.5) * 1.1
    lv_dims = [dims, dims] + [tuple(max(1, s >> k) for s in dims) for k in range(1, 5)]
    orders = [None]
    for l in range(1, 6):
        o = ops.voxel_order(pts.cuda(), lv_dims[l], align)
        oc = o.cpu().long()
        assert sorted(oc.tolist()) == list(range(B * N)), f"level {l}: not a permutation"
        assert torch.equal(oc // N, torch.arange(B).repeat_interleave(N)), f"level {l}: samples not contiguous"
        # sorted by the row-major base voxel of the undisplaced sample (j = 0) inside every sample
        idx, _ = O.corner_indices(pts, lv_dims[l], net_res)          # (B, 7, N, 3) z0,y0,x0
        base = idx[:, 0].reshape(B * N, 3)[oc].long()
        Dd, H, W = lv_dims[l]
        inside = ((base >= -1).all(1)) & (base[:, 0] < Dd) & (base[:, 1] < H) & (base[:, 2] < W)
        key = ((base[:, 0] + 1) * (H + 1) + base[:, 1] + 1) * (W + 1) + base[:, 2] + 1
        for b in range(B):
            kb = key[b * N:(b + 1) * N][inside[b * N:(b + 1) * N]]
            assert bool((kb[1:] >= kb[:-1]).all()), f"level {l} sample {b}: keys not sorted"
        orders.append(o)
    vols = []
    for l, c in enumerate(chans):
        vols.append(torch.randn(B, c, *lv_dims[l], generator=g).requires_grad_(True))
    layout = ops.FeatureLayout(chans)
    disp = float(np.float32(O.ARCH[net_res]["disp"]))
    g_ = O.sample_grid(pts, net_res)
    feats = [F.grid_sample(v, g_, mode="bilinear", padding_mode="zeros", align_corners=align) for v in vols]
    f = torch.cat(feats, 1)
    ref = f.reshape(B, f.shape[1] * 7, N)
    w = torch.randn(ref.shape, generator=g)
    (ref * w).sum().backward()
    gfeat = _gfeat_from_reference_layout(w, layout, B, N).cuda()
    vols_g = [_cl(v.detach()) for v in vols]
    g_nat = [torch.zeros_like(v) for v in vols_g]
    g_ord = [torch.zeros_like(v) for v in vols_g]
    ops.gather_bwd(vols_g, g_nat, pts.cuda(), gfeat, layout, disp, align)
    ops.gather_bwd(vols_g, g_ord, pts.cuda(), gfeat, layout, disp, align, level_orders=orders)
    for l, v in enumerate(vols):
        assert G.rel_err(_ncdhw(g_ord[l]).numpy(), v.grad.numpy()) < 1e-5, f"level {l} (ordered) vs autograd"
        assert G.rel_err(g_ord[l].cpu().numpy(), g_nat[l].cpu().numpy()) < 1e-5, f"level {l} ordered vs natural"
    # the joint ITEM order (svr_gather_item_order: all 7N items of a sample sorted by the base cell of their displaced
    # sample; the atomic scatter walks items with its runs kept open over kItemReps repetitions)
    iorders = [None]
    for l in range(1, 6):
        it = ops.item_order(pts.cuda(), lv_dims[l], disp, align)
        ic = it.cpu().long()
        assert sorted(ic.tolist()) == list(range(7 * B * N)), f"level {l}: item order is not a permutation"
        idx, _ = O.corner_indices(pts, lv_dims[l], net_res)          # (B, 7, N, 3)
        pn, j = ic // 7, ic % 7
        base = idx[pn // N, j, pn % N].long()
        Dd, H, W = lv_dims[l]
        inside = ((base >= -1).all(1)) & (base[:, 0] < Dd) & (base[:, 1] < H) & (base[:, 2] < W)
        key = (((pn // N) * (Dd + 1) + base[:, 0] + 1) * (H + 1) + base[:, 1] + 1) * (W + 1) + base[:, 2] + 1
        n_in = int(inside.sum())
        assert bool(inside[:n_in].all()) and bool((key[1:n_in] >= key[:n_in - 1]).all()), f"level {l}: items not sorted"
        iorders.append(it)
    g_it = [torch.zeros_like(v) for v in vols_g]
    ops.gather_bwd(vols_g, g_it, pts.cuda(), gfeat, layout, disp, align, level_orders=iorders)
    for l, v in enumerate(vols):
        assert G.rel_err(_ncdhw(g_it[l]).numpy(), v.grad.numpy()) < 1e-5, f"level {l} (item order) vs autograd"


@pytest.mark.parametrize("dims,align,N,B", [((16, 16, 16), False, 6000, 2), ((9, 7, 11), False, 3000, 3), ((8, 8, 8), True, 2500, 1),
                                           ((5, 6, 4), True, 700, 2), ((12, 10, 14), False, 1, 2), ((33, 17, 20), False, 40, 1)])
def test_pull_form_scatter_matches_autograd_and_is_reproducible(dims, align, N, B):
    """svr_gather_pull_plan + the pull-form backward scatter (atomic-free: every voxel sums the items of its <= 8 base
    cells in sorted order and is stored once): gradient volumes vs CPU autograd of grid_sample 1e-5 (dense and sparse
    point sets, odd sizes, points outside the volume, both align_corners variants), every voxel written (volumes start
    as NaN), two runs bit-identical, and the plan's keys sorted / CSR offsets exact."""
    ops = _ops()
    chans = [1, 16, 32, 64, 128, 128]
    net_res = 32 if align else 128
    g = torch.Generator().manual_seed(51 + N)
    pts = (torch.rand(B, N, 3, generator=g) - 0.5) * 1.25          # some outside [-0.5, 0.5]
    lv_dims = [dims, dims] + [tuple(max(1, s >> k) for s in dims) for k in range(1, 5)]
    vols = [torch.randn(B, c, *lv_dims[l], generator=g).requires_grad_(True) for l, c in enumerate(chans)]
    layout = ops.FeatureLayout(chans)
    disp = float(np.float32(O.ARCH[net_res]["disp"]))
    g_ = O.sample_grid(pts, net_res)
    f = torch.cat([F.grid_sample(v, g_, mode="bilinear", padding_mode="zeros", align_corners=align) for v in vols], 1)
    ref = f.reshape(B, f.shape[1] * 7, N)
    w = torch.randn(ref.shape, generator=g)
    (ref * w).sum().backward()
    gfeat = _gfeat_from_reference_layout(w, layout, B, N).cuda()
    vols_g = [_cl(v.detach()) for v in vols]
    plans = [None] * 6
    for l in (1, 2, 3):
        assert ops.pull_plan_supported(B, N, lv_dims[l], chans[l], layout.row_stride)
        plans[l] = ops.pull_plan(pts.cuda(), lv_dims[l], chans[l], layout.col[l], layout.row_stride, disp, align)
        keys = plans[l].keys.cpu().long() & 0xFFFFFFFF
        assert bool((keys[1:] >= keys[:-1]).all()), f"level {l}: keys not sorted"
        cells = B * (lv_dims[l][0] + 1) * (lv_dims[l][1] + 1) * (lv_dims[l][2] + 1)
        heads = plans[l].heads.cpu().long()             # CSR offsets: heads[c] = number of items with a key below c
        assert torch.equal(heads, torch.searchsorted(keys, torch.arange(cells + 1)))
    runs = []
    for _ in range(2):
        gv = [torch.full_like(v, float("nan")) if plans[l] is not None else torch.zeros_like(v) for l, v in enumerate(vols_g)]
        ops.gather_bwd(vols_g, gv, pts.cuda(), gfeat, layout, disp, align, level_plans=plans)
        runs.append(gv)
    for l, v in enumerate(vols):
        # levels 4 / 5 go through the float-atomic scatter: at 16^3 the coarsest level is ONE voxel that sums all 42 000
        # items in a run-dependent order (seen: 1.1e-5 against the CPU's serial order), hence the wider gate there
        tol = 1e-5 if plans[l] is not None else 4e-5
        assert G.rel_err(_ncdhw(runs[0][l]).numpy(), v.grad.numpy()) < tol, f"level {l}"
    for l in (1, 2, 3):
        assert torch.equal(runs[0][l], runs[1][l]), f"level {l}: pull scatter not reproducible"


@pytest.mark.parametrize("M,N,K", [(1000, 256, 2592), (130, 512, 64), (4099, 256, 256)])
def test_linear_fwd_bwd(M, N, K):
    ops = _ops()
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    y_ref = F.relu(x.double() @ w.double().t() + b.double())
    errs = {}
    for mode in ("f32", "bf16x6", "f16x3"):  # all forward paths are held to the same f32-level gate
        y = ops.linear_fwd(x.cuda(), w.cuda(), b.cuda(), relu=True, mode=mode)
        assert G.rel_err(y.cpu().numpy(), y_ref.numpy()) < 2e-6, mode
        yl = ops.linear_fwd(x.cuda(), w.cuda(), b.cuda(), relu=False, mode=mode)
        errs[mode] = G.rel_err(yl.cpu().numpy(), (x.double() @ w.double().t() + b.double()).numpy())
        assert errs[mode] < 2e-6, mode
    assert errs["f16x3"] < 3 * errs["f32"] + 1e-7, errs     # the f16 split is as accurate as the exact-f32 MFMA kernel
    # f16x3 over a wide dynamic range (rows from ~1e-8 to ~6e4, weights scaled far from 1: the power-of-two
    # normalisation of W is exact).  Contract (gemm_f16x3.hip): f32-level relative accuracy for |x| >= 6e-5 (the f16
    # normal range), an absolute error floor of 1.5e-11 |w| per term below that.
    for xs, wsc in ((1e-4, 300.0), (2000.0, 1e-5), (1.0, 1.0)):
        x2 = (x * xs * torch.exp(3 * torch.randn(M, 1, generator=g))).clamp(-6e4, 6e4)
        w2 = w * wsc
        ref2 = x2.double() @ w2.double().t()
        y2 = ops.linear_fwd(x2.cuda(), w2.cuda(), None, relu=False, mode="f16x3")
        err = (y2.cpu().double() - ref2).abs().amax(1)
        bound = 2e-6 * ref2.abs().amax(1) + 1.5e-11 * w2.abs().max().item() * K ** 0.5
        assert bool((err <= bound).all()), (xs, wsc, (err / bound).max().item())
    dy = torch.randn(M, N, generator=g)
    dx_ref = (dy.double() @ w.double()) * (x.double() > 0)
    dw_ref = dy.double().t() @ x.double()
    # exact-f32 MFMA: 2e-6;  bf16x3 split (backward default): 2^-16 per product -> 5e-5
    for mode, tol in (("f32", 2e-6), ("bf16x3", 5e-5)):
        dx = ops.linear_bwd_data(dy.cuda(), w.cuda(), mask=x.cuda(), mode=mode)
        assert G.rel_err(dx.cpu().numpy(), dx_ref.numpy()) < tol, mode
        dxn = ops.linear_bwd_data(dy.cuda(), w.cuda(), mode=mode)
        assert G.rel_err(dxn.cpu().numpy(), (dy.double() @ w.double()).numpy()) < tol, mode
        dw, db = ops.linear_bwd_weight(dy.cuda(), x.cuda(), mode=mode)
        assert G.rel_err(dw.cpu().numpy(), dw_ref.numpy()) < tol, mode
        assert G.rel_err(db.cpu().numpy(), dy.double().sum(0).numpy()) < 2e-6


@pytest.mark.parametrize("M,N,K,gscale", [(4099, 256, 256, 1e-6), (3000, 256, 800, 3e-3), (1000, 64, 2592, 1.0), (520, 256, 36, 1e-9)])
def test_linear_backward_at_f32_level_f16x3s(M, N, K, gscale):
    """The scaled f16 split of the two backward GEMMs ("f16x3s": svr_linear_bwd_data_f16x3 / svr_linear_bwd_weight_f16x3 /
    svr_amax_f32) against f64, on GRADIENT-like operands: dY of magnitude `gscale` with a heavy tail (rows spread over
    e^(+-3 sigma)), i.e. far outside f16's range without the power-of-two scale.  Held to the exact-f32 MFMA kernel's gate
    (2e-6) and to <= 3x that kernel's own error; bf16x3 on the same operands is an order of magnitude away.  The |max| that
    the data-gradient kernel leaves for the next layer equals the tensor's."""
    ops = _ops()
    g = torch.Generator().manual_seed(M + K)
    x = F.relu(torch.randn(M, K, generator=g)) * torch.exp(torch.randn(M, 1, generator=g))       # activations, O(1), many zeros
    w = torch.randn(N, K, generator=g) / K ** 0.5
    dy = torch.randn(M, N, generator=g) * gscale * torch.exp(3 * torch.randn(M, 1, generator=g))
    dyc, wc, xc = dy.cuda(), w.cuda(), x.cuda()
    a = ops.amax_of(dyc)
    assert a.dtype == torch.int32 and float(a.view(torch.float32)) == float(dy.abs().max())
    dx_ref = (dy.double() @ w.double()) * (x.double() > 0)
    dw_ref = dy.double().t() @ x.double()
    err = {}
    for mode in ("f32", "f16x3s", "bf16x3"):
        dx = ops.linear_bwd_data(dyc, wc, mask=xc, mode=mode)
        dw, db = ops.linear_bwd_weight(dyc, xc, mode=mode)
        err[mode] = (G.rel_err(dx.cpu().numpy(), dx_ref.numpy()), G.rel_err(dw.cpu().numpy(), dw_ref.numpy()))
        assert G.rel_err(db.cpu().numpy(), dy.double().sum(0).numpy()) < 2e-6
        if mode == "f16x3s":
            assert float(dx._svr_amax.view(torch.float32)) == float(dx.abs().max()), "fused |max| of dX"
            dxn = ops.linear_bwd_data(dyc, wc, mode=mode)                # no mask
            assert G.rel_err(dxn.cpu().numpy(), (dy.double() @ w.double()).numpy()) < 2e-6
    for i, what in enumerate(("dX", "dW")):
        assert err["f32"][i] < 2e-6 and err["f16x3s"][i] < 2e-6, (what, err)
        assert err["f16x3s"][i] < 3 * err["f32"][i] + 1e-7, (what, err)
        if N >= 32 and K % 4 == 0:
            assert err["bf16x3"][i] > 4 * err["f16x3s"][i], (what, err)   # what the mode is for
    # row-strided operands (the kept-column slices of fc_0's backward) and a prepared workspace give the same bits
    if K >= 64:
        big = torch.zeros(M, K + 32, device="cuda")
        big[:, 16:16 + K] = xc
        dw2, _ = ops.linear_bwd_weight(dyc, big[:, 16:16 + K], mode="f16x3s")
        dw1, _ = ops.linear_bwd_weight(dyc, xc, mode="f16x3s")
        assert torch.equal(dw1, dw2)


def test_f16x3s_edge_cases_zero_huge_tiny_and_non_finite_gradients():
    """Scaled f16 split at the edges of its scale: an all-zero gradient (|max| = 0: scale 1, exact zeros out), gradients of
    magnitude 1e30 and 1e-30 (the power-of-two scale is clamped to 2^+-100; f32-level results either way), a single outlier
    1e6 times the rest (the small elements keep >= 11 bits: their products are far below the outlier's), and a NaN / inf
    gradient (propagates as non-finite into the rows / columns it touches and nowhere else -- never a silently wrong finite
    number)."""
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    M, N, K = 1500, 256, 256
    x = F.relu(torch.randn(M, K, generator=g)).cuda()
    w = (torch.randn(N, K, generator=g) / 16).cuda()
    dy0 = torch.randn(M, N, generator=g)
    z = torch.zeros(M, N).cuda()
    assert float(ops.amax_of(z).view(torch.float32)) == 0.0
    assert not ops.linear_bwd_data(z, w, mode="f16x3s").any() and not ops.linear_bwd_weight(z, x, mode="f16x3s")[0].any()
    for scale in (1e30, 1e-30):
        dy = (dy0 * scale).cuda()
        dx = ops.linear_bwd_data(dy, w, mode="f16x3s")
        dw, _ = ops.linear_bwd_weight(dy, x, mode="f16x3s")
        assert G.rel_err(dx.cpu().double().numpy() / scale, (dy0.double() @ w.cpu().double()).numpy()) < 2e-6, scale
        assert G.rel_err(dw.cpu().double().numpy() / scale, (dy0.double().t() @ x.cpu().double()).numpy()) < 2e-6, scale
    dy = dy0.clone()
    dy[7, 3] = 1e6                                              # one outlier sets the scale
    dx = ops.linear_bwd_data(dy.cuda(), w, mode="f16x3s").cpu().double()
    ref = dy.double() @ w.cpu().double()
    rows = torch.arange(M) != 7
    assert G.rel_err(dx[7].numpy(), ref[7].numpy()) < 2e-6
    # rows without the outlier: relative to THEIR magnitude the error is bounded by the split's 2^-11 on elements 2^-20 below |max|
    assert G.rel_err(dx[rows].numpy(), ref[rows].numpy()) < 2e-3
    # a gradient tensor that outlives its amax word (the pool recycles words after ~250 more were handed out): recomputed
    t = (dy0 * 3.0).cuda()
    a0 = ops.amax_of(t)
    for _ in range(300):
        ops.amax_slot(t.device)
    a1 = ops.amax_of(t)
    assert a1.data_ptr() != a0.data_ptr() and float(a1.view(torch.float32)) == float(t.abs().max())
    for bad in (float("nan"), float("inf")):
        dy = dy0.clone()
        dy[11, 5] = bad
        dx = ops.linear_bwd_data(dy.cuda(), w, mode="f16x3s").cpu()
        dw, _ = ops.linear_bwd_weight(dy.cuda(), x, mode="f16x3s")
        assert not torch.isfinite(dx[11]).all() and not torch.isfinite(dw[5].cpu()).all()
        ok_rows = torch.arange(M) != 11
        assert torch.isfinite(dx[ok_rows]).all()


def test_fc_out_and_bce():
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    M, K, B = 3000, 256, 3
    h = F.relu(torch.randn(M, K, generator=g))
    w = torch.randn(K, generator=g) / 16
    b = torch.randn(1, generator=g)
    z = ops.fc_out_fwd(h.cuda(), w.cuda(), b.cuda())
    z_ref = h.double() @ w.double() + b.double()
    assert G.rel_err(z.cpu().numpy(), z_ref.numpy()) < 1e-6
    dz = torch.randn(M, generator=g)
    dh, dw, db = ops.fc_out_bwd(h.cuda(), w.cuda(), dz.cuda())
    assert G.rel_err(dh.cpu().numpy(), (dz[:, None] * w[None, :] * (h > 0)).numpy()) < 1e-6
    assert G.rel_err(dw.cpu().numpy(), (dz.double() @ h.double()).numpy()) < 1e-6
    assert abs(db.item() - dz.double().sum().item()) < 1e-4
    # row_map: rows were processed in a permuted order, logits / dlogits live in the caller's order
    perm = torch.randperm(M, generator=g).to(torch.int32)
    zp = ops.fc_out_fwd(h.cuda(), w.cuda(), b.cuda(), perm.cuda())
    assert torch.equal(zp.cpu()[perm.long()], z.cpu())
    dh2, dw2, db2 = ops.fc_out_bwd(h.cuda(), w.cuda(), dz.cuda(), perm.cuda())
    gperm = dz[perm.long()]
    assert G.rel_err(dh2.cpu().numpy(), (gperm[:, None] * w[None, :] * (h > 0)).numpy()) < 1e-6
    assert G.rel_err(dw2.cpu().numpy(), (gperm.double() @ h.double()).numpy()) < 1e-6
    logits = z_ref.float().view(B, M // B)
    t = (torch.rand(B, M // B, generator=g) < 0.5).float()
    lz = logits.clone().requires_grad_(True)
    ref = F.binary_cross_entropy_with_logits(lz, t, reduction="none").sum(-1).mean()
    ref.backward()
    loss, dl = ops.bce_logits_sum_mean(logits.cuda(), t.cuda())
    assert abs(loss.item() - ref.item()) < 1e-5 * abs(ref.item())
    assert G.rel_err(dl.cpu().numpy(), lz.grad.numpy()) < 1e-5


@pytest.mark.parametrize("B,dims,Ci,Co", [(2, (9, 7, 11), 1, 16), (1, (8, 8, 8), 1, 32), (2, (10, 6, 12), 16, 32),
                                           (1, (8, 8, 8), 32, 32), (2, (7, 5, 6), 32, 64), (1, (6, 6, 6), 64, 64),
                                           (1, (4, 5, 6), 64, 128), (2, (4, 4, 4), 128, 128)])
def test_conv3d_fwd_bwd(B, dims, Ci, Co):
    ops = _ops()
    g = torch.Generator().manual_seed(Ci * 1000 + Co)
    x = torch.randn(B, Ci, *dims, generator=g).requires_grad_(True)
    w = (torch.randn(Co, Ci, 3, 3, 3, generator=g) / (27 * Ci) ** 0.5).requires_grad_(True)
    b = torch.randn(Co, generator=g).requires_grad_(True)
    y_ref = F.relu(F.conv3d(x.double(), w.double(), b.double(), padding=1))
    dy = torch.randn(y_ref.shape, generator=g)
    pre = F.conv3d(x.double(), w.double(), b.double(), padding=1)
    gx, gw, gb = torch.autograd.grad(pre, (x, w, b), dy.double())
    wf, wb = ops.conv3d_pack_weight(w.detach().cuda())
    y = ops.conv3d_k3(_cl(x.detach()), wf, b.detach().cuda(), relu=True)
    assert G.rel_err(_ncdhw(y).numpy(), y_ref.detach().numpy()) < 3e-6
    for mode in ("f32", "bf16x6", "f16x3"):  # production forward: f16x3 where Ci % 16 == 0, all held to the f32 gate
        y2 = ops.conv3d_k3_fwd(_cl(x.detach()), w.detach().cuda(), b.detach().cuda(), relu=True, mode=mode)
        assert G.rel_err(_ncdhw(y2).numpy(), y_ref.detach().numpy()) < 3e-6, mode
    # weight gradient: exact-f32 MFMA kernel, and the production bf16x3 split (f32 fallback for Ci == 1)
    for mode, tol in (("f32", 3e-6), ("bf16x3", 3e-5 if Ci > 1 else 3e-6)):
        dwp, db = ops.conv3d_k3_bwd_weight(_cl(x.detach()), _cl(dy), mode=mode)
        dw = ops.conv3d_unpack_wgrad(dwp, Ci, Co)
        assert G.rel_err(dw.cpu().numpy(), gw.numpy()) < tol, mode
        assert G.rel_err(db.cpu().numpy(), gb.numpy()) < 3e-6
        dw2, db2 = ops.conv3d_k3_bwd_weight(_cl(x.detach()), _cl(dy), mode=mode, param_layout=True)   # one-launch post-processing
        assert torch.equal(dw2, dw) and torch.equal(db2, db), mode
    dx = ops.conv3d_k3(_cl(dy), wb)
    assert G.rel_err(_ncdhw(dx).numpy(), gx.numpy()) < 3e-6
    if Ci > 1:
        dxm = ops.conv3d_k3(_cl(dy), wb, mask=_cl(x.detach()))
        assert G.rel_err(_ncdhw(dxm).numpy(), (gx * (x > 0)).detach().numpy()) < 3e-6
    # the production backward-data path: f32 fallback for Ci == 1, bf16x3 split otherwise (2^-16 per product)
    for mode, tol in (("f32", 3e-6), ("bf16x3", 5e-5 if Ci > 1 else 3e-6)):
        d2 = ops.conv3d_k3_bwd_data(_cl(dy), w.detach().cuda(), mode=mode)
        assert G.rel_err(_ncdhw(d2).numpy(), gx.numpy()) < tol, mode
        if Ci > 1:
            d3 = ops.conv3d_k3_bwd_data(_cl(dy), w.detach().cuda(), mask=_cl(x.detach()), mode=mode)
            assert G.rel_err(_ncdhw(d3).numpy(), (gx * (x > 0)).detach().numpy()) < tol, mode


@pytest.mark.parametrize("B,dims,Ci,Co,gscale", [(2, (10, 6, 12), 16, 32, 1e-5), (1, (8, 8, 8), 32, 32, 1.0), (2, (7, 5, 6), 32, 64, 3e-8),
                                                  (1, (6, 6, 6), 64, 64, 1e-3), (1, (4, 5, 6), 64, 128, 1e-6), (2, (4, 4, 4), 128, 128, 20.0),
                                                  (2, (17, 16, 24), 16, 32, 1e-4)])
def test_conv3d_backward_at_f32_level_f16x3s(B, dims, Ci, Co, gscale):
    """The encoder's two backward products on the scaled f16 split ("f16x3s": svr_conv3d_k3_bwd_data_f16x3 /
    svr_conv3d_k3_bwd_weight_f16x3) against f64 autograd, with a GRADIENT-like dout (magnitude `gscale`, per-voxel spread
    e^(+-2 sigma): outside f16's range without the scale).  Held to the exact-f32 kernel's gate (3e-6) and to <= 3x its own
    error; bf16x3 is an order of magnitude away.  The last case takes the persistent 32-column kernel (>= 512 bricks)."""
    ops = _ops()
    g = torch.Generator().manual_seed(Ci * 1000 + Co + 7)
    x = F.relu(torch.randn(B, Ci, *dims, generator=g)).requires_grad_(True)
    w = (torch.randn(Co, Ci, 3, 3, 3, generator=g) / (27 * Ci) ** 0.5).requires_grad_(True)
    pre = F.conv3d(x.double(), w.double(), None, padding=1)
    dy = torch.randn(pre.shape, generator=g) * gscale * torch.exp(2 * torch.randn(B, 1, *dims, generator=g))
    gx, gw = torch.autograd.grad(pre, (x, w), dy.double())
    dyc, xc, wc = _cl(dy), _cl(x.detach()), w.detach().cuda()
    err = {}
    for mode in ("f32", "f16x3s", "bf16x3"):
        dw, db = ops.conv3d_k3_bwd_weight(xc, dyc, mode=mode, param_layout=True)
        dwp, _ = ops.conv3d_k3_bwd_weight(xc, dyc, mode=mode)
        assert torch.equal(ops.conv3d_unpack_wgrad(dwp, Ci, Co), dw), mode
        assert G.rel_err(db.cpu().numpy(), dy.double().sum((0, 2, 3, 4)).numpy()) < 3e-6
        d2 = ops.conv3d_k3_bwd_data(dyc, wc, mode=mode)
        d3 = ops.conv3d_k3_bwd_data(dyc, wc, mask=xc, mode=mode)
        err[mode] = (G.rel_err(_ncdhw(d2).numpy(), gx.numpy()), G.rel_err(dw.cpu().numpy(), gw.numpy()),
                     G.rel_err(_ncdhw(d3).numpy(), (gx * (x > 0)).detach().numpy()))
        if mode == "f16x3s":
            assert float(d3._svr_amax.view(torch.float32)) == float(d3.abs().max()), "fused |max| of din"
    for i, what in enumerate(("din", "dW", "din masked")):
        assert err["f32"][i] < 3e-6 and err["f16x3s"][i] < 3e-6, (what, err)
        assert err["f16x3s"][i] < 3 * err["f32"][i] + 1e-7, (what, err)
        assert err["bf16x3"][i] > 4 * err["f16x3s"][i], (what, err)


@pytest.mark.parametrize("B,dims,C,pool", [(2, (9, 7, 10), 16, True), (1, (8, 8, 8), 32, True), (2, (5, 4, 6), 64, True),
                                           (3, (4, 4, 4), 128, True), (2, (3, 2, 2), 128, False)])
def test_bn_pool_fwd_bwd(B, dims, C, pool):
    ops = _ops()
    g = torch.Generator().manual_seed(C + B)
    pre = torch.randn(B, C, *dims, generator=g)
    x = F.relu(pre).requires_grad_(True)       # BN input is a ReLU output (many exact zeros -> pool ties)
    gamma = (torch.rand(C, generator=g) + 0.5).requires_grad_(True)
    beta = (torch.rand(C, generator=g) - 0.5).requires_grad_(True)
    rm, rv = torch.zeros(C), torch.ones(C)
    y_ref = F.batch_norm(x, rm, rv, gamma, beta, True, 0.1, 1e-5)
    rm_g, rv_g = torch.zeros(C).cuda(), torch.ones(C).cuda()
    y, pooled, argmax, ss, mean = ops.bn_forward(_cl(x.detach()), gamma.detach().cuda(), beta.detach().cuda(), rm_g, rv_g,
                                                 True, want_pool=pool)
    assert G.rel_err(_ncdhw(y).numpy(), y_ref.detach().numpy()) < 2e-6
    assert G.rel_err(rm_g.cpu().numpy(), rm.numpy()) < 1e-6 and G.rel_err(rv_g.cpu().numpy(), rv.numpy()) < 1e-6
    dy = torch.randn(y_ref.shape, generator=g)
    obj = (y_ref * dy).sum()
    dp = None
    if pool:
        p_ref = F.max_pool3d(y_ref, 2)
        assert G.rel_err(_ncdhw(pooled).numpy(), p_ref.detach().numpy()) < 2e-6
        dp = torch.randn(p_ref.shape, generator=g)
        obj = obj + (p_ref * dp).sum()
    gx, gg, gb = torch.autograd.grad(obj, (x, gamma, beta))
    dx, dgamma, dbeta = ops.bn_backward(_cl(x.detach()), _cl(dy), _cl(dp) if pool else None, argmax, mean, ss, relu_mask=True)
    ref_dx = gx * (x > 0)
    assert G.rel_err(_ncdhw(dx).numpy(), ref_dx.detach().numpy()) < 1e-5
    assert G.rel_err(dgamma.cpu().numpy(), gg.numpy()) < 1e-5
    assert G.rel_err(dbeta.cpu().numpy(), gb.numpy()) < 1e-5
    # eval mode uses the running statistics
    ye_ref = F.batch_norm(x.detach(), rm, rv, gamma.detach(), beta.detach(), False, 0.1, 1e-5)
    ye = ops.bn_forward(_cl(x.detach()), gamma.detach().cuda(), beta.detach().cuda(), rm_g, rv_g, False, want_pool=False)[0]
    assert G.rel_err(_ncdhw(ye).numpy(), ye_ref.numpy()) < 2e-6
    # ... and its backward has no batch-mean terms (autograd of F.batch_norm(training=False))
    xe = x.detach().clone().requires_grad_(True)
    ge_, be_ = gamma.detach().clone().requires_grad_(True), beta.detach().clone().requires_grad_(True)
    obj_e = (F.batch_norm(xe, rm, rv, ge_, be_, False, 0.1, 1e-5) * dy).sum()
    gxe, gge, gbe = torch.autograd.grad(obj_e, (xe, ge_, be_))
    _, _, _, ss_e, mean_e = ops.bn_forward(_cl(x.detach()), gamma.detach().cuda(), beta.detach().cuda(), rm_g, rv_g, False,
                                           want_pool=False)
    dxe, dge, dbe = ops.bn_backward(_cl(x.detach()), _cl(dy), None, None, mean_e, ss_e, relu_mask=True, training=False)
    assert G.rel_err(_ncdhw(dxe).numpy(), (gxe * (x > 0)).detach().numpy()) < 1e-5
    assert G.rel_err(dge.cpu().numpy(), gge.numpy()) < 1e-5 and G.rel_err(dbe.cpu().numpy(), gbe.numpy()) < 1e-5


@pytest.mark.parametrize("B,dims,training", [(2, (9, 7, 11), True), (1, (16, 16, 16), True), (2, (19, 10, 21), True),
                                             (1, (8, 8, 8), False), (3, (3, 2, 5), True)])
def test_stage1_recomputed_conv_in_bn_pool(B, dims, training):
    """stage1.hip: y = BN(relu(conv_in(x))), pooled, and the backward (dgamma, dbeta, conv_in's dW / db, optional dconv)
    against float64 torch autograd of the same three modules (model/ifnet.py:126,138,136 as called at :165-169) -- odd,
    non-cubic volumes (cut pool cells, partial bricks), batch > 1, eval mode."""
    ops = _ops()
    g = torch.Generator().manual_seed(B * 100 + dims[0])
    x = (torch.rand(B, 1, *dims, generator=g) < 0.3).float() * torch.rand(B, 1, *dims, generator=g)
    w = (torch.randn(16, 1, 3, 3, 3, generator=g) / 27 ** 0.5).requires_grad_(True)
    b = (torch.randn(16, generator=g) * 0.3).requires_grad_(True)
    gamma = (torch.rand(16, generator=g) + 0.5).requires_grad_(True)
    beta = (torch.rand(16, generator=g) - 0.5).requires_grad_(True)
    rm, rv = torch.rand(16, generator=g) * 0.2, torch.rand(16, generator=g) + 0.5
    xd = x.double().requires_grad_(True)
    a_ref = F.relu(F.conv3d(xd, w.double(), b.double(), padding=1))
    rm_r, rv_r = rm.double().clone(), rv.double().clone()
    y_ref = F.batch_norm(a_ref, rm_r, rv_r, gamma.double(), beta.double(), training, 0.1, 1e-5)
    pool = min(dims) >= 2
    rm_g, rv_g = rm.clone().cuda(), rv.clone().cuda()
    assert ops.stage1_supported(_cl(x), 16)
    y, pooled, argmax, ss, mean, wp = ops.stage1_fwd(_cl(x), w.detach().cuda(), b.detach().cuda(), gamma.detach().cuda(),
                                                     beta.detach().cuda(), rm_g, rv_g, training, want_pool=pool)
    assert G.rel_err(_ncdhw(y).numpy(), y_ref.detach().numpy()) < 3e-6
    if training:
        assert G.rel_err(rm_g.cpu().numpy(), rm_r.numpy()) < 2e-6 and G.rel_err(rv_g.cpu().numpy(), rv_r.numpy()) < 2e-6
    dy = torch.randn(y_ref.shape, generator=g)
    obj = (y_ref * dy.double()).sum()
    dp = None
    if pool:
        p_ref = F.max_pool3d(y_ref, 2)
        assert G.rel_err(_ncdhw(pooled).numpy(), p_ref.detach().numpy()) < 3e-6
        # the argmax routes the pooled gradient: same voxel as torch's (checked through the gradient below)
        dp = torch.randn(p_ref.shape, generator=g)
        obj = obj + (p_ref * dp.double()).sum()
    gx, gw, gb, gg, gbeta = torch.autograd.grad(obj, (xd, w, b, gamma, beta))
    for want_dout in (False, True):
        dgamma, dbeta, dwp, db, dout = ops.stage1_bwd(_cl(x), wp, b.detach().cuda(), _cl(dy), _cl(dp) if pool else None,
                                                      argmax if pool else None, mean, ss, relu_mask=True, training=training,
                                                      want_dout=want_dout)
        dw = dwp                                      # already in the parameter's layout (16,1,3,3,3)
        assert tuple(dw.shape) == (16, 1, 3, 3, 3) and G.rel_err(dw.cpu().numpy(), gw.numpy()) < 1e-5
        assert G.rel_err(db.cpu().numpy(), gb.numpy()) < 1e-5
        assert G.rel_err(dgamma.cpu().numpy(), gg.numpy()) < 1e-5 and G.rel_err(dbeta.cpu().numpy(), gbeta.numpy()) < 1e-5
        if want_dout:       # d(loss)/d(x) through the separate backward-data kernel, as the encoder's backward does
            dx = ops.conv3d_k3_bwd_data(dout, w.detach().cuda(), mode="f32")
            assert G.rel_err(_ncdhw(dx).numpy(), gx.numpy()) < 1e-5
    # both arithmetics of the recomputed convolution (default: the f16 split of the other forward convolutions; "f32": exact)
    for mode in ("f32", "f16x3"):
        rm3, rv3 = rm.clone().cuda(), rv.clone().cuda()
        y3, p3, am3, ss3, mean3, wp3 = ops.stage1_fwd(_cl(x), w.detach().cuda(), b.detach().cuda(), gamma.detach().cuda(),
                                                       beta.detach().cuda(), rm3, rv3, training, want_pool=pool, mode=mode)
        assert wp3.svr_stage1_arith == (1 if mode == "f16x3" else 0)
        assert G.rel_err(_ncdhw(y3).numpy(), y_ref.detach().numpy()) < 3e-6, mode
        g3 = ops.stage1_bwd(_cl(x), wp3, b.detach().cuda(), _cl(dy), _cl(dp) if pool else None, am3 if pool else None, mean3, ss3,
                            relu_mask=True, training=training)
        assert G.rel_err(g3[2].cpu().numpy(), gw.numpy()) < 1e-5 and G.rel_err(g3[0].cpu().numpy(), gg.numpy()) < 1e-5, mode
    # the same results as the separate kernels it replaces (conv_in + statistics, BatchNorm + pool) on the same input
    if training:
        rm2, rv2 = rm.clone().cuda(), rv.clone().cuda()
        a2, st2 = ops.conv3d_c1_fwd_stats(_cl(x), w.detach().cuda(), b.detach().cuda(), relu=True)
        y2, p2, am2, _, _ = ops.bn_forward(a2, gamma.detach().cuda(), beta.detach().cuda(), rm2, rv2, True, want_pool=pool, stats=st2)
        assert G.rel_err(y.cpu().numpy(), y2.cpu().numpy()) < 3e-6
        if pool:
            assert G.rel_err(pooled.cpu().numpy(), p2.cpu().numpy()) < 3e-6


def test_morton_order_is_a_permutation_and_gather_is_order_independent():
    ops = _ops()
    chans = [1, 16, 32, 64, 128, 128]
    B, dims, N = 3, (16, 16, 16), 1000
    vols = _rand_levels(B, dims, chans, 21)
    g = torch.Generator().manual_seed(22)
    pts = (torch.rand(B, N, 3, generator=g) - 0.5) * 1.2
    pts[0, 0, 0] = float("nan")
    order, spts = ops.morton_order(pts.cuda(), want_sorted=True)
    o = order.cpu().long()
    assert torch.equal(spts.cpu().reshape(-1, 3).nan_to_num(7.0), pts.reshape(-1, 3)[o].nan_to_num(7.0))
    assert sorted(o.tolist()) == list(range(B * N))                 # a permutation
    assert torch.equal(o // N, torch.arange(B).repeat_interleave(N))  # samples stay contiguous
    # Morton keys non-decreasing inside a sample
    q = ((pts.reshape(-1, 3)[o] + 0.5) * 64).clamp(0, 63).nan_to_num(0).long()
    def spread(v):
        out = torch.zeros_like(v)
        for bit in range(6):
            out |= ((v >> bit) & 1) << (3 * bit)
        return out
    key = spread(q[:, 2]) | (spread(q[:, 1]) << 1) | (spread(q[:, 0]) << 2)
    key = key + (o // N) * (1 << 18)
    assert bool((key[1:] >= key[:-1]).all())
    layout = ops.FeatureLayout(chans)
    vols_g = [_cl(v) for v in vols]
    pts_g = pts.nan_to_num(0.1).cuda()
    a = ops.gather_fwd(vols_g, pts_g, layout, float(np.float32(0.0722)), False)
    b = ops.gather_fwd(vols_g, pts_g, layout, float(np.float32(0.0722)), False, order=order)
    assert torch.equal(a, b)
    gf = torch.randn(B * N, layout.row_stride, generator=g).cuda()
    g1 = [torch.zeros_like(v) for v in vols_g]
    g2 = [torch.zeros_like(v) for v in vols_g]
    ops.gather_bwd(vols_g, g1, pts_g, gf, layout, float(np.float32(0.0722)), False)
    ops.gather_bwd(vols_g, g2, pts_g, gf, layout, float(np.float32(0.0722)), False, order=order)
    for x1, x2 in zip(g1, g2):
        assert G.rel_err(x2.cpu().numpy(), x1.cpu().numpy()) < 1e-5


def test_conv_in_fused_bn_statistics():
    """svr_conv3d_c1_fwd_stats: same output as the plain conv_in forward, and the statistics svr_bn_stats would compute."""
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    for B, dims, Co in ((2, (9, 7, 11), 16), (1, (16, 16, 16), 16), (1, (8, 8, 8), 32)):
        x = (torch.rand(B, *dims, 1, generator=g) < 0.3).float()
        w = torch.randn(Co, 1, 3, 3, 3, generator=g) * 0.3
        b = torch.randn(Co, generator=g) * 0.1
        ref = ops.conv3d_k3_fwd(x.cuda(), w.cuda(), b.cuda(), relu=True)
        out, stats = ops.conv3d_c1_fwd_stats(x.cuda(), w.cuda(), b.cuda(), relu=True)
        assert torch.equal(out, ref)
        flat = ref.double().view(-1, Co)
        mean, var = flat.mean(0), flat.var(0, unbiased=False)
        assert G.rel_err(stats[:Co].cpu().numpy(), mean.cpu().numpy()) < 1e-6
        assert G.rel_err(stats[Co:].cpu().numpy(), var.cpu().numpy()) < 1e-6


@pytest.mark.parametrize("dims,align,N,B", [((8, 8, 8), False, 3000, 2), ((5, 6, 4), True, 500, 3), ((16, 16, 16), False, 700, 1)])
def test_projected_scatter_of_dh_rows(dims, align, N, B):
    """svr_gather_project_bwd: dP[b][v][j][0:256] = sum over the items (n, j) that touch voxel v of w * dh[b*N+n][0:256]
    -- the adjoint of sampling a 256-channel volume at the j-th displaced positions -- against CPU autograd of
    grid_sample (1e-5), for the (cell, j) item order."""
    ops = _ops()
    net_res = 32 if align else 128
    disp = float(np.float32(O.ARCH[net_res]["disp"]))
    g = torch.Generator().manual_seed(61 + N)
    pts = (torch.rand(B, N, 3, generator=g) - 0.5) * 1.2
    dh = torch.randn(B * N, 256, generator=g)
    items = ops.item_order(pts.cuda(), dims, disp, align, with_j=True)
    ic = items.cpu().long()
    assert sorted(ic.tolist()) == list(range(7 * B * N))
    dP = ops.gather_project_bwd(pts.cuda(), dh.cuda(), dims, items, disp, align).cpu()          # (B, V, 7, 256)
    grid = O.sample_grid(pts, net_res)                                                          # (B,1,7,N,3)
    for j in range(7):
        vol = torch.zeros(B, 256, *dims, requires_grad=True)
        out = F.grid_sample(vol, grid[:, :, j:j + 1], mode="bilinear", padding_mode="zeros", align_corners=align)  # (B,256,1,1,N)
        (out[:, :, 0, 0].permute(0, 2, 1) * dh.view(B, N, 256)).sum().backward()
        ref = vol.grad.permute(0, 2, 3, 4, 1).reshape(B, -1, 256)
        assert G.rel_err(dP[:, :, j].numpy(), ref.numpy()) < 1e-5, j
    # two-pass form (svr_gather_project_plan / _bwd2: run sums stored, then one gather-form pass per voxel row): the same
    # values without float atomics, every row of dP written (it starts as uninitialised memory), bit-reproducible
    plan = ops.project_plan(pts.cuda(), dims, disp, align)
    assert torch.equal(plan.items, items)
    runs = [ops.gather_project_bwd(pts.cuda(), dh.cuda(), dims, plan, disp, align) for _ in range(2)]
    assert torch.equal(runs[0], runs[1])
    assert G.rel_err(runs[0].cpu().numpy(), dP.numpy()) < 1e-5
    assert bool(torch.isfinite(runs[0]).all())


@pytest.mark.parametrize("chans,ac", [([1, 16, 32, 64, 128, 128], False), ([1, 32, 64, 128], True)], ids=["arch128", "arch32"])
@pytest.mark.parametrize("B,dims,N,spread", [(2, (16, 16, 16), 777, 1.0), (1, (35, 26, 28), 500, 1.3), (3, (32, 32, 32), 1, 1.0),
                                             (2, (24, 24, 24), 1280, 1.05)])
def test_fused_gather_fc0_equals_gather_then_linear(B, dims, N, spread, chans, ac):
    """gather_fc0.hip against the two kernels it fuses (model/ifnet.py:156-197 + fc_0 :43-45,55): the kept feature
    columns are bit-identical to svr_gather_trilinear_fwd's (same geometry, same corner order), h0 agrees with
    svr_linear_fwd_f16x3 on those rows to f32 rounding of a different summation order over K (1e-6 of the row scale),
    and with the float64 product of the oracle's grid_sample features to 2e-6 (the gate of the GEMM itself)."""
    ops = _ops()
    vols = _rand_levels(B, dims, chans, 15)
    g = torch.Generator().manual_seed(16)
    pts = ((torch.rand(B, N, 3, generator=g) - 0.5) * spread).cuda()
    layout = ops.FeatureLayout(chans)
    vols_g = [_cl(v) for v in vols]
    disp = float(np.float32(0.0722))
    w = (torch.randn(256, layout.row_stride, generator=g) / 30).cuda()
    w[:, layout.width:] = 0
    bias = torch.randn(256, generator=g).cuda()
    assert ops.gather_fc0_supported(vols_g, pts, layout, disp, ac)
    rows = ops.gather_fwd(vols_g, pts, layout, disp, ac)
    want = ops.linear_fwd(rows, w, bias, relu=True)
    keep = [l for l, c in enumerate(chans) if c < 128]
    h0, kept = ops.gather_fc0_fwd(vols_g, pts, layout, disp, ac, w, bias, relu=True, keep_levels=keep)
    for l in keep:
        a, b = layout.col[l], layout.col[l] + 7 * chans[l]
        assert torch.equal(kept[:, a:b], rows[:, a:b]), f"level {l}"
    assert torch.all(kept[:, layout.width:] == 0)
    # compact kept-column matrix (what the training step keeps): the same bits at the KeptLayout's columns, zero padding,
    # written into a caller-provided buffer that starts as NaN (every column must be written), same h0
    klay = layout.subset(keep)
    assert klay.width == 7 * sum(chans[l] for l in keep) and klay.row_stride % 32 == 0 and klay.row_stride - klay.width < 32
    assert all((klay.col[l] >= 0) == (l in keep) for l in range(len(chans)))
    buf = torch.full((B * N, klay.row_stride), float("nan"), device="cuda")
    h0c, keptc = ops.gather_fc0_fwd(vols_g, pts, layout, disp, ac, w, bias, relu=True, keep_levels=keep, keep_layout=klay,
                                    rows_out=buf)
    assert keptc.data_ptr() == buf.data_ptr() and torch.equal(h0c, h0)
    for l in keep:
        a, b = layout.col[l], layout.col[l] + 7 * chans[l]
        assert torch.equal(keptc[:, klay.col[l]: klay.col[l] + 7 * chans[l]], rows[:, a:b]), f"compact level {l}"
    assert torch.all(keptc[:, klay.width:] == 0)
    fc = klay.full_cols
    assert torch.equal(keptc[:, :klay.width], rows[:, fc[:klay.width].cuda()]) and bool((fc[klay.width:] == layout.width).all())
    scale = float(want.abs().max())
    assert float((h0 - want).abs().max()) <= 1e-6 * scale
    ref = torch.relu(rows.double().cpu() @ w.double().cpu().t() + bias.double().cpu())
    assert float((h0.cpu().double() - ref).abs().max()) <= 2e-6 * scale
    # inference form: no rows, no ReLU
    h0n, none = ops.gather_fc0_fwd(vols_g, pts, layout, disp, ac, w, bias, relu=False)
    assert none is None
    assert torch.equal(torch.relu(h0n), h0)


def test_fused_gather_fc0_with_non_finite_volume_values():
    """Documented difference (gather_fc0.hip, produce_slab): ATen's grid_sample SKIPS a corner outside the volume; the fused
    kernel reads the CLAMPED (existing) voxel and multiplies by a weight of exactly 0 -- identical for finite volumes, but
    0 * inf = NaN.  A clamped corner is always the same voxel as an in-range corner of the same sample UNLESS the whole
    sample lies outside the volume, so the two differ exactly there: ATen returns 0, the fused kernel NaN when the nearest
    border voxel is non-finite.  The unfused gather (svr_gather_trilinear_fwd, what API-compat callers get) follows ATen.
    Pinned: (a) a non-finite voxel no sample reaches changes nothing; (b) with an inf border plane the unfused rows equal
    F.grid_sample bit for bit, the fused h0 is non-finite wherever ATen's features are AND in the rows whose samples lie
    entirely beyond that plane (finite zeros in ATen), and every other row keeps its bits."""
    import torch.nn.functional as F
    ops = _ops()
    chans = [1, 16, 32, 64, 128, 128]
    B, dims, N = 1, (16, 16, 16), 600
    vols = _rand_levels(B, dims, chans, 41)
    g = torch.Generator().manual_seed(42)
    pts = (torch.rand(B, N, 3, generator=g) - 0.5) * 0.6          # interior: nobody comes near the border voxels
    jit = (torch.rand(60, 3, generator=g) - 0.5) * torch.tensor([0.0, 0.2, 0.2])
    pts[0, :40] = torch.tensor([0.499, 0.0, 0.0]) + jit[:40]       # straddle the last plane of the first spatial axis
    pts[0, 40:60] = torch.tensor([0.62, 0.0, 0.0]) + jit[40:]      # all 7 samples entirely beyond it (source index > 16)
    layout = ops.FeatureLayout(chans)
    disp = float(np.float32(0.0722))
    w = (torch.randn(256, layout.row_stride, generator=g) / 30).cuda()
    w[:, layout.width:] = 0
    bias = torch.randn(256, generator=g).cuda()
    base = ops.gather_fc0_fwd([_cl(v) for v in vols], pts.cuda(), layout, disp, False, w, bias, relu=False)[0]
    assert bool(torch.isfinite(base).all())
    # (a) a corner voxel of the level-1 volume far away from every sample
    va = [v.clone() for v in vols]
    va[1][0, :, 0, 0, 0] = float("inf")
    ha = ops.gather_fc0_fwd([_cl(v) for v in va], pts.cuda(), layout, disp, False, w, bias, relu=False)[0]
    assert torch.equal(ha, base)
    # (b) the whole last plane of the first spatial axis of level 1
    vb = [v.clone() for v in vols]
    vb[1][0, :, -1] = float("inf")
    vols_b = [_cl(v) for v in vb]
    rows = ops.gather_fwd(vols_b, pts.cuda(), layout, disp, False)
    grid = torch.stack([2 * pts[..., 2], 2 * pts[..., 1], 2 * pts[..., 0]], -1).view(B, 1, 1, N, 3)
    want1 = F.grid_sample(vb[1], grid, mode="bilinear", padding_mode="zeros", align_corners=False)[0, :, 0, 0].t()   # (N, 16), j = 0
    got1 = rows[:, layout.col[1]: layout.col[1] + 16].cpu()
    assert bool(((got1 == want1) | (torch.isnan(got1) & torch.isnan(want1))).all())   # the unfused gather IS ATen, inf / nan included
    aten_bad = ~torch.isfinite(rows[:, layout.col[1]: layout.col[1] + 7 * 16]).all(dim=1).cpu()
    assert bool(aten_bad[:40].any()) and not bool(aten_bad[40:].any())  # beyond the plane ATen returns finite zeros
    assert bool((rows[40:60, layout.col[1]: layout.col[1] + 7 * 16] == 0).all())
    hb = ops.gather_fc0_fwd(vols_b, pts.cuda(), layout, disp, False, w, bias, relu=False)[0]
    bad = ~torch.isfinite(hb).all(dim=1).cpu()
    beyond = torch.zeros(N, dtype=torch.bool)
    beyond[40:60] = True
    assert torch.equal(bad, aten_bad | beyond)                       # the documented difference, and nothing else
    assert torch.equal(hb[~bad.cuda()], base[~bad.cuda()])


@pytest.mark.parametrize("bwd", ["f16x3s", "bf16x3"])
def test_prepared_weight_planes_give_the_same_bits_and_expire(bwd):
    """include/svr_hip.h PREPARE / RUN: a split-precision layer run on a workspace prepared ahead (data operand NULL, then W
    NULL) returns the bits of the one-call form, for the forward entry points and the backward-data ones of both split modes
    (the f32-level "f16x3s" default and "bf16x3"); ops.PreparedWeights serves a parameter's planes only until the parameter is
    modified in place (an optimizer step) or invalidate()."""
    ops = _ops()
    saved = (ops.BACKWARD_GEMM, ops.BACKWARD_CONV)
    ops.BACKWARD_GEMM = ops.BACKWARD_CONV = bwd
    kc, kl = ("cbh", "lbh") if bwd == "f16x3s" else ("cb", "lb")
    g = torch.Generator().manual_seed(77)
    x = torch.randn(2, 6, 5, 7, 32, generator=g).cuda()
    w = (torch.randn(64, 32, 3, 3, 3, generator=g) / 30).cuda()
    b = torch.randn(64, generator=g).cuda()
    dy = torch.randn(2, 6, 5, 7, 64, generator=g).cuda()
    xm = torch.randn(3000, 256, generator=g).cuda()
    wl = (torch.randn(256, 256, generator=g) / 16).cuda()
    bl = torch.randn(256, generator=g).cuda()
    try:
        base = (ops.conv3d_k3_fwd(x, w, b), ops.conv3d_k3_bwd_data(dy, w, mask=x), ops.linear_fwd(xm, wl, bl),
                ops.linear_bwd_data(xm, wl, mask=xm))
        prep = ops.PreparedWeights()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            prep.begin()
            prep.add_conv(w)
            prep.add_linear(wl)
            prep.finish(side)
        ops.set_prepared(prep)
        assert all(prep.lookup(k, t) is not None for k, t in (("cf", w), (kc, w), ("lf", wl), (kl, wl)))
        got = (ops.conv3d_k3_fwd(x, w, b), ops.conv3d_k3_bwd_data(dy, w, mask=x), ops.linear_fwd(xm, wl, bl),
               ops.linear_bwd_data(xm, wl, mask=xm))
        for a_, b_ in zip(base, got):
            assert torch.equal(a_, b_)
        # another mode, another tensor: misses
        assert ops._lookup("cf", w, "f32", "f16x3") is None and prep.lookup("cf", w.clone()) is None
        # an in-place update (optimizer step) expires the planes: the op prepares its own again and follows the new values
        w.mul_(2.0)
        assert prep.lookup("cf", w) is None and prep.lookup(kc, w) is None and prep.lookup("lf", wl) is not None
        zb = torch.zeros_like(b)
        y2 = ops.conv3d_k3_fwd(x, w, zb, relu=False)
        assert G.rel_err(y2.cpu().numpy(), (2 * ops.conv3d_k3_fwd(x, w / 2, zb, relu=False)).cpu().numpy()) < 1e-6
        prep.invalidate()
        assert prep.lookup("lf", wl) is None
    finally:
        ops.set_prepared(None)
        ops.BACKWARD_GEMM, ops.BACKWARD_CONV = saved


@pytest.mark.parametrize("B,dims,Ci,Co", [(2, (10, 6, 12), 16, 32), (1, (9, 9, 9), 32, 32), (2, (7, 5, 6), 32, 64), (1, (6, 6, 6), 64, 64),
                                          (2, (4, 5, 6), 64, 128), (8, (16, 16, 16), 32, 32)])
def test_conv_epilogue_batchnorm_statistics(B, dims, Ci, Co):
    """svr_conv3d_k3_fwd_f16x3_stats + svr_bn_finalize_parts: the BatchNorm that follows a stage's last convolution gets its
    statistics from the convolution kernel's epilogue (per-workgroup f64 partial sums of the values it stores).  Same output
    bits as the plain call; BatchNorm output, pooled output and running statistics equal the separate statistics pass to f32
    rounding (both sum in f64, in different orders); odd volumes (partial bricks) and every tile variant."""
    ops = _ops()
    g = torch.Generator().manual_seed(Ci + Co + B)
    x = torch.randn(B, *dims, Ci, generator=g).cuda()
    w = (torch.randn(Co, Ci, 3, 3, 3, generator=g) / (27 * Ci) ** 0.5).cuda()
    b = torch.randn(Co, generator=g).cuda()
    gamma, beta = (torch.rand(Co, generator=g) + 0.5).cuda(), (torch.rand(Co, generator=g) - 0.5).cuda()
    y0 = ops.conv3d_k3_fwd(x, w, b, relu=True)
    y1, st = ops.conv3d_k3_fwd(x, w, b, relu=True, want_stats=True)
    assert st is not None and torch.equal(y0, y1)
    n = B * dims[0] * dims[1] * dims[2]
    sums = st.part.sum(0)                                             # (2, Co) float64
    ref = y0.double().reshape(-1, Co)
    assert G.rel_err(sums[0].cpu().numpy(), ref.sum(0).cpu().numpy()) < 1e-6
    assert G.rel_err(sums[1].cpu().numpy(), (ref * ref).sum(0).cpu().numpy()) < 1e-6
    pool = min(dims) >= 2
    rm0, rv0 = torch.zeros(Co).cuda(), torch.ones(Co).cuda()
    rm1, rv1 = torch.zeros(Co).cuda(), torch.ones(Co).cuda()
    a0 = ops.bn_forward(y0, gamma, beta, rm0, rv0, True, want_pool=pool)
    a1 = ops.bn_forward(y1, gamma, beta, rm1, rv1, True, want_pool=pool, stats=st)
    assert G.rel_err(a1[0].cpu().numpy(), a0[0].cpu().numpy()) < 1e-6
    if pool:
        assert G.rel_err(a1[1].cpu().numpy(), a0[1].cpu().numpy()) < 1e-6
    assert G.rel_err(rm1.cpu().numpy(), rm0.cpu().numpy()) < 1e-6 and G.rel_err(rv1.cpu().numpy(), rv0.cpu().numpy()) < 1e-6
    assert G.rel_err(a1[3].cpu().numpy(), a0[3].cpu().numpy()) < 1e-6 and abs(n - B * dims[0] * dims[1] * dims[2]) == 0


@pytest.mark.parametrize("M,N,K", [(4096, 1792, 128), (1000, 1792, 128), (77, 256, 64), (8192, 448, 32)])
def test_linear_bwd_data_with_split_reduction(M, N, K):
    """ops.linear_bwd_data_splitk (the implicit-GEMM kernel's k = 1 mode: few output tiles, the reduction split over workgroups
    and summed in a fixed order) against float64 and against the point MLP's dX kernel; the |max| word it leaves; bit-reproducible."""
    ops = _ops()
    g = torch.Generator().manual_seed(M + N)
    for scale in (1.0, 1e-6):
        dy = (torch.randn(M, N, generator=g) * scale).cuda()
        w = (torch.randn(N, K, generator=g) / N ** 0.5).cuda()
        ref = dy.double() @ w.double()
        got = ops.linear_bwd_data_splitk(dy, w.t().contiguous())
        old = ops.linear_bwd_data(dy, w, mode="f16x3s")
        rel = lambda x: float((x.double() - ref).abs().max() / ref.abs().max())     # noqa: E731
        assert rel(got) < 2e-6 and rel(old) < 3e-6, (rel(got), rel(old))
        assert got._svr_amax.view(torch.float32).item() == float(got.abs().max())
        assert torch.equal(got, ops.linear_bwd_data_splitk(dy, w.t().contiguous()))
