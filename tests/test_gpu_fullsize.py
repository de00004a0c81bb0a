"""GPU parity at the BENCHED shapes (BASELINE.json configs[2] = 128^3 grid x 50 000 points x batch 8, and the
per-GPU shard of configs[4] = UNet -> project(128^3) -> IF-Net, batch 4), against the CPU oracle on the same
seeded synthetic inputs bench.py uses (seed 103 / 105).  These are the sizes where the 32-bit
offset logic of the gather, the side-stream sorts / memsets and the >=512-workgroup grid heuristics are live.

The oracle runs on the GPU box's host cores: ~1 min per test.  Tolerances as in test_gpu_ifnet_parity.py
(logits 1e-4, loss 1e-5, fc_out gradients 1e-5, other gradients bounded by the mask-flip sensitivity of the
reference's own math, tools/gradient_sensitivity.py); gathered features and corner indices are bit-exact."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ifnet_oracle as O
from tests import _golden as G

pytestmark = pytest.mark.gpu

D, N, B = 128, 50000, 8
DISP = float(np.float32(0.0722))


def _synth(seed, B, D, N):
    """bench.py:synth_batch (uniform points), SURVEY 8d cfg3."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = (torch.rand(B, 1, D, D, D, generator=g) < 0.05).float()
    pts = torch.rand(B, N, 3, generator=g) - 0.5
    occ = (torch.rand(B, N, generator=g) < 0.5).float()
    return x, pts, occ


def _model():
    import svr_amd  # noqa: F401
    from svr_amd.model import IFNet
    st = O.name_seeded_state(128)
    m = IFNet(net_res=128)
    m.load_state_dict(st, strict=False)
    return m.cuda().train(), st


def _ncdhw(v):
    return v.cpu().permute(0, 4, 1, 2, 3).contiguous()


def test_config3_gather_bit_exact_on_real_pyramid_and_scatter():
    """Forward gather at 128^3 x 50k x B=8 on the encoder's real pyramid: every level's rows for a strided subset of
    4 167 points per sample equal F.grid_sample bit for bit (CPU, reference op order model/ifnet.py:156-197), the
    corner indices of ALL points are exact, and the 64-bit-offset body gives the same bytes.  Backward scatter at the
    same size in both forms (atomic-free pull plans / float atomics with voxel orders): samples 0 and 7 against CPU
    autograd of grid_sample, 1e-5; the pull form twice: identical bytes."""
    import svr_amd  # noqa: F401
    from svr_amd import _lib, ops
    m, _ = _model()
    x, pts, _ = _synth(103, B, D, N)
    ext = m.ifnet_feature_extractor
    levels = ext.encode_levels(x.cuda())                       # training-mode BN statistics, channels-last
    assert [tuple(v.shape[1:]) for v in levels] == [(128, 128, 128, 1), (128, 128, 128, 16), (64, 64, 64, 32),
                                                    (32, 32, 32, 64), (16, 16, 16, 128), (8, 8, 8, 128)]
    layout = ext._layout
    pg = pts.cuda()
    rows = ops.gather_fwd(levels, pg, layout, DISP, False)
    sel = torch.arange(0, N, 12)                               # 4 167 points per sample
    sub = pts[:, sel]
    got = rows.view(B, N, -1)[:, sel.cuda()].cpu()             # (B, n_sel, FS)
    for l, v in enumerate(levels):
        vc = _ncdhw(v)
        idx = ops.corner_indices(levels, pg, layout, l, DISP, False).cpu()
        ref_idx, _ = O.corner_indices(pts, vc.shape[2:], 128)
        assert torch.equal(idx, ref_idx), f"corner indices, level {l}"
        ref = F.grid_sample(vc, O.sample_grid(sub, 128), mode="bilinear", padding_mode="zeros", align_corners=False)
        C = vc.shape[1]                                        # ref (B, C, 1, 7, n) -> (B, n, 7, C) = the row slice
        ref_rows = ref[:, :, 0].permute(0, 3, 2, 1).reshape(B, sel.numel(), 7 * C)
        assert torch.equal(got[:, :, layout.col[l]: layout.col[l] + 7 * C], ref_rows), f"features, level {l}"
        del vc, ref
    assert torch.all(rows[:, layout.width:] == 0)
    # the wide-offset body (normally only reached at >= 2^31 elements) writes the same bytes
    rows_wide = ops.gather_fwd(levels, pg, layout, DISP, False, flags=_lib.GATHER_WIDE_OFFSETS)
    assert torch.equal(rows, rows_wide)
    del rows_wide, got
    # ---- backward scatter at the same size in all three forms: pull plans for levels 1-3 (+ item orders for the
    # 128-channel levels), item orders everywhere, per-displacement point orders (round-1 path); "auto" picks between the
    # first two per level.  Plan levels start from NaN: every voxel must be written
    from svr_amd.model import ifnet as ifn
    g = torch.Generator().manual_seed(7)
    gfeat = torch.randn(B * N, layout.row_stride, generator=g).cuda()
    results = {}
    for form in ("pull", "items", "atomic"):
        saved_form, ifn.SCATTER_FORM = ifn.SCATTER_FORM, form
        try:
            orders, plans, ready = ifn._level_orders_async(pg, D, D, D, len(levels), False, layout, DISP)
        finally:
            ifn.SCATTER_FORM = saved_form
        if form == "pull":
            assert [p is not None for p in plans] == [False, True, True, True, False, False]
            assert orders[4].numel() == 7 * B * N and orders[5].numel() == 7 * B * N      # joint item orders
        elif form == "items":
            assert all(p is None for p in plans) and all(o.numel() == 7 * B * N for o in orders[1:])
        else:
            assert all(p is None for p in plans) and sum(o is not None and o.numel() == B * N for o in orders) >= 3
        torch.cuda.current_stream().wait_event(ready)
        gvols = [torch.full_like(v, float("nan")) if plans[l] is not None else torch.zeros_like(v)
                 for l, v in enumerate(levels)]
        ops.gather_bwd(levels, gvols, pg, gfeat, layout, DISP, False, level_orders=orders, level_plans=plans)
        results[form] = gvols
    # the pull form is bit-reproducible (fixed summation order)
    saved_form, ifn.SCATTER_FORM = ifn.SCATTER_FORM, "pull"
    try:
        orders, plans, ready = ifn._level_orders_async(pg, D, D, D, len(levels), False, layout, DISP)
    finally:
        ifn.SCATTER_FORM = saved_form
    torch.cuda.current_stream().wait_event(ready)
    again = [torch.empty_like(v) if plans[l] is not None else torch.zeros_like(v) for l, v in enumerate(levels)]
    ops.gather_bwd(levels, again, pg, gfeat, layout, DISP, False, level_orders=orders, level_plans=plans)
    for l in (1, 2, 3):
        assert torch.equal(again[l], results["pull"][l]), f"pull scatter not reproducible, level {l}"
    del again
    perm = layout.reference_permutation()
    valid = perm >= 0
    for b in (0, B - 1):
        vols_c = [_ncdhw(v[b:b + 1]).requires_grad_(True) for v in levels]
        ref = O.gather_features(vols_c, pts[b:b + 1], 128)     # (1, 2583, N), row k = c*7 + j
        w = torch.empty(1, int(valid.sum()), N)
        w[0, perm[valid]] = gfeat[b * N:(b + 1) * N].cpu()[:, valid].t()
        (ref * w).sum().backward()
        for form, gvols in results.items():
            for l, v in enumerate(vols_c):
                e = G.rel_err(_ncdhw(gvols[l][b:b + 1]).numpy(), v.grad.numpy())
                assert e < 1e-5, (form, b, l, e)


def test_config3_training_step_matches_oracle():
    """One full training step at the benched shape (128^3, 50 000 points, batch 8; BatchNorm in training mode
    couples the 8 samples) against the CPU oracle: logits 1e-4, loss 1e-5, fc_out gradients 1e-5, every other
    gradient within the mask-flip gates, BatchNorm running statistics 1e-5."""
    from svr_amd.trainer import bce_with_logits_sum_mean
    m, st = _model()
    x, pts, occ = _synth(103, B, D, N)
    logits = m(x.cuda(), pts.cuda())
    loss = bce_with_logits_sum_mean(logits, occ.cuda())
    loss.backward()
    torch.cuda.synchronize()
    ref_st = O.make_leaf_state(st)
    ref = O.training_step(ref_st, {"input": x, "points": pts, "occupancies": occ})
    ref["loss"].backward()
    e = G.rel_err(logits.detach().cpu().numpy(), ref["logits"].detach().numpy())
    print("config3 logits rel err", e, "loss", loss.item(), ref["loss"].item())
    assert e < 1e-4
    assert abs(loss.item() - ref["loss"].item()) < 1e-5 * abs(ref["loss"].item())
    worst = {}
    for name, p in m.named_parameters():
        r = ref_st[name].grad.double()
        gq = p.grad.detach().cpu().double()
        e = float((gq - r).abs().max() / r.abs().max().clamp_min(1e-30))
        n = abs(gq.norm().item() - r.norm().item()) / (r.norm().item() + 1e-30)
        med = float((gq - r).abs().median() / r.abs().max().clamp_min(1e-30))
        worst[name] = (e, n, med)
        if name.startswith("fc_out"):
            assert e < 1e-5, (name, e)
        if r.norm().item() < 1e-6 * max(ref_st[k].grad.norm().item() for k in ref_st if ref_st[k].grad is not None):
            continue          # conv bias in front of BatchNorm: true gradient 0, rounding noise on both sides
        # conv_in.bias: 16.8 M masked terms per channel that nearly cancel (the BatchNorm behind the ReLU removes most of the
        # mean): its largest element moves by 1.06e-2 of the largest reference element now that stage 1 recomputes conv_in
        # as the f16 split (2^-22 per weight instead of 2^-24: a few more ReLU masks sit on the other side of 0 than the
        # oracle's), norm and median stay where the other gradients are
        assert e < (2e-2 if name.endswith("conv_in.bias") else 1e-2) and n < 5e-3 and med < 2e-3, (name, e, n, med)
    print("config3 worst gradient max-element err", max(v[0] for v in worst.values()),
          "norm err", max(v[1] for v in worst.values()))
    for name, b in m.named_buffers():
        if "running" in name:
            assert G.rel_err(b.cpu().numpy(), ref_st[name].numpy()) < 1e-5, name


def test_config5_per_gpu_shape_matches_oracle():
    """BASELINE configs[4] per-GPU shard: rgb (4,3,256,256) -> UNet -> unproject -> project(dims =
    128^3, kernel 3, sigma 1.5) -> IF-Net, 50 000 points.  The UNet (stock MIOpen ops) is compared at 2e-5; the HIP
    path proper (unproject, splat, blur, IF-Net, loss, backward to sigma / depth) against the oracle fed with the SAME
    depth map: point cloud 2e-6, voxel occupancy 1e-5, logits 1e-4, loss 1e-5, sigma.grad 2e-2."""
    import svr_amd  # noqa: F401
    from oracle import projection_oracle as P
    from oracle import scene_oracle as S
    from svr_amd.trainer import SceneNetTrainer, default_hparams
    Bs = 4
    dims = (128, 128, 128)
    g = torch.Generator(device="cpu").manual_seed(105)
    rgb = torch.rand(Bs, 3, 256, 256, generator=g) * 2 - 1
    target = torch.rand(Bs, 240, 320, generator=g) * 5 + 0.5
    pts = torch.rand(Bs, N, 3, generator=g) - 0.5
    occ = (torch.rand(Bs, N, generator=g) < 0.5).float()
    tr = SceneNetTrainer(default_hparams(), dims=dims)
    ust = S.name_seeded_like(tr.unet.state_dict(), 1.0, "unet.")
    ist = O.name_seeded_state(128)
    tr.unet.load_state_dict(ust, strict=False)
    tr.ifnet.load_state_dict(ist, strict=False)
    tr = tr.cuda().train()
    batch = {"rgb": rgb.cuda(), "depthmap_target": target.cuda(), "points": pts.cuda(), "occupancies": occ.cuda()}
    logits, depth, pc = tr(batch)
    depth.retain_grad()
    loss = tr.losses_and_logging(batch, depth, logits, batch["occupancies"])
    loss.backward()
    torch.cuda.synchronize()
    with torch.no_grad():
        raw = S.unet_forward({k: v.clone() for k, v in ust.items()}, rgb, "full", True)
        zc = F.interpolate(raw, size=320, mode="bilinear")[:, :, 40:280, :].squeeze(1)
        depth_ref = torch.sigmoid(zc) * (7.0 - 0.1953997164964676) + 0.1953997164964676
    assert G.rel_err(depth.detach().cpu().numpy(), depth_ref.numpy()) < 2e-5
    dcpu = depth.detach().cpu().requires_grad_(True)
    sigma = torch.tensor([1.5, 1.5, 1.5], requires_grad=True)
    ref_st = O.make_leaf_state(ist)
    pc_ref = P.norm_grid_space(P.depthmap_to_gridspace(dcpu, 1), dims)
    vox_ref = P.project_forward(pc_ref, dims, sigma, (3, 3, 3))
    logits_ref = O.ifnet_forward(ref_st, vox_ref, pts, 128, training=True)
    loss_ref = S.scene_loss(logits_ref, dcpu, {"occupancies": occ, "depthmap_target": target})
    loss_ref.backward()
    assert G.rel_err(pc.detach().cpu().numpy(), pc_ref.detach().numpy()) < 2e-6
    with torch.no_grad():
        vox = tr.project(pc.detach())
    assert tuple(vox.shape) == (Bs, 1) + dims
    assert G.rel_err(vox.cpu().numpy(), vox_ref.detach().numpy()) < 1e-5
    e = G.rel_err(logits.detach().cpu().numpy(), logits_ref.detach().numpy())
    print("config5 logits rel err", e)
    assert e < 1e-4
    assert abs(loss.item() - loss_ref.item()) < 1e-5 * abs(loss_ref.item())
    assert G.rel_err(tr.project.sigma.grad.cpu().numpy(), sigma.grad.numpy()) < 2e-2
    gd, rd = depth.grad.cpu().numpy().astype(np.float64), dcpu.grad.numpy().astype(np.float64)
    ed = np.abs(gd - rd) / np.abs(rd).max()
    assert np.median(ed) < 1e-4 and np.quantile(ed, 0.99) < 1e-2, (np.median(ed), np.quantile(ed, 0.99), ed.max())
    go = tr.ifnet.fc_out.weight.grad.cpu().numpy().reshape(-1)
    assert G.rel_err(go, ref_st["fc_out.weight"].grad.numpy().reshape(-1)) < 1e-5


@pytest.mark.parametrize("clustered", [False, True], ids=["uniform", "surface"])
def test_config3_fused_gather_fc0_equals_the_two_kernels(clustered):
    """gather_fc0.hip at the benched size (real pyramid of the seeded model, 8 x 50 000 Morton-sorted points, and the
    surface-clustered distribution of bench.py --dist surface): the kept feature columns bit-identical to
    svr_gather_trilinear_fwd's, h0 equal to svr_linear_fwd_f16x3 on those rows to 2e-6 of the largest value -- three
    times in a row (a hazard between the producer and consumer waves would be timing dependent)."""
    import bench                                  # synth_batch: the benchmark's own inputs (repo root is on sys.path)
    from svr_amd import ops
    m, _ = _model()
    batch = bench.synth_batch(103, B, D, N, torch.device("cuda"), "surface" if clustered else "uniform")
    with torch.no_grad():
        levels = m.encode(batch["input"])
        _, pts = ops.morton_order(batch["points"].contiguous(), want_sorted=True)
        ext = m.ifnet_feature_extractor
        w0p, b0 = m._fc0_internal(), m.fc_0.bias
        rows = ops.gather_fwd(levels, pts, ext._layout, ext._disp, ext._align)
        want = ops.linear_fwd(rows, w0p, b0, relu=True)
        scale = float(want.abs().max())
        keep = [l for l, c in enumerate(ext._layout.channels) if c < 128]
        for _ in range(3):
            h0, kept = ops.gather_fc0_fwd(levels, pts, ext._layout, ext._disp, ext._align, w0p, b0, relu=True, keep_levels=keep)
            for l in keep:
                a, b = ext._layout.col[l], ext._layout.col[l] + 7 * ext._layout.channels[l]
                assert torch.equal(kept[:, a:b], rows[:, a:b]), f"level {l}"
            assert torch.all(kept[:, ext._layout.width:] == 0)
            assert float((h0 - want).abs().max()) <= 2e-6 * scale
            del kept
            # the training step's form: compact kept-column matrix (B*N, 800)
            klay = ext._layout.subset(keep)
            assert klay.row_stride == 800
            h0c, keptc = ops.gather_fc0_fwd(levels, pts, ext._layout, ext._disp, ext._align, w0p, b0, relu=True, keep_levels=keep,
                                            keep_layout=klay)
            assert torch.equal(h0c, h0)
            assert torch.equal(keptc[:, :klay.width], rows[:, klay.full_cols[:klay.width].cuda()])
            assert torch.all(keptc[:, klay.width:] == 0)
            del h0, h0c, keptc
