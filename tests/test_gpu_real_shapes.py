"""GPU parity at the REFERENCE'S OWN training / inference shapes (trainer/trainer_ifnet.py:23-24: grid 139 x 104 x 112;
dataset/implicit_dataset.py:35-47: 2 * num_points points clustered around the surface with sigma 0.1 / 0.01;
model/ifnet.py:202-229: the dense lattice of the grid's resolution incl. the +-0.5 planes), against the CPU oracle.

The non-cubic, odd-sized grid is where the MaxPool floor (139 -> 69 -> 34 -> 17 -> 8), the x-strip tails of the pull
scatter, the (D+1)(H+1)(W+1) cell lattices of the plans and the slab tails of the projected scatter are live."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ifnet_oracle as O
from tests import _golden as G

pytestmark = pytest.mark.gpu

DIMS = (139, 104, 112)
DISP = float(np.float32(0.0722))


def _model(train=True):
    import svr_amd  # noqa: F401
    from svr_amd.model import IFNet
    st = O.name_seeded_state(128)
    m = IFNet(net_res=128)
    m.load_state_dict(st, strict=False)
    m = m.cuda()
    return (m.train() if train else m.eval()), st


def _surface_points(B, N, g):
    """Points clustered around random planes, half of them with sigma 0.1 and half with 0.01 -- the shape of the
    reference's training samples (data_processing/mesh_occupancies.py:14-17 + implicit_dataset.py:35-47)."""
    origin = torch.rand(B, 4, 1, 3, generator=g) - 0.5
    u, v = torch.randn(B, 4, 1, 3, generator=g), torch.randn(B, 4, 1, 3, generator=g)
    ab = torch.rand(B, 4, N // 4, 2, generator=g) - 0.5
    p = origin + ab[..., :1] * u * 0.3 + ab[..., 1:] * v * 0.3
    sigma = torch.cat([torch.full((B, 4, N // 8, 1), 0.1), torch.full((B, 4, N // 4 - N // 8, 1), 0.01)], 2)
    return (p + torch.randn(B, 4, N // 4, 3, generator=g) * sigma).reshape(B, -1, 3).clamp(-0.5, 0.5)


def _ncdhw(v):
    return v.cpu().permute(0, 4, 1, 2, 3).contiguous()


def test_training_step_with_backward_at_139x104x112():
    """One production training step (fused gather -> fc_0, projected 128-channel levels, compact kept-column matrix,
    three-stream backward, step arena) on the reference's grid size, B = 2, N = 4 096 surface-clustered points, against
    the oracle: logits 1e-4, loss 1e-5, fc_out gradients 1e-5, every other gradient inside the mask-flip gates, BatchNorm
    running statistics 1e-5.  A second step through the SAME arena gives the same gradients as a step without one."""
    from svr_amd.model import ifnet as ifn
    from svr_amd.trainer import bce_with_logits_sum_mean
    B, N = 2, 4096
    g = torch.Generator().manual_seed(23)
    x = (torch.rand(B, 1, *DIMS, generator=g) < 0.03).float()
    pts = _surface_points(B, N, g)
    occ = (torch.rand(B, N, generator=g) < 0.5).float()
    assert ifn.FUSE_FC0 and ifn.PROJECT_WIDE_LEVELS and ifn.OVERLAP_BACKWARD and ifn.USE_ARENA
    m, st = _model()
    logits = m(x.cuda(), pts.cuda())
    loss = bce_with_logits_sum_mean(logits, occ.cuda())
    loss.backward()
    torch.cuda.synchronize()
    arena = m.ifnet_feature_extractor._arena
    assert arena.nbytes() > 0 and not arena._leased                  # used, and handed back by the backward
    ref_st = O.make_leaf_state(st)
    ref = O.training_step(ref_st, {"input": x, "points": pts, "occupancies": occ})
    ref["loss"].backward()
    assert G.rel_err(logits.detach().cpu().numpy(), ref["logits"].detach().numpy()) < 1e-4
    assert abs(loss.item() - ref["loss"].item()) < 1e-5 * abs(ref["loss"].item())
    top = max(ref_st[k].grad.norm().item() for k in ref_st if ref_st[k].grad is not None)
    for name, p in m.named_parameters():
        r = ref_st[name].grad.double()
        q = p.grad.detach().cpu().double()
        e = float((q - r).abs().max() / r.abs().max().clamp_min(1e-30))
        n = abs(q.norm().item() - r.norm().item()) / (r.norm().item() + 1e-30)
        if name.startswith("fc_out"):
            assert e < 1e-5, (name, e)
        if r.norm().item() < 1e-6 * top:
            continue              # conv bias in front of BatchNorm: true gradient 0, rounding noise on both sides
        assert e < 1e-2 and n < 5e-3, (name, e, n)
    for name, b in m.named_buffers():
        if "running" in name:
            assert G.rel_err(b.cpu().numpy(), ref_st[name].numpy()) < 1e-5, name
    # the arena is reused by the next step and does not change results: same input again (BatchNorm buffers moved, which
    # does not enter training-mode outputs) vs a fresh model without an arena
    first = {n: p.grad.detach().clone() for n, p in m.named_parameters()}
    grown = arena.grown
    for p in m.parameters():
        p.grad = None
    bce_with_logits_sum_mean(m(x.cuda(), pts.cuda()), occ.cuda()).backward()
    torch.cuda.synchronize()
    assert arena.grown == grown                                       # no new buffers in the second step
    saved, ifn.USE_ARENA = ifn.USE_ARENA, False
    try:
        m2, _ = _model()
        bce_with_logits_sum_mean(m2(x.cuda(), pts.cuda()), occ.cuda()).backward()
        torch.cuda.synchronize()
        assert m2.ifnet_feature_extractor._arena.nbytes() == 0
    finally:
        ifn.USE_ARENA = saved
    for (n, p), (_, p2) in zip(m.named_parameters(), m2.named_parameters()):
        d = float((p.grad - first[n]).norm() / first[n].norm().clamp_min(1e-30))
        d2 = float((p2.grad - first[n]).norm() / first[n].norm().clamp_min(1e-30))
        assert d < 1e-4 and d2 < 1e-4, (n, d, d2)                     # float-atomic order of level 3 / level 4 only


def test_scatter_forms_at_139x104x112_against_cpu_autograd():
    """The backward scatter on the real pyramid of the 139 x 104 x 112 grid with surface-clustered points, in every form
    the step can take -- pull plans (levels 1-3), joint item orders (all levels), the compact kept-column layout, and the
    projected scatter of 256-wide rows (levels 4-5: atomic and two-pass) -- against CPU autograd of F.grid_sample, 1e-5."""
    import svr_amd  # noqa: F401
    from svr_amd import ops
    from svr_amd.model import ifnet as ifn
    B, N = 2, 4096
    g = torch.Generator().manual_seed(29)
    x = (torch.rand(B, 1, *DIMS, generator=g) < 0.03).float()
    pts = _surface_points(B, N, g)
    m, _ = _model()
    ext = m.ifnet_feature_extractor
    levels = ext.encode_levels(x.cuda())
    assert [tuple(v.shape[1:4]) for v in levels] == [(139, 104, 112), (139, 104, 112), (69, 52, 56), (34, 26, 28), (17, 13, 14), (8, 6, 7)]
    layout = ext._layout
    pg = pts.cuda()
    gfeat = torch.randn(B * N, layout.row_stride, generator=g).cuda()
    vols_c = [_ncdhw(v).requires_grad_(True) for v in levels]
    ref = O.gather_features(vols_c, pts, 128)                        # (B, 2583, N), row k = c*7 + j
    perm = layout.reference_permutation()
    valid = perm >= 0
    w = torch.empty(B, int(valid.sum()), N)
    w[:, perm[valid]] = gfeat.view(B, N, -1).cpu()[:, :, valid].permute(0, 2, 1)
    (ref * w).sum().backward()
    want = [v.grad for v in vols_c]
    for form in ("pull", "items"):
        saved, ifn.SCATTER_FORM = ifn.SCATTER_FORM, form
        try:
            orders, plans, ready = ifn._level_orders_async(pg, *DIMS, len(levels), False, layout, DISP)
        finally:
            ifn.SCATTER_FORM = saved
        torch.cuda.current_stream().wait_event(ready)
        if form == "pull":
            assert [p is not None for p in plans] == [False, True, True, True, False, False]
        gvols = [torch.full_like(v, float("nan")) if plans[l] is not None else torch.zeros_like(v) for l, v in enumerate(levels)]
        ops.gather_bwd(levels, gvols, pg, gfeat, layout, DISP, False, level_orders=orders, level_plans=plans)
        for l in range(len(levels)):
            e = G.rel_err(_ncdhw(gvols[l]).numpy(), want[l].numpy())
            assert e < 1e-5, (form, l, e)
    # compact kept-column matrix (levels 0-3), pull plans built for ITS columns; levels 4-5 skipped
    keep = [0, 1, 2, 3]
    klay = layout.subset(keep)
    gk = gfeat[:, klay.full_cols[:klay.row_stride].clamp(max=layout.row_stride - 1).cuda()].contiguous()
    saved, ifn.SCATTER_FORM = ifn.SCATTER_FORM, "pull"
    try:
        orders, plans, ready = ifn._level_orders_async(pg, *DIMS, len(levels), False, klay, DISP, proj_levels=(4, 5))
    finally:
        ifn.SCATTER_FORM = saved
    torch.cuda.current_stream().wait_event(ready)
    gvols = [torch.zeros_like(levels[0])] + [torch.full_like(levels[l], float("nan")) for l in (1, 2, 3)] + [None, None]
    ops.gather_bwd(levels, gvols, pg, gk, klay, DISP, False, level_orders=[None if l >= 4 else o for l, o in enumerate(orders)],
                   level_plans=plans, skip_levels=(4, 5))
    for l in keep:
        e = G.rel_err(_ncdhw(gvols[l]).numpy(), want[l].numpy())
        assert e < 1e-5, ("compact", l, e)
    # projected scatter of 256-wide rows on the two coarse grids (17x13x14, 8x6x7)
    dh = torch.randn(B * N, 256, generator=g)
    grid = O.sample_grid(pts, 128)
    for l in (4, 5):
        dims = tuple(levels[l].shape[1:4])
        items = ops.item_order(pg, dims, DISP, False, with_j=True)
        dP = ops.gather_project_bwd(pg, dh.cuda(), dims, items, DISP, False).cpu()
        plan = ops.project_plan(pg, dims, DISP, False)
        dP2 = ops.gather_project_bwd(pg, dh.cuda(), dims, plan, DISP, False).cpu()
        assert bool(torch.isfinite(dP2).all())
        for j in range(7):
            vol = torch.zeros(B, 256, *dims, requires_grad=True)
            out = F.grid_sample(vol, grid[:, :, j:j + 1], mode="bilinear", padding_mode="zeros", align_corners=False)
            (out[:, :, 0, 0].permute(0, 2, 1) * dh.view(B, N, 256)).sum().backward()
            r = vol.grad.permute(0, 2, 3, 4, 1).reshape(B, -1, 256)
            assert G.rel_err(dP[:, :, j].numpy(), r.numpy()) < 1e-5, (l, j)
            assert G.rel_err(dP2[:, :, j].numpy(), r.numpy()) < 1e-5, (l, j, "two-pass")


def test_dense_lattice_inference_at_139x104x112():
    """evaluate_network_on_grid on the reference's real lattice (model/ifnet.py:215-229: 139 * 104 * 112 = 1 619 072
    points, linspace(-0.5, 0.5) per axis INCLUDING the +-0.5 planes, 32 768-point chunks, eval-mode BatchNorm) against the
    oracle evaluated on (a) a strided subset of the interior and (b) every point of the six outer planes' strided subset
    (their samples touch out-of-range corners: the zero-padding path), sigmoid values within 1e-4 of the logit scale."""
    from svr_amd.model import evaluate_network_on_grid, make_3d_grid
    m, st = _model(train=False)
    g = torch.Generator().manual_seed(31)
    x = (torch.rand(1, 1, *DIMS, generator=g) < 0.03).float()
    got = evaluate_network_on_grid(m, x.cuda(), DIMS, 1)
    assert got.shape == DIMS and np.isfinite(got).all()
    lat = make_3d_grid((-0.5,) * 3, (0.5,) * 3, DIMS).view(*DIMS, 3)
    assert float(lat[0, 0, 0, 0]) == -0.5 and float(lat[-1, -1, -1, 2]) == 0.5
    sel = lat[3::7, 5::7, 2::7].reshape(-1, 3)                       # interior subset
    gsel = got[3::7, 5::7, 2::7].reshape(-1)
    faces = [lat[0, ::5, ::5], lat[-1, ::5, ::5], lat[::5, 0, ::5], lat[::5, -1, ::5], lat[::5, ::5, 0], lat[::5, ::5, -1]]
    gfaces = [got[0, ::5, ::5], got[-1, ::5, ::5], got[::5, 0, ::5], got[::5, -1, ::5], got[::5, ::5, 0], got[::5, ::5, -1]]
    psel = torch.cat([sel] + [f.reshape(-1, 3) for f in faces]).unsqueeze(0)
    gall = np.concatenate([gsel] + [f.reshape(-1) for f in gfaces])
    with torch.no_grad():
        z = O.ifnet_forward({k: v.clone() for k, v in st.items()}, x, psel, 128, training=False)
    want = torch.sigmoid(z).squeeze(0).numpy()
    zmax = float(z.abs().max())
    assert psel.shape[1] > 5000 and np.abs(gall - want).max() < 1e-4 * max(zmax, 1.0), np.abs(gall - want).max()
    # the same through the unprepared per-chunk path and with another chunk size: identical values
    again = evaluate_network_on_grid(m, x.cuda(), DIMS, 1, points_batch_size=50000)
    assert np.array_equal(again, got)


def test_prepared_query_refuses_what_it_was_not_prepared_for():
    """the fused kernel's workspace holds one box record per 64-point tile of the PREPARED point set; a
    larger chunk would write past it on the device.  gather_fc0_run / IFNet.query raise instead -- and a `prepared` handle
    of another pyramid is refused rather than silently used."""
    m, st = _model(train=False)
    g = torch.Generator().manual_seed(37)
    x = (torch.rand(1, 1, 32, 32, 32, generator=g) < 0.05).float().cuda()
    levels = m.encode(x)
    prep = m.prepare_query(levels, 1000)
    assert prep is not None and (prep.B, prep.N) == (1, 1000)
    pts = (torch.rand(1, 1000, 3, generator=g) - 0.5).cuda()
    z = m.query(levels, pts, prepared=prep)
    assert torch.equal(z[:, :300], m.query(levels, pts[:, :300].contiguous(), prepared=prep))     # smaller chunks are fine
    assert torch.equal(z, m.query(levels, pts))                                                   # = the unprepared path
    with pytest.raises(RuntimeError, match="prepared capacity"):
        m.query(levels, (torch.rand(1, 1001, 3, generator=g) - 0.5).cuda(), prepared=prep)
    with pytest.raises(RuntimeError, match="prepared capacity"):
        m.query(levels, torch.cat([pts, pts]), prepared=prep)                                     # another batch size
    other = m.encode((torch.rand(1, 1, 32, 32, 32, generator=g) < 0.05).float().cuda())
    with pytest.raises(RuntimeError, match="another pyramid"):
        m.query(other, pts, prepared=prep)
    torch.cuda.synchronize()
