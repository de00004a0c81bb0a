"""GPU parity of the config-5 composition (SceneNetTrainer mirror: stock-op UNet -> HIP project ->
HIP IF-Net -> HIP BCE + MSE) against the reference modules' own outputs (tests/golden/scene_*.npz)."""
import numpy as np
import pytest
import torch

from oracle import scene_oracle as S
from tests import _golden as G

pytestmark = pytest.mark.gpu


def test_scene_training_step_matches_reference():
    import svr_amd  # noqa: F401
    from svr_amd.trainer import SceneNetTrainer, default_hparams
    z = G.load("scene_cfg5small")
    batch, dims, scale = G.scene_inputs(z)
    tr = SceneNetTrainer(default_hparams(scale_factor=scale))
    assert tuple(int(v) for v in tr.dims) == dims
    tr.unet.load_state_dict(S.name_seeded_like(tr.unet.state_dict(), 1.0, "unet."), strict=False)
    tr.ifnet.load_state_dict(G.state(128, z=z), strict=False)
    tr = tr.cuda().train()
    b = {k: v.cuda() for k, v in batch.items()}
    logits, depth, pc = tr(b)
    assert G.rel_err(G.sample(depth, 8192), z["depth_s"]) < 2e-5          # hand-kernel UNet vs the reference's mkldnn convs
    assert G.rel_err(G.sample(pc, 8192), z["pc_s"]) < 2e-5
    # end to end vs the reference: the UNet runs on MIOpen here and mkldnn there (depth differs by ~1e-5),
    # and that difference is amplified by the splat/clamp/BatchNorm chain -> 1e-3 on the logits
    assert G.rel_err(logits.detach().cpu().numpy(), z["logits"]) < 1e-3
    # the hot path proper (project + IF-Net in HIP) against the oracle fed with the SAME depth map: 1e-4
    from oracle import ifnet_oracle as O
    from oracle import projection_oracle as P
    with torch.no_grad():
        dcpu = depth.detach().cpu()
        pc_ref = P.norm_grid_space(P.depthmap_to_gridspace(dcpu, scale), dims)
        vox_ref = P.project_forward(pc_ref, dims, torch.tensor([1.5, 1.5, 1.5]), (3, 3, 3))
        ist = {k: v.clone() for k, v in G.state(128, z=z).items()}
        logits_ref = O.ifnet_forward(ist, vox_ref, batch["points"], 128, training=True)
    assert G.rel_err(logits.detach().cpu().numpy(), logits_ref.numpy()) < 1e-4
    loss = tr.losses_and_logging(b, depth, logits, b["occupancies"])
    assert abs(loss.item() - float(z["loss"])) < 1e-4 * float(z["loss"])
    assert abs(tr.last_log["train_ce_loss"].item() - float(z["ce"])) < 1e-4 * float(z["ce"])
    loss.backward()
    # gradients: mask-flip sensitivity as in test_gpu_ifnet_parity (tools/gradient_sensitivity.py)
    assert G.rel_err(tr.project.sigma.grad.cpu().numpy(), z["sigma_grad"]) < 2e-2
    for prefix, mod in (("unet.", tr.unet), ("ifnet.", tr.ifnet)):
        top = max(float(z["grad_norm/" + prefix + n]) for n, _ in mod.named_parameters())
        for name, p in mod.named_parameters():
            ref_n = float(z["grad_norm/" + prefix + name])
            got_n = p.grad.double().norm().item()
            if ref_n < 1e-3 * top:
                # e.g. a conv bias directly followed by BatchNorm: its true gradient is 0, both sides hold rounding noise
                assert got_n < 2e-3 * top, (prefix + name, got_n, ref_n)
                continue
            # 35x26x28 grid, 300 points, batch 2: a depth map that differs in the 6th digit (hand-kernel UNet here,
            # mkldnn in the fixture) moves a few of the 76 800 projected points across voxel boundaries, and at this
            # size one flipped voxel shifts a deep encoder bias gradient by a few per cent (observed 2.8 %)
            assert abs(got_n - ref_n) < 5e-2 * ref_n, (prefix + name, got_n, ref_n)
            got, ref = G.sample(p.grad, 256).astype(np.float64), z["grad/" + prefix + name].astype(np.float64)
            # UNet gradients arrive through the projection's discrete voxelisation: the 1e-5 MIOpen-vs-mkldnn depth
            # difference moves some points across voxel boundaries, which shifts d(loss)/d(depth) by ~1 % (observed
            # median 3-4e-3 of the largest element, run-dependent with MIOpen's algorithm choice) -> 1e-2 for them
            gate = 1e-2          # (IF-Net encoder tensors see the same voxel flips: observed 7e-3 on conv_in_bn.weight)
            assert np.median(np.abs(got - ref)) <= gate * np.abs(ref).max(), prefix + name
    # the Lightning contract
    out = tr.training_step(b, 0)
    assert set(out) == {"loss"} and out["loss"].dim() == 0
    opt = tr.configure_optimizers()[0][0]
    assert [g["lr"] for g in opt.param_groups] == [1e-4, 1e-3, 1e-4]


def test_scene_step_ifnet_gradients_with_the_reference_depth_map():
    """The same fixture with the UNet taken out of the comparison: the depth map is computed by the CPU oracle's UNet (the
    fixture's own arithmetic -- pinned against its depth samples below) and fed to the HIP project + IF-Net path as is
    (`skip_unet`), so no voxel can flip because of a 1e-6 depth difference.  IF-Net's parameter gradients and sigma's only
    depend on the cross-entropy term, and are held to the ORIGINAL gates of this file (gradient norms 2e-2, sampled medians
    5e-3, logits 1e-4 against the reference itself); the wider gates above apply to the run through the hand-kernel UNet."""
    import torch.nn.functional as F
    import svr_amd  # noqa: F401
    from svr_amd.trainer import SceneNetTrainer, default_hparams
    z = G.load("scene_cfg5small")
    batch, dims, scale = G.scene_inputs(z)
    full = SceneNetTrainer(default_hparams(scale_factor=scale))
    unet_st = S.name_seeded_like(full.unet.state_dict(), 1.0, "unet.")
    with torch.no_grad():
        raw = S.unet_forward({k: v.clone() for k, v in unet_st.items()}, batch["rgb"], "full", True)
        zz = F.interpolate(raw, size=320, mode="bilinear")[:, :, 40:280, :].squeeze(1)
        depth_ref = torch.sigmoid(zz) * (7.0 - 0.1953997164964676) + 0.1953997164964676
    assert G.rel_err(G.sample(depth_ref, 8192), z["depth_s"]) < 1e-6          # the oracle's UNet IS the fixture's
    tr = SceneNetTrainer(default_hparams(scale_factor=scale, skip_unet=True))
    tr.ifnet.load_state_dict(G.state(128, z=z), strict=False)
    tr = tr.cuda().train()
    b = {k: v.cuda() for k, v in batch.items()}
    b["depthmap_target"] = depth_ref.cuda()
    logits, depth, pc = tr(b)
    assert G.rel_err(logits.detach().cpu().numpy(), z["logits"]) < 1e-4
    loss = tr.losses_and_logging(b, depth, logits, b["occupancies"])           # MSE(depth, depth) = 0: the CE term alone
    assert abs(loss.item() - float(z["ce"])) < 1e-4 * float(z["ce"])
    loss.backward()
    assert G.rel_err(tr.project.sigma.grad.cpu().numpy(), z["sigma_grad"]) < 2e-2
    top = max(float(z["grad_norm/ifnet." + n]) for n, _ in tr.ifnet.named_parameters())
    for name, p in tr.ifnet.named_parameters():
        ref_n = float(z["grad_norm/ifnet." + name])
        got_n = p.grad.double().norm().item()
        if ref_n < 1e-3 * top:
            assert got_n < 2e-3 * top, (name, got_n, ref_n)
            continue
        assert abs(got_n - ref_n) < 2e-2 * ref_n, (name, got_n, ref_n)
        got, ref = G.sample(p.grad, 256).astype(np.float64), z["grad/ifnet." + name].astype(np.float64)
        assert np.median(np.abs(got - ref)) <= 5e-3 * np.abs(ref).max(), name
