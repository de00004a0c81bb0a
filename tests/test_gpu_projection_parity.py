"""GPU parity of the projection HIP path (unproject / splat / blur) against the reference's own
outputs (tests/golden/project_*.npz) and the CPU oracle.  Voxel indices and the validity mask are
bit-exact; floating point within the written tolerances (f32 atomics reorder the splat sums)."""
import numpy as np
import pytest
import torch

from oracle import projection_oracle as P
from tests import _golden as G

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", G.PROJECT_CASES)
def test_project_matches_reference(case):
    import svr_amd  # noqa: F401
    from svr_amd import ops
    from svr_amd.model import project
    z = G.load("project_" + case)
    depth, wsel, dims, ks, sigma, scale = G.project_inputs(z)
    stride = int(z["stride"])
    mod = project(dims, ks, sigma).cuda()
    dg = depth.cuda().requires_grad_(True)
    gp = mod.depthmap_to_gridspace(dg, scale)
    assert G.rel_err(gp.detach().cpu().reshape(-1, 3)[::stride].numpy(), z["grid_pc_s"]) < 1e-6
    npc = mod.norm_grid_space(gp)
    assert G.rel_err(npc.detach().cpu().reshape(-1, 3)[::stride].numpy(), z["norm_pc_s"]) < 2e-6
    fused = mod.depthmap_to_gridspace(dg, scale, normalize=True)
    assert G.rel_err(fused.detach().cpu().numpy(), npc.detach().cpu().numpy()) < 2e-6
    # bit-exact gate: defined at pc_voxels' input -> feed the oracle's own fp32 points
    npc_cpu = P.norm_grid_space(P.depthmap_to_gridspace(depth, scale), dims)
    _, base, valid = ops.splat_fwd(npc_cpu.cuda().contiguous(), dims, want_indices=True)
    v_ref, b_ref, _ = P.splat_indices(npc_cpu, dims)
    assert np.array_equal(np.packbits(valid.cpu().numpy()), z["valid_bits"])
    assert torch.equal(valid.cpu().bool(), v_ref)
    assert torch.equal(base.cpu().long()[v_ref], b_ref[v_ref])
    assert np.array_equal(base.cpu().reshape(-1, 3)[::stride].to(torch.int16).numpy(), z["vox_idx_s"])
    raw = mod.pc_voxels(npc)
    assert G.rel_err(G.sample(raw, 16384), z["vox_raw_s"]) < 1e-5
    occ = mod(npc)
    assert occ.shape == (depth.shape[0], 1) + tuple(dims)
    assert G.rel_err(G.sample(occ, 16384), z["occ_s"]) < 1e-5
    assert abs(occ.double().sum().item() - float(z["occ_sum"])) < 1e-5 * float(z["occ_sum"])
    (occ * wsel.cuda()).sum().backward()
    assert G.rel_err(mod.sigma.grad.cpu().numpy(), z["sigma_grad"]) < 2e-3
    assert G.rel_err(G.sample(dg.grad, 16384), z["depth_grad_s"]) < 2e-4
    assert abs(dg.grad.double().norm().item() - float(z["depth_grad_norm"])) < 2e-4 * float(z["depth_grad_norm"])


def test_splat_ignores_out_of_range_and_nan_points():
    import svr_amd  # noqa: F401
    from svr_amd import ops
    pts = torch.tensor([[[0.6, 0.0, 0.0], [float("nan"), 0.0, 0.0], [0.0, 0.0, 0.0], [-0.5, 0.1, 0.1],
                         [0.4999995, 0.0, 0.0]]])
    acc, base, valid = ops.splat_fwd(pts.cuda(), (5, 6, 7), want_indices=True)
    ref = P.pc_voxels(pts, (5, 6, 7))
    assert valid.cpu().tolist() == [[0, 0, 1, 0, 0]]
    assert G.rel_err(ops.scale_clamp01(acc, 8.0).cpu().numpy(), ref.numpy()) < 1e-6
    empty, _, _ = ops.splat_fwd(torch.zeros(2, 0, 3).cuda(), (4, 4, 4))
    assert float(empty.abs().sum()) == 0.0
